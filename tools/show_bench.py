"""Print the essentials of bench.py JSON lines: python tools/show_bench.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        print("%s: %.1f %s, %.2f ms/step, %s" % (path, d["value"], d["unit"], d["ms_per_step"], d["config"]["workload"]))
        if "feature_net_ms_per_tile" in d:
            print("   FeatureNet0 %.3f ms per tile; images to maps %.1f maps/s per GPU" % (d["feature_net_ms_per_tile"], d.get("end_to_end_maps_per_s_per_gpu", 0.0)))
        if "phase_ms_per_step" in d:
            print("   phases:", "  ".join("%s %.2f" % (k.split(".", 1)[-1], v) for k, v in d["phase_ms_per_step"].items()))
        if "cost_reg_layers_ms" in d:
            print("   CostRegNet2D layers:", "  ".join("%s %.2f" % kv for kv in d["cost_reg_layers_ms"].items()))
        r = d.get("roofline")
        if r:
            print("   roofline: %.1f %s (%.1f %% of %.1f), %.2f ms per launch" % (r["achieved"], r["unit"], 100 * r["frac"], r["peak"], r.get("launch_ms", float("nan"))))
