#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/<name>_traffic.json (per-launch KiB per kernel).

    python tools/make_traffic_json.py <workload> <tiles_per_launch> <out.json> <pmc_dir> [<pmc_dir> ...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_stamp  # noqa: E402  (bench.py refuses a traffic file measured on another build)

workload, tiles, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
seen = set()
for d in sys.argv[4:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "adamvs" not in k:
                continue
            c = acc[k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
            if (path, r["Dispatch_Id"]) not in seen:
                seen.add((path, r["Dispatch_Id"]))
                dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                dur[k][1] += 1
kernels = {}
for k, cs in sorted(acc.items(), key=lambda kv: -dur[kv[0]][0]):
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        kernels[k[:96]] = {"fetch_size_kib": round(cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1], 1),
                           "write_size_kib": round(cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1], 1),
                           "avg_us": round(dur[k][0] / dur[k][1], 1), "launches": dur[k][1]}
# whole hot path: every kernel of a pass (= one bench step) summed; FeatureNet0 and the repack kernels run in the setup only.
# passes = launches of k_pair_similarity (one per pass: stage 1, pass A)
setup = ("k_fconv", "k_context", "k_conv0_fused", "k_pack_nhwc", "k_unpack_nchw", "k_relative_transforms")
passes = max([v["launches"] for k, v in kernels.items() if "k_pair_similarity" in k] or [0])
hot = {k: v for k, v in kernels.items() if not any(s in k for s in setup)}
per_pass = sum((2 * v["fetch_size_kib"] + v["write_size_kib"]) * 1024 * v["launches"] for v in hot.values()) / passes if passes else None
json.dump({"config": {"workload": workload, "tiles_per_launch": tiles}, "precision": os.environ.get("PRECISION", "fp32"),
           "source_stamp": source_stamp(), "passes": passes, "hot_path_bytes_per_pass": per_pass,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --batch %d --no-graph --steps 1 "
                   "--warmup 1); per-launch averages, KiB. gfx950: FETCH_SIZE counts half of a 16-byte-per-lane coalesced read "
                   "(MI355X_MICROARCH.md, HBM) -> hbm_bytes = 2*FETCH + WRITE for the float4 kernels." % tiles,
           "kernels": kernels}, open(out, "w"), indent=1)
print("wrote", out, len(kernels), "kernels")
