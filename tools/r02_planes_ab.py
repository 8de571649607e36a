"""A/B of generated vs materialised hypothesis planes: python tools/r02_planes_ab.py [--workload cfg2] [--batch 128]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import synth  # noqa: E402
from bench import build_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, _ = build_model(a.workload, dev, a.precision)
c = synth.CONFIGS[a.workload]
imgs, proj, dv = synth.tile_inputs(a.workload, batch=a.batch, seed=0)
proj = {k: v.to(dev) for k, v in proj.items()}
dv = dv.to(dev)
interval = (synth.DEPTH_RANGE[1] - synth.DEPTH_RANGE[0]) / c["num_depth"]
with torch.no_grad():
    feats, shapes = model.extract_features(imgs.to(dev))
    del imgs
    for flag in (True, False, True, False):
        model.materialize_planes = flag
        for _ in range(2):
            model.infer_from_features(feats, shapes, proj, dv, interval)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.infer_from_features(feats, shapes, proj, dv, interval)
        torch.cuda.synchronize()
        print("%s B=%d %s planes %-12s %8.2f ms/step" % (a.workload, a.batch, a.precision, "materialised" if flag else "generated", (time.perf_counter() - t0) / 3 * 1e3))
