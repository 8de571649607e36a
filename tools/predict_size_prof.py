"""Time and peak device memory of one forward at the reference's default predict configuration
(predict_whu.py:24-55: 5 views, --max_w 3712 --max_h 5504 --resize_scale 0.5 -> 1856 x 2752, --ndepths 48,32,8
--numdepth 192, one sample per step, eager launches, synthetic views).

    python tools/predict_size_prof.py [--model adamvs|msrednet] [--precision fp32|bf16x3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="adamvs")
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
cfg = dict(views=5, H=2752, W=1856, ndepths=[48, 32, 8], num_depth=192)
if a.model == "adamvs":
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet as M
    m = M(cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision=a.precision)
else:
    from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet as M
    m = M(cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
m.load_state_dict(synth.seeded_state_dict(m, seed=0))
m = m.cuda().eval()
imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=0)
args = (imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
with torch.no_grad():
    for _ in range(2):
        out = m(*args)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(5):
        out = m(*args)
    torch.cuda.synchronize()
eager_ms = (time.perf_counter() - t0) / 5 * 1e3
if a.model == "adamvs":
    # the predict loop's default since round 6: one captured hipGraph per input shape (ada_mvs_amd/graphed.py), host depth_values
    from ada_mvs_amd.graphed import GraphedForward
    fwd = GraphedForward(m)
    dv_host = dv.clone()
    with torch.no_grad():
        for _ in range(2):
            g = fwd(args[0], args[1], dv_host)
        torch.cuda.synchronize()
        same = bool(torch.equal(g["depth"], out["depth"])) and bool(torch.equal(g["photometric_confidence"], out["photometric_confidence"]))
        t0 = time.perf_counter()
        for i in range(5):
            dv_host[0, 1] = 600.0 + i            # another depth range per sample: the same graph
            g = fwd(args[0], args[1], dv_host)
        torch.cuda.synchronize()
    print("%s %s  graphed forward (one hipGraph per shape): %.1f ms per forward against %.1f eager; maps bit-identical to the eager forward: %s" % (
        a.model, a.precision, (time.perf_counter() - t0) / 5 * 1e3, eager_ms, same))
    t0 = time.perf_counter() - eager_ms * 5e-3
print("%s %s  1856x2752, 5 views, 48/32/8: %.1f ms per forward, peak device memory %.2f GB (allocated now %.2f GB), depth %s finite %s" % (
    a.model, a.precision, (time.perf_counter() - t0) / 5 * 1e3, torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_allocated() / 1e9,
    tuple(out["depth"].shape), bool(torch.isfinite(out["depth"]).all())))
