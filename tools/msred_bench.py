"""Throughput of the MS-REDNet inference model on synthetic tiles.

    python tools/msred_bench.py [--views 5 --height 384 --width 768 --ndepths 48,32,8 --numdepth 192 --batch 1 --steps 5]

Prints depth maps/s of the three stages on resident features (eager launches, and one hipGraph replay per step),
FeatureNet time, and the per-stage split.  Under rocprofv3 --kernel-trace --stats it gives the per-kernel picture.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import synth  # noqa: E402
from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--ndepths", default="48,32,8")
    ap.add_argument("--numdepth", type=int, default=192)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    nd = [int(x) for x in a.ndepths.split(",")]
    cfg = dict(views=a.views, H=a.height, W=a.width, ndepths=nd, num_depth=a.numdepth)
    m = Infer_CascadeREDNet(a.numdepth, nd, synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=a.batch, seed=0)
    imgs, dv = imgs.cuda(), dv.cuda()
    proj = {k: v.cuda() for k, v in proj.items()}
    interval = (float(dv[0, -1]) - float(dv[0, 0])) / a.numdepth
    with torch.no_grad():
        maps, shapes = m.extract_features(imgs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m.extract_features(imgs)
        torch.cuda.synchronize()
        t_feat = (time.perf_counter() - t0) / 3
        out = m.infer_from_features(maps, shapes, proj, dv, interval)          # warm-up (packs weights)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = m.infer_from_features(maps, shapes, proj, dv, interval)
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / a.steps
        print("MS-REDNet %d views %dx%d ndepths %s batch %d: FeatureNet %.2f ms, stages %.1f ms eager -> %.2f maps/s"
              % (a.views, a.width, a.height, a.ndepths, a.batch, t_feat * 1e3, t_eager * 1e3, a.batch / t_eager))
        if not a.no_graph:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                m.infer_from_features(maps, shapes, proj, dv, interval)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                out_g = m.infer_from_features(maps, shapes, proj, dv, interval)
            g.replay()
            torch.cuda.synchronize()
            err = float((out_g["depth"] - out["depth"]).abs().max())
            t0 = time.perf_counter()
            for _ in range(a.steps):
                g.replay()
            torch.cuda.synchronize()
            t_graph = (time.perf_counter() - t0) / a.steps
            print("hipGraph replay: %.1f ms -> %.2f maps/s (max |depth - eager| = %.2e); images to maps %.2f maps/s"
                  % (t_graph * 1e3, a.batch / t_graph, err, a.batch / (t_graph + t_feat)))


if __name__ == "__main__":
    main()
