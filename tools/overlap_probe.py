#!/usr/bin/env python3
"""Do pass A (pair similarity + CostRegNet2D + softmax: bound by the fp32 matrix pipe) and pass B (aggregation, recurrence,
soft-argmin: bound by memory traffic and latency) of DIFFERENT tile groups overlap when issued on two HIP streams?

    python tools/overlap_probe.py [--tiles 64] [--iters 3] [--precision fp32]

Times K passes A of group 0 alone, K passes B of group 1 alone, and both at once (stream 0: A, A, ...; stream 1: B, B, ...).
If the two-stream wall time is well below the sum, a step can be software-pipelined over tile groups (pass B of group g next
to pass A of group g + 1)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import _lib, hip_ops, synth  # noqa: E402
from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--same", action="store_true", help="both streams run the WHOLE stage of their tile group (small groups: do latency-bound launches of two groups share the chip?)")
    a = ap.parse_args()
    cfg = "cfg2"
    c = synth.CONFIGS[cfg]
    dev = torch.device("cuda:0")
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO[:1], False, [8, 8, 8], precision=a.precision)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.to(dev).eval()
    B = a.tiles
    net = m.DepthNet[0]
    groups = []
    with torch.no_grad():
        for g in range(2):
            imgs = torch.cat([synth.tile_inputs(cfg, 1, seed=g * B + t)[0] for t in range(B)], 0).to(dev)
            _, proj, dv = synth.tile_inputs(cfg, batch=B, seed=0)
            feats, shapes = m.extract_features(imgs)
            del imgs
            Bq, C, h, w = shapes[0]
            rt = hip_ops.relative_transforms(proj["stage1"].to(dev))
            planes = hip_ops.plane_source(dv.to(dev), c["ndepths"][0], 4.0 * (200.0 / c["num_depth"]), [Bq, h, w])
            S = c["views"] - 1
            outs = (torch.zeros(S, B, h, w, device=dev), torch.zeros(S, B, h, w, device=dev),
                    torch.zeros(B, 2 * h, 2 * w, device=dev), torch.zeros(B, 2 * h, 2 * w, device=dev))
            groups.append(dict(feat=feats[0], shape=shapes[0], rt=rt, planes=planes, outs=outs, ws={}))
        REST = _lib.PHASE_AGGREGATE | _lib.PHASE_RECURRENCE | _lib.PHASE_SOFT_ARGMIN

        def run(g, phases):
            G = groups[g]
            Bq, C, h, w = G["shape"]
            net.run(G["feat"], Bq, C, h, w, G["rt"], None, None, g, False, planes=G["planes"], num_depth=c["ndepths"][0],
                    workspaces=G["ws"], phases=phases, outputs=G["outs"])

        for g in range(2):                      # warm up: workspaces, view weights of both groups
            run(g, _lib.PHASE_VIEW_WEIGHTS)
            run(g, REST)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

        def timed(do_a, do_b):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                if do_a:
                    with torch.cuda.stream(s0):
                        run(0, _lib.PHASE_VIEW_WEIGHTS)
                if do_b:
                    with torch.cuda.stream(s1):
                        run(1, REST)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / a.iters

        if a.same:
            ALL = _lib.PHASE_VIEW_WEIGHTS | REST

            def both(two_streams):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    for g in range(2):
                        with torch.cuda.stream((s0, s1)[g] if two_streams else s0):
                            run(g, ALL)
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t0) / a.iters
            both(False), both(True)
            t1, t2 = both(False), both(True)
            print("whole stage 1 of two groups of %d tiles (%s): one stream %.2f ms, two streams %.2f ms (%.2f x)" % (B, a.precision, t1, t2, t2 / t1))
            return
        ta, tb = timed(True, False), timed(False, True)
        tab = timed(True, True)
        print("pass A alone %.1f ms, pass B alone %.1f ms, sum %.1f ms; both streams %.1f ms (%.2f x the sum) -- %d tiles per group, %s"
              % (ta, tb, ta + tb, tab, tab / (ta + tb), B, a.precision))


if __name__ == "__main__":
    main()
