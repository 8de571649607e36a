"""Profiling driver for one recurrent regularisation step (SliceCostRegNetRED) at a workload's stage-1 shape.

    rocprofv3 --kernel-trace --stats -d /tmp/p -- python3 tools/step_prof.py --workload cfg2 --batch 32 --iters 40

Runs `iters` steps of adamvs_slice_reg_step (conv1, both ConvGRU levels, decoder) on random maps; the per-kernel
averages of the rocprof summary are the per-step costs of the recurrence.  Prints wall time per step as well.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import hip_ops, synth  # noqa: E402
from ada_mvs_amd._lib import PRECISIONS  # noqa: E402
from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--stage", type=int, default=0)
    ap.add_argument("--precision", default="fp32")
    a = ap.parse_args()
    c = synth.CONFIGS[a.workload]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO[:len(c["ndepths"])], False, [8, 8, 8],
                        precision=a.precision)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    net = m.DepthNet[a.stage].reg_fuse
    scale = 4 >> a.stage if len(c["ndepths"]) == 3 else 4
    h, w = c["H"] // scale, c["W"] // scale
    C = net.in_channels
    B = a.batch
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    cost = torch.randn(B, h * w, C, generator=g).to(dev)
    s1 = torch.zeros(B, h * w, 8, device=dev)
    s2 = torch.zeros(B, (h // 2) * (w // 2), 16, device=dev)
    fuse = net.packed(dev)
    prec = PRECISIONS[a.precision]
    for _ in range(3):
        hip_ops.slice_reg_step(cost, s1, s2, fuse, B, C, h, w, net.up, prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        hip_ops.slice_reg_step(cost, s1, s2, fuse, B, C, h, w, net.up, prec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print("%s stage %d  B=%d  %dx%d C=%d  %s: %.1f us per step (wall, eager launches)" % (a.workload, a.stage, B, h, w, C,
                                                                                         a.precision, dt * 1e6))


if __name__ == "__main__":
    main()
