"""Profiling driver for FeatureNet0 (adamvs_feature_net0):  rocprofv3 --kernel-trace --stats -- python3 tools/feat_prof.py --images 160"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import synth  # noqa: E402
from ada_mvs_amd.models.adamvs import FeatureNet0  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=160)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--torch", action="store_true", help="the PyTorch/MIOpen layers instead of the HIP kernels")
    a = ap.parse_args()
    net = FeatureNet0(8)
    net.load_state_dict(synth.seeded_state_dict(net, seed=2))
    net = net.cuda().eval()
    x = torch.randn(a.images, 3, a.height, a.width, device="cuda")
    run = (lambda: net.forward_torch(x)) if a.torch else (lambda: net.forward_cl(x))
    with torch.no_grad():
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            run()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print("FeatureNet0 %s: %d images %dx%d: %.2f ms = %.1f us per image" % ("torch" if a.torch else "hip", a.images, a.width,
                                                                          a.height, dt * 1e3, dt * 1e6 / a.images))


if __name__ == "__main__":
    main()
