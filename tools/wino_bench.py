#!/usr/bin/env python3
"""One stride-1 CostRegNet2D layer on the direct kernel and in the F(2x2, 3x3) and F(2x4, 3x3) forms (GPU box): error of both against a float64
convolution on small maps, then the time of the five layer shapes of cfg2 at 128 tiles (N = 512 maps, D = 192).

    python tools/wino_bench.py                      -> profiles/r03_wino_layer_bench.txt is the output of this command
    python tools/wino_bench.py --widths             efficiency of the three forms against the network width (D = 64 ... 384, 96x192 maps)
    python tools/wino_bench.py --softmax            `prob` + softmax / regression: two ops through the score volume against the fused partials + merge
    python tools/wino_bench.py --one                one shape (128 maps, D = 192, 96x192), for timing builds and counter passes
                                                    (profiles/r04_wino_forms.txt = the first three of these, one after the other)
    ADAMVS_LIB_PATH=ada-mvs_amd/libadamvs_hip.<name>.so python tools/wino_bench.py --time-only     (a timing build, tools/build_variant.py)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import hip_ops, packing  # noqa: E402

# the F(2x4, 3x3) form lives outside the product library (tools/experiments/wino24: `wino24.py --build` first); without it the
# third column is skipped
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments", "wino24"))
try:
    import wino24  # noqa: E402
    HAVE24 = os.path.exists(wino24.LIB)
except Exception:          # noqa: BLE001
    wino24, HAVE24 = None, False

dev = "cuda"


def accuracy(N, D, h, w, relu=1, skip=False):
    g = torch.Generator().manual_seed(N * 1000 + D + h + w)
    x = torch.randn(N, D, h, w, generator=g)
    wt = torch.randn(D, D, 3, 3, generator=g) / (3 * D ** 0.5)
    scale, b = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double(), (wt * scale.reshape(-1, 1, 1, 1)).double(), b.double(), padding=1)
    ref = torch.relu(ref) if relu else ref
    x_cl = x.permute(0, 2, 3, 1).reshape(N, h * w, D).contiguous().to(dev)
    sk = torch.randn(N, h * w, D, generator=g).to(dev) if skip else None
    if skip:
        ref = ref + sk.cpu().double().reshape(N, h, w, D).permute(0, 3, 1, 2)
    pk = packing.pack_reg_layer(wt, scale, b, False).to(dev)
    yd = hip_ops.conv3x3_dd(x_cl, pk[:9 * D * D], pk[9 * D * D:], sk, N, D, h, w, 0, relu)
    yw = hip_ops.conv3x3_dd_wino(x_cl, packing.pack_reg_layer_wino(wt, scale).to(dev), b.to(dev), sk, N, D, h, w, relu)
    y4 = wino24.conv3x3_dd_wino24(x_cl, wino24.pack_reg_layer_wino24(wt, scale).to(dev), b.to(dev), sk, N, D, h, w, relu) if HAVE24 else yw
    torch.cuda.synchronize()
    back = lambda y: y.cpu().double().reshape(N, h, w, D).permute(0, 3, 1, 2)
    rel = lambda y: (back(y) - ref).abs().mean().item() / ref.abs().mean().item()
    print("N=%d D=%d %dx%d relu=%d skip=%d   relative L1 against float64: direct %.2e  F(2x2,3x3) %.2e  F(2x4,3x3) %.2e" % (N, D, h, w, relu, skip, rel(yd), rel(yw), rel(y4)),
          flush=True)


def timing(N, D, h, w, reps=5):
    x_cl = torch.randn(N, h * w, D, device=dev)
    wt = torch.randn(D, D, 3, 3) / (3 * D ** 0.5)
    b = torch.randn(D) * 0.1
    pk = packing.pack_reg_layer(wt, torch.ones(D), b, False).to(dev)
    pw, bias = packing.pack_reg_layer_wino(wt, torch.ones(D)).to(dev), b.to(dev)
    pw4 = wino24.pack_reg_layer_wino24(wt, torch.ones(D)).to(dev) if HAVE24 else None
    out = torch.empty(N, h * w, D, device=dev)
    flops = 2.0 * 9 * D * D * h * w * N
    for name, fn, executed in (("direct", lambda: hip_ops.conv3x3_dd(x_cl, pk[:9 * D * D], pk[9 * D * D:], None, N, D, h, w, 0, 1, out=out), 1.0),
                               ("F(2x2,3x3)", lambda: hip_ops.conv3x3_dd_wino(x_cl, pw, bias, None, N, D, h, w, 1, out=out), 16.0 / 36.0),
                               ("F(2x4,3x3)", (lambda: wino24.conv3x3_dd_wino24(x_cl, pw4, bias, None, N, D, h, w, 1, out=out)) if HAVE24 else None, 24.0 / 72.0)):
        if fn is None:
            continue
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("N=%d D=%d %dx%d %-11s %7.3f ms   executed %.1f TFLOP/s (%.1f %% of 157.3)   direct-form %.1f TFLOP/s" % (
            N, D, h, w, name, ms, flops * executed / ms / 1e9, flops * executed / ms / 1e9 / 157.3 * 100, flops / ms / 1e9), flush=True)


def softmax_timing(N=512, D=192, h=96, w=192, B=128, reps=5):
    """`prob` + softmax / regression: two ops through the score volume against the fused partials + merge."""
    x_cl = torch.randn(N, h * w, D, device=dev)
    wt = torch.randn(D, D, 3, 3) * (2.0 / (9 * D)) ** 0.5 * 3.0
    pw, bias = packing.pack_reg_layer_wino(wt, torch.ones(D)).to(dev), (torch.randn(D) * 0.3).to(dev)
    planes = (400.0 + (180.0 / D) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) + torch.zeros(B, 1, h, w)).contiguous().to(dev)
    out = torch.empty(N, h * w, D, device=dev)

    def two():
        hip_ops.conv3x3_dd_wino(x_cl, pw, bias, None, N, D, h, w, 0, out=out)
        return hip_ops.softmax_max_regress(out, planes, N // B, B, D, h, w)
    for name, fn in (("two ops", two), ("fused", lambda: hip_ops.prob_softmax_regress_wino(x_cl, pw, bias, torch.tensor([[400.0, 580.0]] * B, device=dev), N // B, B, D, h, w))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("prob + softmax, N=%d D=%d %dx%d  %-8s %7.3f ms" % (N, D, h, w, name, e0.elapsed_time(e1) / reps), flush=True)


if __name__ == "__main__":
    if "--softmax" in sys.argv:
        softmax_timing()
        sys.exit(0)
    if "--one" in sys.argv:                          # one shape, for timing builds
        timing(128, 192, 96, 192)
        sys.exit(0)
    if "--widths" in sys.argv:                       # efficiency against the width (the transformed filters of a layer against the 4 MB L2 of an XCD)
        for D, N in ((64, 512), (128, 256), (192, 128), (256, 128), (384, 64)):
            timing(N, D, 96, 192)
        sys.exit(0)
    if "--time-only" not in sys.argv:
        for case in ((1, 192, 6, 32), (2, 192, 13, 45, 0), (2, 192, 7, 70, 1, True), (1, 64, 8, 40), (1, 128, 13, 33), (1, 256, 6, 32, 0)):
            accuracy(*case)
    for shape in ((512, 192, 96, 192), (512, 192, 48, 96), (512, 192, 24, 48), (512, 192, 12, 24), (16, 192, 96, 192)):
        timing(*shape)
