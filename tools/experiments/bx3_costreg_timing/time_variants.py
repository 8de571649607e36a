#!/usr/bin/env python3
"""GPU box: one stride-1 CostRegNet2D layer (conv0's shape at cfg3 / 32 tiles: 128 maps of 96 x 192, D = 192) in the bf16x3 mode on the
shipped library and on every timing build of tools/experiments/timing_builds.py, each in its own process (ADAMVS_LIB_PATH)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
import ada_mvs_amd
from ada_mvs_amd import hip_ops, packing
N, D, h, w = 128, 192, 96, 192
g = torch.Generator().manual_seed(0)
x = torch.randn(N, h * w, D, generator=g).cuda()
wt = torch.randn(D, D, 3, 3, generator=g) / (3 * D ** 0.5)
pk = packing.pack_reg_layer_bf16x3(wt, torch.ones(D), torch.zeros(D), False).cuda()
nw = pk.numel() - D
out = torch.empty(N, h * w, D, device="cuda")
for _ in range(3):
    hip_ops.conv3x3_dd(x, pk[:nw], pk[nw:], None, N, D, h, w, 0, 1, out=out, precision=1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    hip_ops.conv3x3_dd(x, pk[:nw], pk[nw:], None, N, D, h, w, 0, 1, out=out, precision=1)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print("%%-16s sum %%.6e " %% (sys.argv[1], float(out.double().abs().sum())), end="")
print("%%-16s %%.3f ms per layer   %%.0f TFLOP/s of issued bf16 products (3 per product of the direct form)" %% (sys.argv[1], ms, 3 * 2 * 9 * D * D * h * w * N / ms / 1e9))
''' % ROOT
names = ["shipped"] + sorted(f[len("libadamvs_hip."):-3] for f in os.listdir(os.path.join(ROOT, "ada-mvs_amd")) if f.startswith("libadamvs_hip.bxc_") and f.endswith(".so"))
for n in names:
    env = dict(os.environ)
    if n != "shipped":
        env["ADAMVS_LIB_PATH"] = os.path.join(ROOT, "ada-mvs_amd", "libadamvs_hip.%s.so" % n)
    r = subprocess.run([sys.executable, "-c", CODE, n], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-600:], flush=True)
