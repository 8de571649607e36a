#!/usr/bin/env python3
"""F(2x4, 3x3) form of a stride-1 CostRegNet2D layer: an EXPERIMENT, not part of libadamvs_hip.so (moved out of the product
library in round 5: adamvs_cost_reg_net_2d never used it -- 3.5 % faster than F(2x2, 3x3) at the widest level only, slower on
every smaller one; costreg2d_wino24.hip's header has the numbers).

    python tools/experiments/wino24/wino24.py --build          hipcc -> tools/experiments/wino24/libadamvs_wino24.so (links the product
                                                               library for set_error / make_tile_grid / resident_blocks)
    python tools/experiments/wino24/wino24.py --check          CPU: the packing evaluates the convolution (no GPU needed)
    python tools/experiments/wino24/wino24.py --check-gpu      GPU box: the kernel against a float64 convolution and the direct kernel
    tools/wino_bench.py imports conv3x3_dd_wino24 / pack_reg_layer_wino24 from here when the library has been built.
"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import build as B, packing  # noqa: E402
from ada_mvs_amd.packing import WINO_G  # noqa: E402

LIB = os.path.join(HERE, "libadamvs_wino24.so")


def build():
    B.build(verbose=False)
    cmd = [B._hipcc()] + B.FLAGS + ["-shared", os.path.join(HERE, "costreg2d_wino24.hip"), "-o", LIB, "-L", B.HERE, "-l:libadamvs_hip.so",
                                    "-Wl,-rpath," + B.HERE]
    subprocess.run(cmd, check=True)
    print("built", LIB)


_lib = None


def load():
    global _lib
    if _lib is None:
        from ada_mvs_amd import _lib as product
        product.load()                                   # the experiment's library resolves its helpers against the product's
        _lib = ctypes.CDLL(LIB, mode=ctypes.RTLD_GLOBAL)
        _lib.adamvs_conv3x3_dd_wino24.restype = ctypes.c_int
        _lib.adamvs_conv3x3_dd_wino24.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    return _lib


def conv3x3_dd_wino24(x_cl, wino_layer, bias, skip, N, D, h, w, relu, out=None):
    """A stride-1 CostRegNet2D layer in the F(2x4, 3x3) form; wino_layer from pack_reg_layer_wino24."""
    from ada_mvs_amd import _lib as product
    out = torch.empty(N, h * w, D, device=x_cl.device) if out is None else out
    rc = load().adamvs_conv3x3_dd_wino24(x_cl.data_ptr(), wino_layer.data_ptr(), bias.data_ptr(), skip.data_ptr() if skip is not None else None,
                                         out.data_ptr(), N, D, h, w, int(relu), torch.cuda.current_stream().cuda_stream)
    product.check(rc, "conv3x3_dd_wino24")
    return out


WINO_G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                        [0, 0, 1]], dtype=torch.float64)


def pack_reg_layer_wino24(w, scale):
    """A stride-1 CostRegNet2D layer for the F(2x4, 3x3) kernel (csrc/costreg2d_wino24.hip): U = G w G4^T, patch rows i by F(2, 3),
    patch columns j by F(4, 3), formed in double precision and rounded once; A fragments [D/4][4 = i][D/16][64][4 = j 0..3]
    followed by [D/4][4][D/16][64][2 = j 4, 5] (include/adamvs_hip.h)."""
    w = (w.detach().to(torch.float64).cpu() * scale.detach().to(torch.float64).cpu().reshape(-1, 1, 1, 1))   # [co][ci][3][3]
    d = w.shape[0]
    assert w.shape[1] == d and d % 16 == 0
    u = torch.einsum("ik,ockl,jl->ijoc", WINO_G, w, WINO_G4).to(torch.float32)                              # [i 4][j 6][co][ci]
    # (i, j, tile, co16, kc, k4) -> (kc, i, tile, k4, co16, j): lane = k4*16 + co16
    f = u.reshape(4, 6, d // 16, 16, d // 4, 4).permute(4, 0, 2, 5, 3, 1).contiguous()
    return torch.cat([f[..., :4].reshape(-1), f[..., 4:].reshape(-1)])




def check_packing():
    """pack_reg_layer_wino24: U = G w G4^T in the two-array fragment order of include/adamvs_hip.h; Y = At2[(U . V)]A4 with
    V = Bt2 d B4 from the PACKED arrays reproduces the 3x3 convolution (the arithmetic of csrc/costreg2d_wino24.hip, on the CPU)."""
    D, h, w = 64, 6, 8
    g = torch.Generator().manual_seed(6)
    wt = torch.randn(D, D, 3, 3, generator=g, dtype=torch.float64)
    scale = torch.rand(D, generator=g, dtype=torch.float64) + 0.5
    x = torch.randn(1, D, h, w, generator=g, dtype=torch.float64)
    pk = pack_reg_layer_wino24(wt.float(), scale.float())
    assert pk.numel() == 24 * D * D
    lo = pk[:16 * D * D].reshape(D // 4, 4, D // 16, 64, 4).double()      # [kc][i][tile][lane][j 0..3]
    hi = pk[16 * D * D:].reshape(D // 4, 4, D // 16, 64, 2).double()      # [kc][i][tile][lane][j 4, 5]
    frag = torch.cat([lo, hi], dim=-1)
    u = torch.zeros(4, 6, D, D, dtype=torch.float64)                        # [i][j][co][ci]
    for lane in range(64):
        co16, k4 = lane & 15, lane >> 4
        u[:, :, co16::16, k4::4] = frag[:, :, :, lane, :].permute(1, 3, 2, 0)
    Bt2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    Bt4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                        [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
    At4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    xp = torch.nn.functional.pad(x[0], (1, 1, 1, 1))
    out = torch.zeros(D, h, w, dtype=torch.float64)
    for ty in range(h // 2):
        for tx in range(w // 4):
            d = xp[:, 2 * ty:2 * ty + 4, 4 * tx:4 * tx + 6]                 # [ci][4][6]
            v = torch.einsum("ik,ckl,jl->ijc", Bt2, d, Bt4)
            m = torch.einsum("ijoc,ijc->ijo", u, v)
            out[:, 2 * ty:2 * ty + 2, 4 * tx:4 * tx + 4] = torch.einsum("ai,ijo,bj->oab", At2, m, At4)
    ref = torch.nn.functional.conv2d(x, wt * scale.reshape(-1, 1, 1, 1), padding=1)[0]
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-6         # U is rounded to fp32 once




GPU_CASES = [(1, 192, 4, 64, 1, False), (2, 192, 13, 45, 0, False), (2, 192, 7, 70, 1, True), (1, 192, 1, 1, 1, False), (1, 64, 8, 130, 1, False),
             (1, 128, 13, 33, 0, True), (1, 256, 6, 32, 0, False), (1, 384, 5, 66, 1, False), (3, 192, 24, 192, 1, False),
             (12, 192, 30, 128, 1, False), (40, 64, 24, 64, 0, True)]


def check_gpu():
    """The kernel against a float64 convolution and the direct kernel: full and ragged blocks of 4 x 64 pixels, a map smaller
    than a block, every supported width, more tiles than the persistent grid has workgroups."""
    from ada_mvs_amd import hip_ops
    dev = lambda t: t.cuda().contiguous()      # noqa: E731
    for N, D, h, w, relu, skip in GPU_CASES:
        g = torch.Generator().manual_seed(N * 1000 + D + h + w)
        x = torch.randn(N, D, h, w, generator=g)
        wt = torch.randn(D, D, 3, 3, generator=g) / (3 * D ** 0.5)
        scale, shift = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
        sk = torch.randn(N, h * w, D, generator=g) if skip else None
        ref = torch.nn.functional.conv2d(x.double(), (wt * scale.reshape(-1, 1, 1, 1)).double(), shift.double(), padding=1)
        ref = torch.relu(ref) if relu else ref
        if skip:
            ref = ref + sk.double().reshape(N, h, w, D).permute(0, 3, 1, 2)
        x_cl = dev(x.permute(0, 2, 3, 1).reshape(N, h * w, D).contiguous())
        pk = dev(packing.pack_reg_layer(wt, scale, shift, False))
        out = conv3x3_dd_wino24(x_cl, dev(pack_reg_layer_wino24(wt, scale)), dev(shift), dev(sk) if skip else None, N, D, h, w, relu)
        direct = hip_ops.conv3x3_dd(x_cl, pk[:9 * D * D], pk[9 * D * D:], dev(sk) if skip else None, N, D, h, w, 0, relu)
        back = lambda y: y.cpu().double().reshape(N, h, w, D).permute(0, 3, 1, 2)      # noqa: E731
        rel = lambda a, b: float((a - b).abs().mean() / b.abs().mean())               # noqa: E731
        e0, e1 = rel(back(out), ref), rel(back(out), back(direct))
        print("N=%d D=%d %dx%d relu=%d skip=%d: %.2e against float64, %.2e against the direct kernel" % (N, D, h, w, relu, skip, e0, e1))
        assert e0 < 3e-6 and e1 < 3e-6         # the column transform's factors 4, 5, 8: about twice F(2x2, 3x3)'s error


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    if "--check" in sys.argv:
        check_packing()
        print("packing ok")
    if "--check-gpu" in sys.argv:
        check_gpu()
