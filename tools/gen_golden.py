#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (build container only).

Imports /root/reference/models/{adamvs,module}.py (read-only, never copied),
with a harness-side no-op for the hard-coded `.cuda()` calls of the inference
path (models/adamvs.py:448-459), feeds it the seeded recipes of
ada-mvs_amd/synth.py and stores inputs + outputs as small fixtures.  The
fixtures are data; the reference itself never travels to the GPU box.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
torch.Tensor.cuda = lambda self, *a, **k: self          # CPU run of a .cuda()-hard-coded path

import ada_mvs_amd  # noqa: E402  (repo shim -> ada-mvs_amd/)
from ada_mvs_amd import synth  # noqa: E402
from models import adamvs as ref_adamvs  # noqa: E402   (the reference)
from models import module as ref_module  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def build_ref_model(cfg):
    c = synth.CONFIGS[cfg]
    m = ref_adamvs.Infer_AdaMVSNet(num_depth=c["num_depth"], ndepths=c["ndepths"],
                                   depth_intervals_ratio=synth.DEPTH_INTERVALS_RATIO[:len(c["ndepths"])],
                                   share_cr=False, cr_base_chs=[8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m.eval()
    return m


@torch.no_grad()
def op_level():
    g = torch.Generator().manual_seed(1)
    # ---- a3 warp: in-bounds rig + a wide-baseline rig with out-of-bounds taps
    B, C, h, w = 2, 8, 24, 40
    src = synth.smooth_features(B, C, h, w, seed=3)
    for tag, baseline in (("warp_inb", 8.0), ("warp_oob", 150.0)):
        proj = synth.rig_projections(3, 4 * h, 4 * w, batch=B, baseline=baseline)["stage1"]
        depth = 400 + 200 * torch.rand(B, 3, h, w, generator=g)
        out = ref_module.homo_warping_float(src, proj[:, 2], proj[:, 0], depth)
        save("op_" + tag, src=src, src_proj=proj[:, 2], ref_proj=proj[:, 0], depth=depth, out=out)
    # ---- a3 without its missing guard (module.py:549-553 divides by X2 whatever its sign): planes that pass BEHIND the
    # source camera for part of the image (X2 < 0: finite, mirrored coordinates) and a row on the camera's focal plane
    # (X2 == 0 exactly: inf, and 0/0 = NaN where X0 == 0 too).  ref_proj = I, so T = src_proj exactly; with
    # T[2] = (0, 1/64, 1, -(Y0 + 64)) and d = 64, X2 = y - Y0 in exact arithmetic at every row.
    B, C, h, w, Y0 = 1, 8, 16, 24, 6
    gb = torch.Generator().manual_seed(11)                   # its own generator: the fixtures below keep their draws
    src = synth.smooth_features(B, C, h, w, seed=5)
    sp = torch.eye(4)[None].clone()
    sp[0, 0, 0], sp[0, 0, 3] = 1.0 / 64, -12.0               # X0 = x - 12 at d = 64: zero at x = 12
    sp[0, 1, 1], sp[0, 1, 3] = 1.0 / 64, -8.0                # X1 = y - 8
    sp[0, 2, 1], sp[0, 2, 3] = 1.0 / 64, -(Y0 + 64.0)
    depth = torch.stack([torch.full((h, w), 64.0), 60 + 8 * torch.rand(h, w, generator=gb), torch.full((h, w), 128.0)])[None]
    depth[0, 1, 3] = 64.0                                    # plane 1: one more exact row of its own (X2 = 3 - 6 = -3: behind)
    out = ref_module.homo_warping_float(src, sp, torch.eye(4)[None], depth)
    assert bool(torch.isnan(out[0, :, 0, Y0]).all()) and bool(torch.isfinite(out[0, :, 0, :Y0]).all())
    save("op_warp_behind", src=src, src_proj=sp, ref_proj=torch.eye(4)[None], depth=depth, out=out, y0=Y0)
    # ---- a2 both branches
    dv = torch.tensor([[400.0, 600.0], [380.0, 640.0]])
    s1 = ref_module.get_depth_range_samples(dv, 12, 4 * 200 / 48, "cpu", torch.float32, [2, 6, 10])
    cur = 400 + 200 * torch.rand(2, 6, 10, generator=g)
    s2 = ref_module.get_depth_range_samples(cur, 8, 2 * 200 / 48, "cpu", torch.float32, [2, 6, 10])
    save("op_depth_samples", dv=dv, s1=s1, cur=cur, s2=s2, interval1=4 * 200 / 48, interval2=2 * 200 / 48)
    # ---- depth_regression (2-D and 4-D depth values) + softmax/max
    p = torch.softmax(torch.randn(2, 12, 6, 10, generator=g), 1)
    save("op_depth_regression", p=p, dv2=s1[:, :, 0, 0].contiguous(), out2=ref_module.depth_regression(p, s1[:, :, 0, 0]),
         dv4=s1, out4=ref_module.depth_regression(p, s1))
    # ---- a7 the two interpolate uses
    x = torch.rand(2, 1, 6, 10, generator=g)
    save("op_upsample2x", x=x, out=torch.nn.functional.interpolate(x, [12, 20], mode="bilinear", align_corners=False))


@torch.no_grad()
def net_level():
    """a5 CostRegNet2D, a9 ConvGRUCell + SliceCostRegNetRED for the 3 stage variants."""
    m = build_ref_model("tiny")           # D1 = 16
    g = torch.Generator().manual_seed(2)
    sd = {k: v for k, v in m.state_dict().items()}
    x = torch.randn(2, 16, 16, 24, generator=g) * 0.5
    save("net_costreg2d", x=x, out=m.DepthNet[0].reg(x))
    for k, (C, h, w) in enumerate(((32, 16, 24), (16, 16, 24), (8, 16, 24))):
        cost = torch.randn(2, C, h, w, generator=g) * 0.5
        s1 = torch.randn(2, 8, h, w, generator=g) * 0.5
        s2 = torch.randn(2, 16, h // 2, w // 2, generator=g) * 0.5
        reg, n1, n2 = m.DepthNet[k].reg_fuse(cost, s1, s2)
        save("net_slice_step%d" % k, cost=cost, state1=s1, state2=s2, reg=reg, new1=n1, new2=n2)
    xg = torch.randn(2, 8, 10, 12, generator=g)
    hg = torch.randn(2, 8, 10, 12, generator=g)
    save("net_gru_cell", x=xg, h=hg, out=m.DepthNet[0].reg_fuse.conv_gru1(xg, hg)[0])
    del sd


@torch.no_grad()
def end_to_end(cfg, with_inputs):
    m = build_ref_model(cfg)
    c = synth.CONFIGS[cfg]
    imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=0)
    feats = [m.feature(imgs[:, v]) for v in range(c["views"])]
    out = m(imgs, proj, dv)
    arrays = {}
    for s in range(len(c["ndepths"])):
        st = out["stage%d" % (s + 1)]
        arrays["s%d_depth" % (s + 1)] = st["depth"]
        arrays["s%d_conf" % (s + 1)] = st["photometric_confidence"]
        S = c["views"] - 1
        for i in range(S):
            arrays["s%d_pairconf%d" % (s + 1, i)] = st["pair_confidence"][i]
        for i, pr in enumerate(st["pair_result"]):
            arrays["s%d_pairdepth%d" % (s + 1, i)] = pr
        arrays["s%d_n_pairconf" % (s + 1)] = len(st["pair_confidence"])
    arrays["depth"] = out["depth"]
    arrays["photometric_confidence"] = out["photometric_confidence"]
    if with_inputs:
        arrays["imgs"] = imgs
        arrays["depth_values"] = dv
        for k, v in proj.items():
            arrays["proj_" + k] = v
        for s in (1, 2, 3):
            arrays["feat_stage%d" % s] = torch.stack([f["stage%d" % s] for f in feats], 1)  # [B,V,C,h,w]
    save("e2e_" + cfg, **arrays)

    if with_inputs:
        # second oracle: the vectorised train/test twin in eval mode
        t = ref_adamvs.AdaMVSNet(ndepths=c["ndepths"], depth_intervals_ratio=synth.DEPTH_INTERVALS_RATIO)
        t.load_state_dict(m.state_dict())
        t.eval()
        # its depth_values are [min, max, interval] (adamvs.py:344-347)
        dv3 = torch.cat([dv, (dv[:, 1:2] - dv[:, 0:1]) / c["num_depth"]], 1)
        o2 = t(imgs, proj, dv3)
        save("e2e_" + cfg + "_twin", depth=o2["depth"], photometric_confidence=o2["photometric_confidence"])


@torch.no_grad()
def two_ranges():
    """Quirk Q4: a batch whose items have different depth ranges -- the interval comes from item 0 (adamvs.py:569-571), the
    stage-1 planes from each item's own [min, max].  Inputs: synth.tile_inputs("tiny", batch=2, seed=3) with these ranges."""
    m = build_ref_model("tiny")
    imgs, proj, _ = synth.tile_inputs("tiny", batch=2, seed=3)
    dv = torch.tensor([[400.0, 600.0], [450.0, 640.0]])
    r = m(imgs, proj, dv)
    save("e2e_tiny_two_ranges", depth_values=dv, depth=r["depth"], photometric_confidence=r["photometric_confidence"],
         s1_depth=r["stage1"]["depth"], s2_depth=r["stage2"]["depth"])


def behind_rig(cfg, batch=1):
    """synth's rig with source view 2 turned by 0.3 rad and moved 480 forward: X2 = a2 . d - 480 with a2 in [0.83, 1.08] over
    the image, so for d in [400, 600] every hypothesis plane crosses that camera's focal plane somewhere in the image --
    behind it on one side (mirrored coordinates), in front on the other (module.py:549-553 has no guard)."""
    import math
    c = synth.CONFIGS[cfg]
    proj = synth.rig_projections(c["views"], c["H"], c["W"], batch=batch)
    H, W = c["H"], c["W"]
    K = np.array([[1.2 * W, 0, W / 2.0], [0, 1.2 * W, H / 2.0], [0, 0, 1.0]])
    a = 0.3
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    P = np.eye(4)
    P[:3, :3], P[:3, 3] = K @ R, K @ np.array([-16.0, 0.0, -480.0])
    for name, s in (("stage1", 4.0), ("stage2", 2.0), ("stage3", 1.0)):
        m = P.copy()
        m[:2] /= s
        proj[name][:, 2] = torch.from_numpy(m.astype(np.float32))
    return proj


@torch.no_grad()
def end_to_end_behind():
    """The whole cascade of the reference on the rig above (inputs: synth.tile_inputs("tiny", 1, seed=0) images, behind_rig)."""
    m = build_ref_model("tiny")
    imgs, _, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    proj = behind_rig("tiny")
    r = m(imgs, proj, dv)
    arrays = {"proj_" + k: v for k, v in proj.items()}
    for s in (1, 2, 3):
        arrays["s%d_depth" % s] = r["stage%d" % s]["depth"]
        arrays["s%d_conf" % s] = r["stage%d" % s]["photometric_confidence"]
    for i in range(2):
        arrays["s1_pairconf%d" % i] = r["stage1"]["pair_confidence"][i]
        arrays["s1_pairdepth%d" % i] = r["stage1"]["pair_result"][i]
    # how much of the image the near / far plane puts behind view 2 at stage 1 (recorded for the test's own sanity check)
    T = torch.matmul(proj["stage1"][0, 2], torch.inverse(proj["stage1"][0, 0]))
    h, w = 16, 24
    y, x = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    a2 = T[2, 0] * x + T[2, 1] * y + T[2, 2]
    arrays["behind_fraction"] = torch.tensor([float((a2 * d + T[2, 3] < 0).float().mean()) for d in (400.0, 500.0, 600.0)])
    assert all(bool(torch.isfinite(v).all()) for v in arrays.values())
    save("e2e_tiny_behind", **arrays)


SHARP_CASES = (        # (fixture, ndepths, recipe): synth.tile_inputs(dict(views=3, H=64, W=96, ...), batch=1, seed=0)
    ("e2e_tiny_sharp", [40, 8, 4], "sharp"),           # D1 = 40: CostRegNet2D zero-padded to the 48-channel tiling, -1e30 pad scores
    ("e2e_tiny_sharp64", [64, 8, 4], "sharp"),         # D1 = 64: the F(2x2, 3x3) `prob` with per-lane softmax partials + merge
    ("e2e_tiny_overflow", [64], "overflow"),           # one stage: exp(reg_cost) = inf at part of the image (adamvs.py:516-531)
)


@torch.no_grad()
def end_to_end_sharp():
    """The reference on weights with a trained network's dynamic range (synth.LOGIT_GAINS): near one-hot stage-1 softmaxes,
    reg_cost up to +-60, and one case in which the unstabilised exp of adamvs.py:516 overflows: inf / NaN maps, recorded as
    the reference returns them."""
    for name, nd, recipe in SHARP_CASES:
        cfg = dict(views=3, H=64, W=96, ndepths=nd, num_depth=nd[0])
        m = ref_adamvs.Infer_AdaMVSNet(num_depth=nd[0], ndepths=nd, depth_intervals_ratio=synth.DEPTH_INTERVALS_RATIO[:len(nd)],
                                       share_cr=False, cr_base_chs=[8, 8, 8])
        m.load_state_dict(synth.seeded_state_dict(m, seed=0, recipe=recipe))
        m.eval()
        lo, hi = [1e30] * 3, [-1e30] * 3
        for i, dn in enumerate(m.DepthNet):          # the range of reg_cost per stage, recorded with the fixture
            def hook(mod, inp, out, i=i):
                r = out[0][torch.isfinite(out[0])]
                if r.numel():
                    lo[i], hi[i] = min(lo[i], float(r.min())), max(hi[i], float(r.max()))
            dn.reg_fuse.register_forward_hook(hook)
        imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=0)
        r = m(imgs, proj, dv)
        arrays = {"ndepths": np.asarray(nd), "reg_cost_min": np.asarray(lo[:len(nd)]), "reg_cost_max": np.asarray(hi[:len(nd)])}
        for s in range(1, len(nd) + 1):
            arrays["s%d_depth" % s] = r["stage%d" % s]["depth"]
            arrays["s%d_conf" % s] = r["stage%d" % s]["photometric_confidence"]
        for i in range(2):
            arrays["s1_pairconf%d" % i] = r["stage1"]["pair_confidence"][i]
            arrays["s1_pairdepth%d" % i] = r["stage1"]["pair_result"][i]
        finite = all(bool(torch.isfinite(v).all()) for v in arrays.values() if torch.is_tensor(v))
        assert finite == (recipe != "overflow"), (name, finite)
        print("%-20s reg_cost per stage: %s .. %s; mean pair confidence %.3f; non-finite share of the final depth map %.4f" % (
            name, ["%.1f" % v for v in lo[:len(nd)]], ["%.1f" % v for v in hi[:len(nd)]], float(arrays["s1_pairconf0"].mean()),
            float((~torch.isfinite(r["depth"])).float().mean())))
        save(name, **arrays)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--only-sharp" in sys.argv:
        end_to_end_sharp()
        sys.exit(0)
    if "--only-two-ranges" in sys.argv:
        two_ranges()
        sys.exit(0)
    if "--only-behind" in sys.argv:
        end_to_end_behind()
        sys.exit(0)
    op_level()
    net_level()
    end_to_end("tiny", with_inputs=True)
    end_to_end("cfg1", with_inputs=False)
    two_ranges()
    end_to_end_behind()
    end_to_end_sharp()
