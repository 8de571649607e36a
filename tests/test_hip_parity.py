"""GPU parity tests: every C-ABI entry point against the CPU oracle and against
fixtures generated from the real reference (tests/golden/).

Bar (BASELINE.json north_star): depth / probability maps within 1e-3 relative L1
of the reference CPU path.  The asserts below use tighter, per-op tolerances
(fp32 kernels agree to ~1e-5) so that a real bug cannot hide under the bar.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l1
import ada_mvs_amd  # noqa: F401
from ada_mvs_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NORTH_STAR_TOL = 1e-3      # stated bar
OP_TOL = 5e-5              # what fp32 kernels should reach per op
E2E_TOL = 3e-4             # whole cascade (GRU recurrence amplifies rounding)


@pytest.fixture(scope="module")
def hip():
    from ada_mvs_amd import hip_ops, _lib
    _lib.load()
    assert torch.cuda.is_available()
    return hip_ops


@pytest.fixture(scope="module")
def O():
    from oracle import adamvs_oracle
    return adamvs_oracle


def dev(t):
    return t.cuda().contiguous()


def _model(cfg):
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS[cfg]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO[:len(c["ndepths"])], False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


# --------------------------------------------------------------------------- geometry
def test_relative_transforms(hip, O):
    proj = synth.rig_projections(5, 384, 768, batch=3)["stage1"]
    rt = hip.relative_transforms(dev(proj)).cpu()
    for s in range(4):
        R, t = O.relative_transform(proj[:, s + 1], proj[:, 0])
        assert torch.allclose(rt[:, s, :9].reshape(3, 3, 3), R, rtol=1e-4, atol=1e-5)
        assert torch.allclose(rt[:, s, 9:], t, rtol=1e-4, atol=1e-3)


def test_homo_warping_float_golden(hip):
    from ada_mvs_amd.models.module import homo_warping_float
    for name in ("op_warp_inb", "op_warp_oob"):
        g = load_golden(name)
        out = homo_warping_float(dev(g["src"]), dev(g["src_proj"]), dev(g["ref_proj"]), dev(g["depth"]))
        assert out.shape == g["out"].shape
        assert rel_l1(out, g["out"]) < OP_TOL, name
        # out-of-bounds pixels are exactly zero where the reference's are
        zero_ref = g["out"].abs().sum(1) == 0
        assert float(out.cpu().abs().sum(1)[zero_ref].max() if zero_ref.any() else 0.0) < 1e-4


def test_depth_range_samples_bit_exact_arithmetic(hip):
    """Hypothesis planes in the reference's fp32 operation order (module.py:632-641, 651-656): step = (hi - lo) / (D - 1),
    plane d = lo + d * step with a rounded product and a rounded sum -- emulated in numpy float32, compared bit for bit."""
    import numpy as np
    rng = np.random.RandomState(0)
    for D in (12, 48, 192, 256):
        lo = (350 + 100 * rng.rand(3)).astype(np.float32)
        hi = (lo + 150 + 100 * rng.rand(3)).astype(np.float32)
        step = ((hi - lo) / np.float32(D - 1)).astype(np.float32)
        want = (lo[:, None] + (np.arange(D, dtype=np.float32)[None] * step[:, None]).astype(np.float32)).astype(np.float32)
        got = hip.depth_range_samples(dev(torch.from_numpy(np.stack((lo, hi), 1))), D, 0.0, [3, 2, 4]).cpu().numpy()
        assert np.array_equal(got, np.broadcast_to(want[:, :, None, None], got.shape)), D
        cur = (400 + 200 * rng.rand(3, 2, 4)).astype(np.float32)
        interval = 2.0 * 200.0 / 48                                   # a Python float, as in the reference's driver
        half = np.float32(D / 2 * interval)
        wlo, whi = (cur - half).astype(np.float32), (cur + half).astype(np.float32)
        wstep = ((whi - wlo) / np.float32(D - 1)).astype(np.float32)
        wwant = wlo[:, None] + (np.arange(D, dtype=np.float32)[None, :, None, None] * wstep[:, None]).astype(np.float32)
        wgot = hip.depth_range_samples(dev(torch.from_numpy(cur)), D, interval, [3, 2, 4]).cpu().numpy()
        assert np.array_equal(wgot, wwant.astype(np.float32)), D


def test_depth_range_samples_golden(hip):
    from ada_mvs_amd.models.module import get_depth_range_samples
    g = load_golden("op_depth_samples")
    s1 = get_depth_range_samples(dev(g["dv"]), 12, g["interval1"], "cuda", torch.float32, [2, 6, 10])
    s2 = get_depth_range_samples(dev(g["cur"]), 8, g["interval2"], "cuda", torch.float32, [2, 6, 10])
    assert torch.allclose(s1.cpu(), g["s1"], rtol=0, atol=1e-4)
    assert torch.allclose(s2.cpu(), g["s2"], rtol=0, atol=1e-4)


def test_depth_regression_and_resize_golden(hip):
    from ada_mvs_amd.models.module import depth_regression
    g = load_golden("op_depth_regression")
    assert rel_l1(depth_regression(dev(g["p"]), dev(g["dv2"])), g["out2"]) < 1e-6
    assert rel_l1(depth_regression(dev(g["p"]), dev(g["dv4"])), g["out4"]) < 1e-6
    u = load_golden("op_upsample2x")
    assert torch.allclose(hip.resize_bilinear(dev(u["x"]), (12, 20)).cpu(), u["out"], atol=1e-6)


def test_pack_unpack_roundtrip(hip):
    x = torch.randn(3, 16, 10, 14)
    cl = hip.pack_features(dev(x))
    assert torch.equal(cl.cpu(), x.permute(0, 2, 3, 1).reshape(3, 140, 16))
    assert torch.equal(hip.unpack_features(cl, 10, 14).cpu(), x)


# --------------------------------------------------------------------------- FeatureNet0
@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (3, 128, 160), (1, 96, 224)])
def test_feature_net0_against_oracle(hip, O, N, H, W):
    """adamvs_feature_net0 (MFMA convolutions, folded BatchNorm, context branches applied at pooled resolution)
    against the CPU restatement of FeatureNet0.forward, stage by stage; also the NCHW dict of the mirror's forward()."""
    from ada_mvs_amd.models.adamvs import FeatureNet0
    net = FeatureNet0(8)
    sd = synth.seeded_state_dict(net, seed=2)
    net.load_state_dict(sd)
    x = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(H))
    ref = O.feature_net(x, sd, "")
    net = net.cuda().eval()
    assert net.hip_supported(dev(x))
    maps = net.forward_cl(dev(x))
    for k, scale in enumerate((4, 2, 1)):
        got = hip.unpack_features(maps[k], H // scale, W // scale)
        assert got.shape == ref["stage%d" % (k + 1)].shape
        assert rel_l1(got, ref["stage%d" % (k + 1)]) < OP_TOL, "stage%d" % (k + 1)
    out = net(dev(x))
    assert rel_l1(out["stage2"], ref["stage2"]) < OP_TOL
    net.workspace_limit_bytes = hip.feature_net0_workspace_bytes(1, H, W)          # one image per call: same maps
    chunked = net.forward_cl(dev(x))
    assert all(torch.equal(a, b) for a, b in zip(chunked, maps))


def test_feature_net0_in_chunks_equals_one_call(hip):
    """forward_cl bounds its workspace by running large batches in chunks that write into slices of the whole maps."""
    m, _ = _model("tiny")
    x = torch.randn(5, 3, 64, 96, generator=torch.Generator().manual_seed(8)).cuda()
    with torch.no_grad():
        whole = m.feature.forward_cl(x)
        per_image = hip.feature_net0_workspace_bytes(1, 64, 96)
        limit = m.feature.workspace_limit_bytes
        m.feature.workspace_limit_bytes = 2 * per_image          # chunks of 2, 2, 1 images
        try:
            parts = m.feature.forward_cl(x)
        finally:
            m.feature.workspace_limit_bytes = limit
    for a, b in zip(whole, parts):
        assert torch.equal(a, b)
    with pytest.raises(Exception, match="out tensors"):
        hip.feature_net0(x[:1], m.feature.packed(x.device), out=tuple(t[:1, :, :4] for t in whole))


def test_feature_net0_has_no_fallback(hip):
    """forward() and forward_cl() are the HIP kernels or an error: CPU tensors and sizes that are not multiples of 32 (the
    reference's own rule, quirk Q9) raise instead of running the layers through PyTorch."""
    from ada_mvs_amd._lib import AdaMVSHipError
    m, _ = _model("tiny")
    with torch.no_grad():
        for x in (torch.zeros(1, 3, 64, 96), torch.zeros(1, 3, 40, 96).cuda()):
            with pytest.raises(AdaMVSHipError, match="no fallback"):
                m.feature(x)
            with pytest.raises(AdaMVSHipError, match="no fallback"):
                m.feature.forward_cl(x)


def test_feature_net0_golden(hip):
    """Against the reference's own FeatureNet0 outputs (end-to-end fixture, 64x96, 3 views)."""
    g = load_golden("e2e_tiny")
    m, _ = _model("tiny")
    imgs = g["imgs"]
    B, V = imgs.shape[:2]
    feats_cl, shapes = m.extract_features(dev(imgs))
    for k in range(3):
        ref = g["feat_stage%d" % (k + 1)]                       # [B][V][C][h][w]
        got = hip.unpack_features(feats_cl[k], shapes[k][2], shapes[k][3]).reshape(V, B, *ref.shape[2:]).transpose(0, 1)
        assert rel_l1(got, ref) < OP_TOL, "stage%d" % (k + 1)


# --------------------------------------------------------------------------- pass A
@pytest.mark.parametrize("C", [32, 16, 8])
def test_pair_similarity(hip, O, C):
    B, S, D, h, w = 2, 2, 12, 24, 40
    feats = [synth.smooth_features(B, C, h, w, seed=v) for v in range(S + 1)]
    proj = synth.rig_projections(S + 1, 4 * h, 4 * w, batch=B, baseline=60.0)["stage1"]
    planes = O.depth_range_samples(torch.tensor([[400.0, 600.0]] * B), D, 0.0, [B, h, w])
    feat_cl = hip.pack_features(dev(torch.stack(feats, 0).reshape(-1, C, h, w)))
    rt = hip.relative_transforms(dev(proj))
    sim = hip.pair_similarity(feat_cl, rt, dev(planes), B, S, C, D, h, w).cpu().reshape(S, B, h, w, D)
    for s in range(S):
        R, t = O.relative_transform(proj[:, s + 1], proj[:, 0])
        ref = O.pair_similarity_volume(feats[0], feats[s + 1], R, t, planes)        # [B,D,h,w]
        assert rel_l1(sim[s].permute(0, 3, 1, 2), ref) < OP_TOL


def test_cost_reg_net_2d_golden(hip):
    g = load_golden("net_costreg2d")
    m, _ = _model("tiny")
    out = m.DepthNet[0].reg(dev(g["x"]))
    assert rel_l1(out, g["out"]) < OP_TOL


@pytest.mark.parametrize("D,h,w", [(48, 16, 24), (192, 8, 16), (32, 8, 8), (64, 8, 8), (96, 8, 8), (128, 8, 8), (256, 8, 8)])
def test_cost_reg_net_2d_widths(hip, O, D, h, w):
    from ada_mvs_amd.models.adamvs import CostRegNet2D
    net = CostRegNet2D(D)
    sd = synth.seeded_state_dict(net, seed=1)
    net.load_state_dict(sd)
    x = torch.randn(2, D, h, w, generator=torch.Generator().manual_seed(D)) * 0.5
    ref = O.cost_reg_net_2d(x, sd, "")
    out = net.cuda()(dev(x))
    assert rel_l1(out, ref) < OP_TOL


@pytest.mark.parametrize("N,D,h,w,relu,skip", [(1, 192, 6, 32, 1, False), (2, 192, 13, 45, 0, False), (2, 192, 7, 70, 1, True),
                                               (1, 192, 1, 1, 1, False), (1, 64, 8, 40, 1, False), (1, 128, 13, 33, 0, True),
                                               (1, 256, 6, 32, 0, False), (3, 192, 24, 48, 1, False),
                                               (12, 192, 30, 64, 1, False), (40, 64, 24, 64, 0, True),
                                               (1, 512, 7, 34, 1, True)])
def test_conv3x3_dd_winograd(hip, N, D, h, w, relu, skip):
    """A stride-1 CostRegNet2D layer in the F(2x2, 3x3) form (csrc/costreg2d_wino.hip) against a float64 convolution
    (ConvBnReLU.forward, reference models/module.py:254-261, BN folded) and against the direct kernel: full and ragged
    blocks of 6 x 32 pixels, maps smaller than one block, every supported width; the last two cases have more tiles (360, 320)
    than the persistent grid has workgroups (256): a workgroup walks several, with the next tile's requests in flight over the
    epilogue of the current one."""
    from ada_mvs_amd import packing
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
    out = hip.conv3x3_dd_wino(x_cl, dev(packing.pack_reg_layer_wino(wt, scale)), dev(shift), dev(sk) if skip else None, N, D, h, w, relu)
    direct = hip.conv3x3_dd(x_cl, pk[:9 * D * D], pk[9 * D * D:], dev(sk) if skip else None, N, D, h, w, 0, relu)
    back = lambda y: y.cpu().double().reshape(N, h, w, D).permute(0, 3, 1, 2)
    assert rel_l1(back(out), ref) < 2e-6                    # measured 2e-7 ... 5e-7 (the direct kernel: 2e-7 ... 7e-7)
    assert rel_l1(back(out), back(direct)) < 2e-6


def test_winograd_one_and_two_workgroups_per_cu_give_the_same_bits(hip, set_option):
    """csrc/costreg2d_wino.hip has a 6 x 32-pixel form (one workgroup per CU) and a 4 x 32 form (two); launch_conv_wino picks by map
    size.  Option wino_wps forces one: the same layer through both, bit for bit."""
    from ada_mvs_amd import packing
    g = torch.Generator().manual_seed(3)
    N, D, h, w = 3, 192, 26, 70
    x = dev(torch.randn(N, h * w, D, generator=g))
    wt = torch.randn(D, D, 3, 3, generator=g) / 72
    outs = []
    for wps in (1, 2):
        set_option("wino_wps", wps)
        outs.append(hip.conv3x3_dd_wino(x, dev(packing.pack_reg_layer_wino(wt, torch.ones(D))), dev(torch.zeros(D)), None, N, D, h, w, 1).cpu())
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0


@pytest.mark.parametrize("switch", ["winograd", "wino_softmax"])
def test_cost_reg_net_2d_direct_kernels_behind_their_options(hip, O, set_option, switch):
    """The A/B options of CostRegNet2D.  winograd = 0: its stride-1 layers on the direct kernel at the widths the F(2x2, 3x3) kernel
    otherwise takes, the softmax epilogue of the direct `prob` layer included.  wino_softmax = 0: the F(2x2, 3x3) `prob` layer writes
    its scores and k_softmax_regress reads them, as before round 4's partials + merge (test_piecewise_phase_masks runs stage 1 at
    D = 192 through it).  The tests named below run again under the option."""
    set_option(switch, 0)
    import inspect
    me = sys.modules[__name__]
    ran = 0
    for name in ("test_cost_reg_net_2d_widths", "test_prob_softmax_regress_fused", "test_generated_planes_equal_materialised_planes",
                 "test_piecewise_phase_masks"):
        fn = getattr(me, name)
        marks = [mk for mk in getattr(fn, "pytestmark", []) if mk.name == "parametrize"]
        names = inspect.signature(fn).parameters
        cases = [{}]
        for mk in marks:
            keys = [k.strip() for k in mk.args[0].split(",")]
            cases = [dict(c, **dict(zip(keys, v if len(keys) > 1 else (v,)))) for c in cases for v in mk.args[1]]
        for c in cases:
            fixtures = {"hip": hip, "O": O}
            fn(**{k: (c[k] if k in c else fixtures[k]) for k in names})
            ran += 1
    assert ran >= 8


@pytest.mark.parametrize("D,h,w", [(32, 16, 24), (64, 8, 16), (192, 8, 16), (256, 8, 8)])
def test_cost_reg_net_2d_bf16x3(hip, O, D, h, w):
    """Split-bf16 MFMA path (three bf16 MFMAs per product, fp32 accumulate) against the fp32 oracle."""
    from ada_mvs_amd.models.adamvs import CostRegNet2D
    net = CostRegNet2D(D)
    sd = synth.seeded_state_dict(net, seed=1)
    net.load_state_dict(sd)
    net.precision = "bf16x3"
    x = torch.randn(2, D, h, w, generator=torch.Generator().manual_seed(D)) * 0.5
    ref = O.cost_reg_net_2d(x, sd, "")
    out = net.cuda()(dev(x))
    err = rel_l1(out, ref)
    assert err < 2e-4, err            # 16 significant bits per operand, 11 layers deep; fp32 path: < 5e-5


def test_end_to_end_bf16x3(hip, O):
    """Whole cascade with precision='bf16x3' (D1 = 32 so that the split-bf16 kernels are actually used)."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    cfg = dict(views=3, H=64, W=96, ndepths=[32, 8, 4], num_depth=32)
    m = Infer_AdaMVSNet(cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision="bf16x3")
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    assert m.DepthNet[0].reg.effective_precision() == "bf16x3" and m.DepthNet[2].reg_fuse.precision == "bf16x3"
    imgs, proj, dv = synth.tile_inputs(cfg, batch=2, seed=4)
    with torch.no_grad():
        out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
        ref = O.infer_adamvs_forward(imgs, proj, dv, sd, cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    for key in ("depth", "photometric_confidence"):
        assert rel_l1(out[key], ref[key]) < NORTH_STAR_TOL / 2, key
    assert rel_l1(out["stage1"]["pair_confidence"][0], ref["stage1"]["pair_confidence"][0]) < NORTH_STAR_TOL / 2


def test_softmax_max_regress(hip, O):
    S, B, D, h, w = 2, 2, 48, 6, 10
    g = torch.Generator().manual_seed(5)
    score = torch.randn(S * B, D, h, w, generator=g) * 3
    planes = O.depth_range_samples(torch.tensor([[400.0, 600.0], [380.0, 650.0]]), D, 0.0, [B, h, w])
    vw, pd = hip.softmax_max_regress(hip.pack_features(dev(score)), dev(planes), S, B, D, h, w)
    for s in range(S):
        rvw, rpd = O.softmax_max_regress(score[s * B:(s + 1) * B], planes)
        assert rel_l1(vw[s], rvw[:, 0]) < 1e-5 and rel_l1(pd[s], rpd) < 1e-5


# --------------------------------------------------------------------------- pass B
@pytest.mark.parametrize("k", [0, 1, 2])
def test_slice_reg_step_golden(hip, k):
    g = load_golden("net_slice_step%d" % k)
    m, _ = _model("tiny")
    reg, n1, n2 = m.DepthNet[k].reg_fuse(dev(g["cost"]), dev(g["state1"]), dev(g["state2"]))
    assert reg.shape == g["reg"].shape
    assert rel_l1(n1, g["new1"]) < OP_TOL, "GRU level 1"
    assert rel_l1(n2, g["new2"]) < OP_TOL, "GRU level 2"
    assert rel_l1(reg, g["reg"]) < OP_TOL, "decoder"


@pytest.mark.parametrize("k", [0, 1, 2])
def test_slice_reg_step_bf16x3(hip, k):
    """One recurrent step with the split-bf16 convolutions against the reference fixture (fp32)."""
    g = load_golden("net_slice_step%d" % k)
    m, _ = _model("tiny")
    net = m.DepthNet[k].reg_fuse
    net.precision = "bf16x3"
    reg, n1, n2 = net(dev(g["cost"]), dev(g["state1"]), dev(g["state2"]))
    assert rel_l1(n1, g["new1"]) < 2e-4, "GRU level 1"
    assert rel_l1(n2, g["new2"]) < 2e-4, "GRU level 2"
    assert rel_l1(reg, g["reg"]) < 2e-4, "decoder"


@pytest.mark.parametrize("k,h,w", [(0, 22, 38), (1, 30, 18), (2, 26, 50), (0, 4, 6)])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_slice_reg_step_ragged_shapes(hip, O, precision, k, h, w):
    """Sizes that are no multiple of any tile (4 x 16 GRU tiles, 6 x 30 decoder tiles, 8 x 16 conv1 tiles): every
    kernel takes its edge path on some tiles and its interior path on others."""
    m, sd = _model("tiny")
    net = m.DepthNet[k].reg_fuse
    net.precision = precision
    B, C = 3, net.in_channels
    g = torch.Generator().manual_seed(100 * k + h)
    cost = torch.randn(B, C, h, w, generator=g)
    s1 = torch.randn(B, 8, h, w, generator=g) * 0.5
    s2 = torch.randn(B, 16, h // 2, w // 2, generator=g) * 0.5
    ref, r1, r2 = O.slice_reg_step(cost, s1, s2, sd, "DepthNet.%d.reg_fuse." % k, in_up=(k < 2))
    reg, n1, n2 = net(dev(cost), dev(s1), dev(s2))
    tol = OP_TOL if precision == "fp32" else 2e-4
    assert reg.shape == ref.shape
    assert rel_l1(n1, r1) < tol and rel_l1(n2, r2) < tol and rel_l1(reg, ref) < tol


@pytest.mark.parametrize("k", [0, 1, 2])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_slice_reg_step_many_tiles(hip, precision, k):
    """Persistent kernels past their resident capacity (every workgroup walks several tiles, interior and edge),
    for the three stage variants (C = 32 / 16 / 8 conv1, transposed and flat last layer): a batch of identical tiles
    must give, entry by entry, the single-tile result (itself pinned by the fixtures)."""
    m, _ = _model("tiny")
    net = m.DepthNet[k].reg_fuse
    net.precision = precision
    C, h, w, B = net.in_channels, 64, 96, 80
    g = torch.Generator().manual_seed(11 + k)
    cost = torch.randn(1, C, h, w, generator=g)
    s1 = torch.randn(1, 8, h, w, generator=g) * 0.5
    s2 = torch.randn(1, 16, h // 2, w // 2, generator=g) * 0.5
    reg1, a1, b1 = net(dev(cost), dev(s1), dev(s2))
    regB, aB, bB = net(dev(cost).expand(B, -1, -1, -1).contiguous(), dev(s1).expand(B, -1, -1, -1).contiguous(),
                       dev(s2).expand(B, -1, -1, -1).contiguous())
    for one, many, name in ((reg1, regB, "decoder"), (a1, aB, "GRU level 1"), (b1, bB, "GRU level 2")):
        assert torch.equal(many, one.expand_as(many)), name


@pytest.mark.parametrize("C,h,w,D,baseline", [(32, 16, 40, 3, 80.0), (16, 20, 36, 3, 80.0), (8, 24, 70, 3, 80.0),
                                              (32, 24, 40, 24, 8.0),        # narrow sweep: one LDS-resident chunk
                                              (32, 24, 40, 16, 400.0),      # wide sweep: chunks split, patches leave the image
                                              (16, 40, 24, 10, 2500.0)])    # extreme: per-plane patches / global fallback
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_aggregate_conv1(hip, O, C, h, w, D, baseline, precision):
    """Weighted aggregation (register-resident taps, arbitrary planes) + two-row conv1 (fp32 and split-bf16 MFMA)."""
    import torch.nn.functional as F
    B, S = 2, 3
    feats = [synth.smooth_features(B, C, h, w, seed=10 + v) for v in range(S + 1)]
    proj = synth.rig_projections(S + 1, 4 * h, 4 * w, batch=B, baseline=baseline)["stage1"]
    g = torch.Generator().manual_seed(3)
    if D <= 3:
        planes = 400 + 200 * torch.rand(B, D, h, w, generator=g)            # independent random planes
    else:                                                                    # per-pixel monotone sweeps, like the cascade
        lo = 380 + 40 * torch.rand(B, 1, h, w, generator=g)
        step = (200 + 40 * torch.rand(B, 1, h, w, generator=g)) / (D - 1)
        planes = lo + step * torch.arange(D, dtype=torch.float32).reshape(1, D, 1, 1)
    vw = torch.rand(S, B, h, w, generator=g)
    w1 = torch.randn(8, C, 3, 3, generator=g) * 0.1
    from ada_mvs_amd import packing
    c1 = hip.aggregate_conv1(hip.pack_features(dev(torch.stack(feats, 0).reshape(-1, C, h, w))),
                             hip.relative_transforms(dev(proj)), dev(planes), dev(vw),
                             (packing.pack_conv1_two_row(w1) if precision == "fp32" else packing.pack_conv1_two_row_bf16x3(w1)).cuda(),
                             B, S, C, D, h, w, precision=0 if precision == "fp32" else 1).cpu()      # [D,B,hw,8]
    Rs, ts = zip(*[O.relative_transform(proj[:, s + 1], proj[:, 0]) for s in range(S)])
    zero_frac = 0.0
    for d in range(D):
        sim = O.aggregate_similarity(feats[0], feats[1:], Rs, ts, planes[:, d], [vw[s].unsqueeze(1) for s in range(S)])
        zero_frac += float((sim.abs().sum(1) == 0).float().mean()) / D
        ref = F.relu(F.conv2d(sim, w1, None, 1, 1))
        assert rel_l1(c1[d].reshape(B, h, w, 8).permute(0, 3, 1, 2), ref) < (OP_TOL if precision == "fp32" else 2e-4), "plane %d" % d
    if baseline >= 400.0:
        assert zero_frac > 0.02, "case must include pixels whose every view projects outside (%g)" % zero_frac


# --------------------------------------------------------------------------- stages / end to end
def _run_stages_from_golden_features(cfg, g):
    m, sd = _model(cfg)
    c = synth.CONFIGS[cfg]
    B = 1
    feats_cl, shapes = [], []
    from ada_mvs_amd import hip_ops
    for s in (1, 2, 3):
        f = g["feat_stage%d" % s]                       # [B,V,C,h,w]
        V, C, h, w = f.shape[1:]
        feats_cl.append(hip_ops.pack_features(dev(f.transpose(0, 1).reshape(V * B, C, h, w))))
        shapes.append((B, C, h, w))
    proj = {k[5:]: dev(v) for k, v in g.items() if k.startswith("proj_")}
    dv = g["depth_values"]
    interval = (float(dv[0, 1]) - float(dv[0, 0])) / c["num_depth"]
    return m.infer_from_features(feats_cl, shapes, proj, dev(dv), interval)


def _check_against_golden(cfg, g, out, tol):
    c = synth.CONFIGS[cfg]
    worst = 0.0
    for s in range(len(c["ndepths"])):
        st = out["stage%d" % (s + 1)]
        pairs = [(st["depth"], g["s%d_depth" % (s + 1)]), (st["photometric_confidence"], g["s%d_conf" % (s + 1)])]
        for i in range(c["views"] - 1):
            pairs.append((st["pair_confidence"][i], g["s%d_pairconf%d" % (s + 1, i)]))
        for i, pr in enumerate(st["pair_result"]):
            pairs.append((pr, g["s%d_pairdepth%d" % (s + 1, i)]))
        assert len(st["pair_confidence"]) == g["s%d_n_pairconf" % (s + 1)]       # quirk Q1 list lengths
        for a, b in pairs:
            assert a.shape == b.shape
            worst = max(worst, rel_l1(a, b))
    assert worst < tol, "worst relL1 %.3e" % worst
    assert rel_l1(out["depth"], g["depth"]) < tol
    assert rel_l1(out["photometric_confidence"], g["photometric_confidence"]) < tol
    return worst


def test_hot_path_from_reference_features_tiny(hip):
    """Hot path only: reference FeatureNet0 outputs in, reference maps out."""
    g = load_golden("e2e_tiny")
    with torch.no_grad():
        out = _run_stages_from_golden_features("tiny", g)
    worst = _check_against_golden("tiny", g, out, E2E_TOL)
    assert worst < NORTH_STAR_TOL


def test_drop_in_forward_tiny_and_cfg1(hip):
    """Infer_AdaMVSNet.forward(imgs, proj_matrices, depth_values) vs the reference's outputs."""
    for cfg in ("tiny", "cfg1"):
        g = load_golden("e2e_" + cfg)
        m, _ = _model(cfg)
        imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=0)
        with torch.no_grad():
            out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
        assert set(("depth", "photometric_confidence", "pair_confidence", "pair_result", "stage1", "stage2", "stage3")) <= set(out)
        _check_against_golden(cfg, g, out, NORTH_STAR_TOL)


def test_nine_views_end_to_end(hip, O):
    """BASELINE cfg5 has 8 source views: the S > 4 instantiation of the sweep, small size, against the oracle."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    cfg = dict(views=9, H=64, W=96, ndepths=[16, 8, 4], num_depth=16)
    m = Infer_AdaMVSNet(cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=9)
    with torch.no_grad():
        out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
        ref = O.infer_adamvs_forward(imgs, proj, dv, sd, cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    assert len(out["stage1"]["pair_result"]) == 8
    for key in ("depth", "photometric_confidence"):
        assert rel_l1(out[key], ref[key]) < E2E_TOL, key


def test_data_parallel_wrapper_and_module_prefixed_checkpoint(hip):
    """predict_whu.py:82-89 wraps the model in nn.DataParallel and loads 'module.'-prefixed keys."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS["tiny"]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = {"module." + k: v for k, v in synth.seeded_state_dict(m, seed=0).items()}
    dp = torch.nn.DataParallel(m).cuda()
    dp.load_state_dict(sd)
    dp.eval()
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    with torch.no_grad():
        out = dp(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
    g = load_golden("e2e_tiny")
    assert rel_l1(out["depth"], g["depth"]) < NORTH_STAR_TOL


def test_batch_of_tiles_matches_single_tiles(hip, O):
    """Batched tiles (the sharding unit) against the oracle run tile by tile, with different rigs per tile."""
    m, sd = _model("tiny")
    c = synth.CONFIGS["tiny"]
    imgs, proj, dv = synth.tile_inputs("tiny", batch=3, seed=7)
    with torch.no_grad():
        out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
        for b in range(3):
            ref = O.infer_adamvs_forward(imgs[b:b + 1], {k: v[b:b + 1] for k, v in proj.items()}, dv[b:b + 1], sd,
                                         c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO)
            assert rel_l1(out["depth"][b:b + 1], ref["depth"]) < E2E_TOL
            assert rel_l1(out["photometric_confidence"][b:b + 1], ref["photometric_confidence"]) < NORTH_STAR_TOL


def test_batch_items_with_different_depth_ranges(hip):
    """Quirk Q4: interval from batch item 0, stage-1 planes per item -- against the reference-run fixture."""
    g = load_golden("e2e_tiny_two_ranges")
    m, _ = _model("tiny")
    imgs, proj, _ = synth.tile_inputs("tiny", batch=2, seed=3)
    with torch.no_grad():
        out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(g["depth_values"]))
    for key in ("depth", "photometric_confidence"):
        assert rel_l1(out[key], g[key]) < E2E_TOL, key
    assert rel_l1(out["stage1"]["depth"], g["s1_depth"]) < E2E_TOL and rel_l1(out["stage2"]["depth"], g["s2_depth"]) < E2E_TOL


# --------------------------------------------------------------------------- full-size properties (cfg2 shapes)
def test_full_size_properties_cfg2(hip):
    """BASELINE cfg2 (5 views, 384x768, single stage, 192 hypotheses): too big for the oracle in seconds,
    so check size-independent properties: determinism, batch invariance, value ranges."""
    m, _ = _model("cfg2")
    imgs, proj, dv = synth.tile_inputs("cfg2", batch=2, seed=1)
    imgs[1] = imgs[0]
    proj = {k: torch.stack((v[0], v[0])) for k, v in proj.items()}
    with torch.no_grad():
        feats_cl, shapes = m.extract_features(dev(imgs))
        pj = {k: dev(v) for k, v in proj.items()}
        interval = 200.0 / 192
        a = m.infer_from_features(feats_cl, shapes, pj, dev(dv), interval)
        b = m.infer_from_features(feats_cl, shapes, pj, dev(dv), interval)
    d, p = a["depth"], a["photometric_confidence"]
    assert d.shape == (2, 192, 384) and p.shape == (2, 192, 384)          # quirk Q10: half resolution
    assert torch.equal(d, b["depth"]) and torch.equal(p, b["photometric_confidence"])      # deterministic
    assert torch.equal(d[0], d[1]) and torch.equal(p[0], p[1])                             # batch invariant
    assert bool(torch.isfinite(d).all()) and bool(torch.isfinite(p).all())
    assert float(d.min()) >= 400.0 - 1e-2 and float(d.max()) <= 600.0 + 1e-2               # convex combination of planes
    assert float(p.min()) > 0.0 and float(p.max()) <= 1.0 + 1e-6
    for vw in a["pair_confidence"][:4]:
        assert float(vw.min()) >= 1.0 / 192 - 1e-6 and float(vw.max()) <= 1.0 + 1e-6         # max of a softmax over 192


# --------------------------------------------------------------------------- error behaviour
def test_fails_loudly(hip):
    from ada_mvs_amd._lib import AdaMVSHipError
    m, _ = _model("tiny")
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    with pytest.raises(AdaMVSHipError):
        m(imgs, proj, dv)                                   # CPU tensors: no fallback
    with pytest.raises(AdaMVSHipError):
        hip.cost_reg_net_2d(torch.zeros(1, 64, 40, device="cuda"), torch.zeros(10, device="cuda"), 8, 8)   # D=40 unsupported
    with pytest.raises(AdaMVSHipError):
        hip.pack_features(torch.zeros(1, 6, 4, 4, device="cuda"))     # C % 4 != 0


def test_train_test_twin_golden(hip):
    """AdaMVSNet (the reference's train/test model, eval mode) against the reference's own output for it
    (tools/gen_golden.py: same weights, depth_values = [min, max, interval])."""
    from ada_mvs_amd.models.adamvs import AdaMVSNet
    g, tw = load_golden("e2e_tiny"), load_golden("e2e_tiny_twin")
    c = synth.CONFIGS["tiny"]
    m = AdaMVSNet(c["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    dv = g["depth_values"]
    dv3 = torch.cat([dv, (dv[:, 1:2] - dv[:, 0:1]) / c["num_depth"]], 1)
    proj = {"stage%d" % k: dev(g["proj_stage%d" % k]) for k in (1, 2, 3)}
    with torch.no_grad():
        out = m(dev(g["imgs"]), proj, dev(dv3))
    assert rel_l1(out["depth"], tw["depth"]) < OP_TOL                       # measured 2e-7
    assert rel_l1(out["photometric_confidence"], tw["photometric_confidence"]) < OP_TOL     # measured 2e-6
    assert len(out["stage3"]["pair_confidence"]) == g["imgs"].shape[1] - 1
    # and it is NOT the inference model: the two agree only to 1e-4 (different eps placement and resampling)
    assert rel_l1(out["depth"], g["depth"]) > OP_TOL
    m.train()
    with pytest.raises(Exception):
        m(dev(g["imgs"]), proj, dev(dv3))


# --------------------------------------------------------------------------- software-pipelined recurrence
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("cfg,batch", [("tiny", 3), ("cfg1", 2)])
def test_pipelined_recurrence_is_bit_identical_to_sequential(hip, set_option, cfg, batch, precision):
    """The stage driver skews the steps (level 1 of hypothesis t, level 2 of t-1 / t-2 and the decoder of t-2 / t-3 share
    launches, state rings of 4 / 2 buffers, hypotheses in chunks of 32 across which the pipeline keeps running): the
    arithmetic per tile is that of the one-role kernels, so the maps of both schedules must equal, bit for bit, those
    of option recur_mode = 0 (six dependent launches per hypothesis, states updated in place).  cfg1 has 48 hypotheses
    at stage 1: two chunks."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS[cfg]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision=precision)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=batch, seed=21)
    args = (dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
    set_option("gru_wino", 0)       # mode 0 on the direct kernels too (its F(2x2, 3x3) form: the test below)
    outs = {}
    for mode in ("0", "1", "3", "5"):      # 5: one launch per hypothesis (both levels fused)
        set_option("recur_mode", int(mode))
        with torch.no_grad():
            outs[mode] = m(*args)
        torch.cuda.synchronize()
    for mode in ("1", "3", "5"):
        for s in ("stage1", "stage2", "stage3"):
            for key in ("depth", "photometric_confidence"):
                assert torch.equal(outs["0"][s][key], outs[mode][s][key]), (mode, s, key)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("stage,h,w", [(1, 22, 38), (2, 26, 50), (1, 4, 6)])
def test_pipelined_recurrence_on_ragged_stage_sizes(hip, O, set_option, precision, stage, h, w):
    """One cascade stage (InferDepthNet0.forward, reference adamvs.py:433-533) on maps no tile size of any role divides
    (level-2 maps 11 x 19, 13 x 25, 2 x 3; the end-to-end model only meets multiples of 8) and 40 hypotheses = one
    full chunk of 32 plus a ragged one.  Every schedule of the recurrence -- the two-launch one runs conv2 inside the
    level-2 gate tiles -- must equal the sequential launches bit for bit, and those the CPU oracle."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    B, V, D = 2, 3, 40
    m = Infer_AdaMVSNet(48, [48, 32, 8], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision=precision)
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    net = m.DepthNet[stage]
    C = (32, 16, 8)[stage]
    feats = [synth.smooth_features(B, C, h, w, seed=40 + v) for v in range(V)]
    proj = synth.rig_projections(V, 4 * h, 4 * w, batch=B)["stage1"]
    g = torch.Generator().manual_seed(7)
    near = 420.0 + 20.0 * torch.rand(B, 1, h, w, generator=g)
    planes = (near + 4.0 * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).contiguous()
    prev = [torch.rand(B, 1, h // 2, w // 2, generator=g) for _ in range(V - 1)]
    set_option("gru_wino", 0)       # mode 0 on the direct kernels too
    outs = {}
    for mode in ("0", "1", "3", "5"):      # 5: one launch per hypothesis (both levels fused)
        set_option("recur_mode", int(mode))
        with torch.no_grad():
            outs[mode] = net([dev(f) for f in feats], dev(proj), dev(planes), D, [dev(c) for c in prev])
        torch.cuda.synchronize()
    for mode in ("1", "3", "5"):
        for key in ("depth", "photometric_confidence"):
            assert torch.equal(outs["0"][key], outs[mode][key]), (mode, key)
    with torch.no_grad():
        ref = O.infer_depth_stage(feats, proj, planes, sd, "DepthNet.%d." % stage, net.in_up, prev)
    tol = E2E_TOL if precision == "fp32" else 5e-4
    for key in ("depth", "photometric_confidence"):
        assert outs["0"][key].shape == ref[key].shape
        assert rel_l1(outs["0"][key], ref[key]) < tol, key


@pytest.mark.parametrize("seed", list(range(8)))
def test_stage_on_random_shapes_against_oracle(hip, O, seed):
    """InferDepthNet0.forward (reference adamvs.py:433-533) on shapes drawn at random -- stage (first stage with
    CostRegNet2D, or a later one with resampled view weights), batch, number of source views, map size (even, down to
    4 x 4), number of hypotheses, precision -- against the CPU oracle.  Seeds are fixed: the cases are reproducible."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    rng = np.random.default_rng(1000 + seed)
    stage = int(rng.integers(0, 3))
    precision = "bf16x3" if seed % 2 else "fp32"
    B, V = int(rng.integers(1, 4)), int(rng.choice([2, 3, 5, 9]))
    if stage == 0:
        D = int(rng.choice([32, 48, 64]))
        h, w = 8 * int(rng.integers(1, 4)), 8 * int(rng.integers(1, 5))          # CostRegNet2D: three stride-2 levels
    else:
        D = int(rng.integers(2, 41))          # one hypothesis is refused (the reference's plane spacing divides by D - 1)
        h, w = 2 * int(rng.integers(2, 20)), 2 * int(rng.integers(2, 24))
    m = Infer_AdaMVSNet(D if stage == 0 else 48, [D if stage == 0 else 48, 32, 8], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8],
                        precision=precision)
    sd = synth.seeded_state_dict(m, seed=seed)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    net = m.DepthNet[stage]
    C = (32, 16, 8)[stage]
    feats = [synth.smooth_features(B, C, h, w, seed=100 * seed + v) for v in range(V)]
    proj = synth.rig_projections(V, 4 * h, 4 * w, batch=B)["stage1"]
    g = torch.Generator().manual_seed(seed)
    near = 420.0 + 20.0 * torch.rand(B, 1, h, w, generator=g)
    planes = (near + (160.0 / max(D, 2)) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).contiguous()
    prev = None if stage == 0 else [torch.rand(B, 1, max(h // 2, 1), max(w // 2, 1), generator=g) for _ in range(V - 1)]
    with torch.no_grad():
        got = net([dev(f) for f in feats], dev(proj), dev(planes), D, None if prev is None else [dev(c) for c in prev])
        ref = O.infer_depth_stage(feats, proj, planes, sd, "DepthNet.%d." % stage, net.in_up, prev)
    tol = E2E_TOL if precision == "fp32" else 5e-4
    case = (stage, precision, B, V, h, w, D)
    for key in ("depth", "photometric_confidence"):
        assert got[key].shape == ref[key].shape, case
        assert rel_l1(got[key], ref[key]) < tol, (key,) + case
    for a, b in zip(got["pair_confidence"][:V - 1], ref["pair_confidence"]):
        assert rel_l1(a, b) < tol, ("pair_confidence",) + case


def test_soft_argmin_op(hip):
    """adamvs_soft_argmin (reference adamvs.py:516-531) by itself: exp without max subtraction, strict
    '<' running maximum from 0, +1e-10 on the sum.  (The 2x-upsampled-plane form and the chunked accumulation of the
    stage driver are held by the stage / end-to-end fixtures.)"""
    B, D, h, w = 2, 7, 6, 10
    g = torch.Generator().manual_seed(3)
    planes = 400 + 200 * torch.rand(B, D, h, w, generator=g)
    vol = torch.randn(B, D, h, w, generator=g) * 2
    vol[0, :, 0, 0] = -200.0                               # exp underflows to 0 everywhere: depth = 0 / 1e-10, confidence 0
    p = vol.exp()
    den = p.sum(1) + 1e-10
    ref_d, ref_c = (planes * p).sum(1) / den, p.max(1)[0] / den
    d, cf = hip.soft_argmin(dev(vol), dev(planes), B, D, h, w)
    assert rel_l1(d, ref_d) < 1e-6 and rel_l1(cf, ref_c) < 1e-5
    assert float(d[0, 0, 0]) == 0.0 and float(cf[0, 0, 0]) == 0.0


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_generated_planes_equal_materialised_planes(hip, precision):
    """The stage kernels generate the hypothesis planes (uniform over [min, max] at stage 1, the per-pixel window around the
    previous depth afterwards; reference module.py:628-663) instead of reading a [B,D,h,w] tensor: with the reference's
    operation order (rounded product, rounded sum) every map must equal, bit for bit, the run on materialised planes --
    pass A, the aggregation sweep, and the soft-argmin with and without the 2x upsampling of the planes."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS["cfg1"]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision=precision)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs("cfg1", batch=2, seed=5)
    dv[1] = torch.tensor([380.0, 640.0])                       # per-tile ranges (quirk Q4: the interval still comes from tile 0)
    args = (dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
    outs = []
    for flag in (False, True):
        m.materialize_planes = flag
        with torch.no_grad():
            outs.append(m(*args))
    for s in ("stage1", "stage2", "stage3"):
        for key in ("depth", "photometric_confidence"):
            assert torch.equal(outs[0][s][key], outs[1][s][key]), (s, key)
        for a, b in zip(outs[0][s]["pair_confidence"][:2], outs[1][s]["pair_confidence"][:2]):
            assert torch.equal(a, b)
    for a, b in zip(outs[0]["stage1"]["pair_result"], outs[1]["stage1"]["pair_result"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D,stage", [(48, 0), (192, 0), (40, 1), (24, 1)])
def test_piecewise_phase_masks(hip, D, stage):
    """adamvs_depth_stage_forward's phase mask (include/adamvs_hip.h): VIEW_WEIGHTS in a call of its own followed by
    AGGREGATE | RECURRENCE | SOFT_ARGMIN in one call equals PHASE_ALL bit for bit at any D.  The workspace keeps ONE chunk of
    32 hypotheses, so with D > 32 a proper subset of the last three is refused (it cannot hand results to a later call)
    (the measurement entry point adamvs_bench_stage_phase runs it for its duration alone); with D <= 32 the phases may run one by
    one and still produce the maps."""
    from ada_mvs_amd import _lib
    from ada_mvs_amd._lib import AdaMVSHipError
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    B, V, h, w = 2, 3, 16, 24
    m = Infer_AdaMVSNet(D if stage == 0 else 48, [D if stage == 0 else 48, 32, 8], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    net = m.DepthNet[stage]
    C = (32, 16, 8)[stage]
    feats = [synth.smooth_features(B, C, h, w, seed=70 + v) for v in range(V)]
    feat_cl = hip.pack_features(dev(torch.stack(feats, 0).reshape(-1, C, h, w)))
    rt = hip.relative_transforms(dev(synth.rig_projections(V, 4 * h, 4 * w, batch=B)["stage1"]))
    g = torch.Generator().manual_seed(11)
    planes = dev((430.0 + 10.0 * torch.rand(B, 1, h, w, generator=g) + (150.0 / D) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).contiguous())
    prev = None if stage == 0 else dev(torch.rand(V - 1, B, h // 2, w // 2, generator=g))
    Ho, Wo = (2 * h, 2 * w) if net.in_up else (h, w)

    def outputs():
        return (torch.zeros(V - 1, B, h, w, device="cuda"), torch.zeros(V - 1, B, h, w, device="cuda") if stage == 0 else None,
                torch.zeros(B, Ho, Wo, device="cuda"), torch.zeros(B, Ho, Wo, device="cuda"))

    rest = _lib.PHASE_AGGREGATE | _lib.PHASE_RECURRENCE | _lib.PHASE_SOFT_ARGMIN
    with torch.no_grad():
        whole = net.run(feat_cl, B, C, h, w, rt, planes, prev)
        o = outputs()
        net.run(feat_cl, B, C, h, w, rt, planes, prev, phases=_lib.PHASE_VIEW_WEIGHTS, outputs=o)
        net.run(feat_cl, B, C, h, w, rt, planes, prev, phases=rest, outputs=o)
        torch.cuda.synchronize()
        for a, b in zip(whole, o):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b)
        for sub in (_lib.PHASE_AGGREGATE, _lib.PHASE_RECURRENCE, _lib.PHASE_SOFT_ARGMIN, _lib.PHASE_AGGREGATE | _lib.PHASE_RECURRENCE,
                    _lib.PHASE_VIEW_WEIGHTS | _lib.PHASE_SOFT_ARGMIN):
            if D > 32:
                with pytest.raises(AdaMVSHipError, match="adamvs_bench_stage_phase"):
                    net.run(feat_cl, B, C, h, w, rt, planes, prev, phases=sub, outputs=outputs())
                net.run(feat_cl, B, C, h, w, rt, planes, prev, phases=sub, outputs=outputs(), timing_only=True)    # runs; no maps promised
        if D <= 32:            # one chunk: the phases one by one, in order, still make the maps
            o = outputs()
            for ph in (_lib.PHASE_VIEW_WEIGHTS, _lib.PHASE_AGGREGATE, _lib.PHASE_RECURRENCE, _lib.PHASE_SOFT_ARGMIN):
                net.run(feat_cl, B, C, h, w, rt, planes, prev, phases=ph, outputs=o)
            torch.cuda.synchronize()
            assert torch.equal(whole[2], o[2]) and torch.equal(whole[3], o[3])
        torch.cuda.synchronize()


@pytest.mark.parametrize("D,h,w", [(192, 16, 32), (48, 24, 40), (64, 8, 16), (256, 8, 16)])
def test_prob_softmax_regress_fused(hip, D, h, w):  # fp32; the bf16x3 twin is held by the stage / end-to-end bf16x3 tests
    """adamvs_prob_softmax_regress -- the last CostRegNet2D layer with softmax / max / depth regression in its epilogue
    (reference adamvs.py:238, 481-486; what the fp32 stage runs) -- against the two separate ops, whose parity with the
    oracle the tests above hold: same view weights and pair depths to fp32 summation order, on maps whose sizes are not
    multiples of the 8 x 16 block (h = 24) and with per-pixel planes."""
    S, B = 2, 2
    g = torch.Generator().manual_seed(D + h)
    x = dev(torch.randn(S * B, h * w, D, generator=g))
    wl = dev(torch.randn(9 * D * D + D, generator=g) * (2.0 / (9 * D)) ** 0.5 * 3.0)        # gain 3 on the logits layer, as in synth
    planes = dev((400.0 + 20.0 * torch.rand(B, 1, h, w, generator=g) + (180.0 / D) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).contiguous())
    score = hip.conv3x3_dd(x, wl, wl[9 * D * D:], None, S * B, D, h, w, 0, False)
    vw0, pd0 = hip.softmax_max_regress(score, planes, S, B, D, h, w)
    vw1, pd1 = hip.prob_softmax_regress(x, wl, wl[9 * D * D:], planes, S, B, D, h, w)
    torch.cuda.synchronize()
    assert rel_l1(vw1, vw0) < 2e-6 and rel_l1(pd1, pd0) < 2e-6
    assert float((vw1 - vw0).abs().max()) < 1e-5 and float((pd1 - pd0).abs().max() / 500.0) < 1e-5


@pytest.mark.parametrize("D,h,w", [(64, 24, 40), (192, 5, 70), (192, 32, 48), (384, 9, 33), (512, 6, 35)])
def test_prob_softmax_regress_winograd(hip, D, h, w):
    """adamvs_prob_softmax_regress_wino -- `prob` in the F(2x2, 3x3) form with every lane's softmax partial in its epilogue and a
    merge kernel behind it, the score volume never stored (what the fp32 stage runs at these widths; reference adamvs.py:238,
    481-486) -- against the layer and the softmax / regression as two ops: ragged tiles, planes uniform per tile, one to six channel
    groups per pixel, both tilings of the kernel (32 x 48 pixels take the two-workgroups-per-CU form)."""
    from ada_mvs_amd import packing
    S, B = 2, 2
    g = torch.Generator().manual_seed(D + h)
    x = dev(torch.randn(S * B, h * w, D, generator=g))
    wt = torch.randn(D, D, 3, 3, generator=g) * (2.0 / (9 * D)) ** 0.5 * 3.0                 # gain 3 on the logits layer, as in synth
    bias = dev(torch.randn(D, generator=g) * 0.3)
    wk = dev(packing.pack_reg_layer_wino(wt, torch.ones(D)))
    rng = torch.tensor([[400.0, 580.0], [350.0, 610.0]])                                      # first and last plane of the two tiles
    step = (rng[:, 1] - rng[:, 0]) / (D - 1)
    planes = dev((rng[:, 0].view(B, 1, 1, 1) + torch.arange(D, dtype=torch.float32).view(1, D, 1, 1) * step.view(B, 1, 1, 1)).expand(B, D, h, w).contiguous())
    score = hip.conv3x3_dd_wino(x, wk, bias, None, S * B, D, h, w, 0)
    vw0, pd0 = hip.softmax_max_regress(score, planes, S, B, D, h, w)
    vw1, pd1 = hip.prob_softmax_regress_wino(x, wk, bias, dev(rng), S, B, D, h, w)
    torch.cuda.synchronize()
    assert rel_l1(vw1, vw0) < 2e-6 and rel_l1(pd1, pd0) < 2e-6
    assert float((vw1 - vw0).abs().max()) < 1e-5 and float((pd1 - pd0).abs().max() / 500.0) < 1e-5


@pytest.mark.parametrize("C,S", [(32, 1), (32, 4), (32, 5), (32, 8), (16, 2), (16, 4), (16, 7), (8, 1), (8, 3), (8, 4), (8, 8)])
def test_sweep_blend_views_and_widths(hip, O, C, S):
    """The aggregation sweep (k_sweep_blend, reference adamvs.py:495-512) for every channel width and 1 ... 8 source views
    (lane q of a quad projects views q, q + 4; with C = 8 a quad holds two pixels and a lane projects up to four views),
    on a map whose width is no multiple of the pixels a workgroup takes, 19 planes (two full groups of 8 and a ragged one)
    with per-pixel spacing, a baseline that sends the far views out of bounds: against the oracle."""
    import torch.nn.functional as F
    from ada_mvs_amd import packing
    B, D, h, w = 2, 19, 14, 42
    feats = [synth.smooth_features(B, C, h, w, seed=30 + v) for v in range(S + 1)]
    proj = synth.rig_projections(S + 1, 4 * h, 4 * w, batch=B, baseline=20.0)["stage1"]
    g = torch.Generator().manual_seed(100 * C + S)
    lo = 380 + 40 * torch.rand(B, 1, h, w, generator=g)
    step = (200 + 40 * torch.rand(B, 1, h, w, generator=g)) / (D - 1)
    planes = (lo + step * torch.arange(D, dtype=torch.float32).reshape(1, D, 1, 1)).contiguous()
    vw = torch.rand(S, B, h, w, generator=g)
    w1 = torch.randn(8, C, 3, 3, generator=g) * 0.1
    args = (hip.pack_features(dev(torch.stack(feats, 0).reshape(-1, C, h, w))), hip.relative_transforms(dev(proj)), dev(planes), dev(vw),
            packing.pack_conv1_two_row(w1).cuda(), B, S, C, D, h, w)
    c1 = hip.aggregate_conv1(*args).cpu()
    Rs, ts = zip(*[O.relative_transform(proj[:, s + 1], proj[:, 0]) for s in range(S)])
    for d in (0, 7, 8, 18):
        sim = O.aggregate_similarity(feats[0], feats[1:], Rs, ts, planes[:, d], [vw[s].unsqueeze(1) for s in range(S)])
        ref = F.relu(F.conv2d(sim, w1, None, 1, 1))
        assert rel_l1(c1[d].reshape(B, h, w, 8).permute(0, 3, 1, 2), ref) < OP_TOL, "plane %d" % d


def test_feature_net0_reads_views_in_place(hip):
    """FeatureNet0.forward_cl on the [B,V,3,H,W] tensor Infer_AdaMVSNet.forward receives (adamvs_feature_net0_views: image
    m = v*B + b read from imgs[b][v], reference adamvs.py:574-577 runs the net view by view) equals, bit for bit, the
    maps of the view-major copy -- in one call and in chunks of three images (an uneven split of the 2 x 3 = 6)."""
    from ada_mvs_amd.models.adamvs import FeatureNet0
    B, V, H, W = 2, 3, 64, 96
    net = FeatureNet0(base_channels=8, stride=4, num_stage=3)
    net.load_state_dict(synth.seeded_state_dict(net, seed=2))
    net = net.cuda().eval()
    imgs = dev(torch.randn(B, V, 3, H, W, generator=torch.Generator().manual_seed(9)))
    want = net.forward_cl(imgs.transpose(0, 1).reshape(B * V, 3, H, W).contiguous())
    got = net.forward_cl(imgs)
    net.workspace_limit_bytes = 4 * hip.feature_net0_workspace_bytes(1, H, W)          # chunks of 4 and 2 images
    got_chunked = net.forward_cl(imgs)
    torch.cuda.synchronize()
    for a, b, c in zip(want, got, got_chunked):
        assert a.shape == b.shape == c.shape
        assert torch.equal(a, b) and torch.equal(a, c)


# --------------------------------------------------------------------------- planes behind / on a source camera (module.py:549-553)
def _nan_aware_rel_l1(x, ref):
    x, ref = x.detach().double().cpu(), ref.detach().double().cpu()
    assert torch.equal(torch.isnan(x), torch.isnan(ref)), "NaN pattern differs: %d vs %d" % (int(torch.isnan(x).sum()), int(torch.isnan(ref).sum()))
    ok = ~torch.isnan(ref)
    return float((x[ok] - ref[ok]).abs().mean() / ref[ok].abs().mean().clamp_min(1e-30))


def _behind_rig(hip):
    g = load_golden("op_warp_behind")
    proj = torch.stack((g["ref_proj"], g["src_proj"]), 1)                    # [1, 2, 4, 4]: view 0 = reference (identity)
    rt = hip.relative_transforms(dev(proj))
    assert torch.equal(rt.cpu()[0, 0, :9].reshape(3, 3), g["src_proj"][0, :3, :3]) and torch.equal(rt.cpu()[0, 0, 9:], g["src_proj"][0, :3, 3]), \
        "P_src . I^-1 must be P_src exactly: the fixture's X2 == 0 row depends on it"
    return g, rt


def test_homo_warping_float_behind_the_source_camera(hip):
    """The reference divides by X2 whatever its sign (module.py:553).  Fixture = a run of the reference: rows where the plane
    is behind the source camera (mirrored, finite coordinates), one row ON its focal plane (inf / NaN coordinates -> NaN in
    every channel out of grid_sample).  adamvs_homo_warp must give the same values and the same NaNs."""
    from ada_mvs_amd.models.module import homo_warping_float
    g = load_golden("op_warp_behind")
    out = homo_warping_float(dev(g["src"]), dev(g["src_proj"]), dev(g["ref_proj"]), dev(g["depth"]))
    assert int(torch.isnan(g["out"]).sum()) == 8 * 24
    assert _nan_aware_rel_l1(out, g["out"]) < OP_TOL
    for d in range(3):
        assert _nan_aware_rel_l1(out[:, :, d], g["out"][:, :, d]) < OP_TOL, d


@pytest.mark.parametrize("order", ["as_is", "reversed"])
def test_pair_similarity_behind_the_source_camera(hip, order):
    """adamvs_pair_similarity on the same rig: expected = mean_c(ref * warped) with the REFERENCE's warped planes
    (adamvs.py:475-476 on the fixture's output), NaN where the reference's is.  Both plane orders: the kernel caches the taps of
    a source cell across planes, and a NaN plane must neither reuse nor poison the cache of its neighbours."""
    g, rt = _behind_rig(hip)
    B, C, h, w = 1, 8, 16, 24
    D = g["depth"].shape[1]
    sel = list(range(D)) if order == "as_is" else list(range(D))[::-1]
    ref = synth.smooth_features(B, C, h, w, seed=9)
    feat_cl = hip.pack_features(dev(torch.cat((ref, g["src"]), 0)))
    planes = g["depth"][:, sel].contiguous()
    sim = hip.pair_similarity(feat_cl, rt, dev(planes), B, 1, C, D, h, w).cpu().reshape(B, h, w, D).permute(0, 3, 1, 2)
    want = (ref.unsqueeze(2) * g["out"][:, :, sel]).mean(1)                   # [B, D, h, w]
    assert int(torch.isnan(want).sum()) == 24
    assert _nan_aware_rel_l1(sim, want) < OP_TOL


@pytest.mark.parametrize("S", [1, 2, 5])
def test_sweep_behind_the_source_camera(hip, O, S):
    """The aggregation sweep on the same rig: source view 0 is the fixture's camera (expected from the reference's own warped
    planes), further views are benign ones (expected from the oracle's warp, itself pinned by the fixtures).  The aggregated
    similarity the sweep leaves in its workspace must carry NaN exactly where the reference's sum does (adamvs.py:495-512)."""
    from ada_mvs_amd import packing
    g, rt0 = _behind_rig(hip)
    B, C, h, w = 1, 8, 16, 24
    D = g["depth"].shape[1]
    ref = synth.smooth_features(B, C, h, w, seed=9)
    others = [synth.smooth_features(B, C, h, w, seed=20 + v) for v in range(S - 1)]
    sp = g["src_proj"].clone()
    projs = [g["ref_proj"], g["src_proj"]]
    for v in range(S - 1):                                                   # in front of every plane: X2 = d + 40 + 8 v
        q = torch.eye(4)[None].clone()
        q[0, 0, 3], q[0, 1, 3], q[0, 2, 3] = 40.0 * (v + 1), -25.0 * (v + 1), 40.0 + 8.0 * v
        projs.append(q)
    proj = torch.stack(projs, 1)
    rt = hip.relative_transforms(dev(proj))
    assert torch.equal(rt[:, 0], rt0[:, 0])
    gen = torch.Generator().manual_seed(S)
    vw = torch.rand(S, B, h, w, generator=gen) + 0.05
    w1 = torch.randn(8, C, 3, 3, generator=gen) * 0.1
    feat_cl = hip.pack_features(dev(torch.cat([ref, g["src"]] + others, 0)))
    c1, sim = hip.aggregate_conv1(feat_cl, rt, dev(g["depth"]), dev(vw), packing.pack_conv1_two_row(w1).cuda(), B, S, C, D, h, w,
                                  return_similarity=True)
    sim = sim.cpu().reshape(D, B, h, w, C).permute(1, 4, 0, 2, 3)           # [B, C, D, h, w]
    for d in range(D):
        num = g["out"][:, :, d] * ref * vw[0].unsqueeze(1)
        den = 1e-5 + vw[0].unsqueeze(1)
        for v in range(S - 1):
            R, t = O.relative_transform(proj[:, 2 + v], proj[:, 0])
            num = num + O.warp_plane(others[v], R, t, g["depth"][:, d]) * ref * vw[1 + v].unsqueeze(1)
            den = den + vw[1 + v].unsqueeze(1)
        assert _nan_aware_rel_l1(sim[:, :, d], num / den) < OP_TOL, d
    assert bool(torch.isnan(sim[0, :, 0, 6]).all()) and bool(torch.isfinite(sim[0, :, 0, :6]).all())


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_end_to_end_with_a_view_behind_the_planes(hip, precision):
    """The drop-in forward() on a rig whose second source camera has every hypothesis plane crossing its focal plane inside the
    image (fixture: a run of the reference; tools/gen_golden.py::behind_rig) -- mirrored samples on one side, none exactly on it."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    g = load_golden("e2e_tiny_behind")
    c = synth.CONFIGS["tiny"]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision=precision)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs, _, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    proj = {k[5:]: dev(v) for k, v in g.items() if k.startswith("proj_")}
    with torch.no_grad():
        out = m(dev(imgs), proj, dev(dv))
    tol = E2E_TOL if precision == "fp32" else 5e-4
    for s in (1, 2, 3):
        assert rel_l1(out["stage%d" % s]["depth"], g["s%d_depth" % s]) < tol, s
        assert rel_l1(out["stage%d" % s]["photometric_confidence"], g["s%d_conf" % s]) < tol, s
    for i in range(2):
        assert rel_l1(out["stage1"]["pair_confidence"][i], g["s1_pairconf%d" % i]) < tol
        assert rel_l1(out["stage1"]["pair_result"][i], g["s1_pairdepth%d" % i]) < tol


# --------------------------------------------------------------------------- a trained network's dynamic range (adamvs.py:481, 516-531)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name,recipe", [("e2e_tiny_sharp", "sharp"), ("e2e_tiny_sharp64", "sharp"), ("e2e_tiny_overflow", "overflow")])
def test_end_to_end_on_a_trained_networks_dynamic_range(hip, name, recipe, precision):
    """Fixtures: runs of the reference on synth.LOGIT_GAINS "sharp" / "overflow" weights (tools/gen_golden.py::end_to_end_sharp).
    Near one-hot stage-1 softmaxes through the zero-padded 48-channel tiling with -1e30 pad scores (D1 = 40) and through the
    F(2x2, 3x3) `prob` with per-lane online-softmax partials + k_softmax_merge (D1 = 64); reg_cost up to +-60 through the
    unstabilised exp of the running soft-argmin; and a single stage in which that exp overflows: where the reference returns
    inf / NaN (adamvs.py:529-531) so must the kernels, pixel for pixel (two pixels of slack for scores within rounding of the
    overflow threshold)."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    from test_oracle_golden import finite_rel_l1, nonfinite_mismatch
    g = load_golden(name)
    nd = [int(v) for v in g["ndepths"]]
    cfg = dict(views=3, H=64, W=96, ndepths=nd, num_depth=nd[0])
    m = Infer_AdaMVSNet(nd[0], nd, synth.DEPTH_INTERVALS_RATIO[:len(nd)], False, [8, 8, 8], precision=precision)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0, recipe=recipe))
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=0)
    with torch.no_grad():
        out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
    tol = E2E_TOL if precision == "fp32" else 5e-4
    for s in range(1, len(nd) + 1):
        for key, gk in (("depth", "s%d_depth" % s), ("photometric_confidence", "s%d_conf" % s)):
            got, ref = out["stage%d" % s][key].cpu(), g[gk]
            if recipe == "overflow":
                assert nonfinite_mismatch(got, ref) <= 2, (name, key, nonfinite_mismatch(got, ref))
            else:
                assert bool(torch.isfinite(got).all())
            assert finite_rel_l1(got, ref) < tol, (name, s, key, finite_rel_l1(got, ref))
    for i in range(2):
        assert rel_l1(out["stage1"]["pair_confidence"][i], g["s1_pairconf%d" % i]) < tol
        assert rel_l1(out["stage1"]["pair_result"][i], g["s1_pairdepth%d" % i]) < tol


# --------------------------------------------------------------------------- any number of hypotheses / views (adamvs.py:198-228, :464, :501)
@pytest.mark.parametrize("D,precision", [(40, "fp32"), (80, "fp32"), (160, "fp32"), (384, "fp32"), (272, "fp32"), (24, "fp32"),
                                         (160, "bf16x3"), (384, "bf16x3"), (288, "bf16x3"),
                                         (512, "fp32"), (448, "bf16x3")])        # (a minute and a half of CPU oracle each: 512 recurrent planes)
def test_stage_one_at_any_hypothesis_count(hip, O, D, precision):
    """CostRegNet2D(in_channels) is built for any D in the reference; here D hypotheses run at the next width the kernels are
    built for (zero filters, zero similarity channels, -1e30 pad scores: csrc/costreg2d.hip::costreg_width).  Stage 1 with
    caller-made (explicit) planes against the oracle, whose network has exactly D channels."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    from ada_mvs_amd import packing
    B, V, h, w = 2, 3, 8, 16
    m = Infer_AdaMVSNet(D, [D, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision=precision)
    sd = synth.seeded_state_dict(m, seed=D)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    net = m.DepthNet[0]
    assert net.reg.effective_precision() == precision and packing.reg_width(D, precision) >= D
    feats = [synth.smooth_features(B, 32, h, w, seed=7 * D + v) for v in range(V)]
    proj = synth.rig_projections(V, 4 * h, 4 * w, batch=B, baseline=30.0)["stage1"]
    g = torch.Generator().manual_seed(D)
    near = 420.0 + 20.0 * torch.rand(B, 1, h, w, generator=g)
    planes = (near + (160.0 / D) * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).contiguous()
    with torch.no_grad():
        got = net([dev(f) for f in feats], dev(proj), dev(planes), D, None)
        ref = O.infer_depth_stage(feats, proj, planes, sd, "DepthNet.0.", True, None)
    tol = E2E_TOL if precision == "fp32" else 5e-4
    for key in ("depth", "photometric_confidence"):
        assert rel_l1(got[key], ref[key]) < tol, key
    for a, b in zip(got["pair_confidence"][:V - 1], ref["pair_confidence"]):
        assert rel_l1(a, b) < tol
    for a, b in zip(got["pair_result"], ref["pair_result"]):
        assert rel_l1(a, b) < tol


@pytest.mark.parametrize("ndepths", [[40, 24, 6], [80, 16, 8]])
def test_forward_with_hypothesis_counts_off_the_kernel_widths(hip, O, ndepths):
    """The drop-in forward() (generated planes) for a checkpoint trained with --ndepths 40,... / 80,...: every stage map against the oracle."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    cfg = dict(views=3, H=64, W=96, ndepths=ndepths, num_depth=ndepths[0])
    m = Infer_AdaMVSNet(cfg["num_depth"], ndepths, synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=3)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=2, seed=5)
    with torch.no_grad():
        out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
        ref = O.infer_adamvs_forward(imgs, proj, dv, sd, cfg["num_depth"], ndepths, synth.DEPTH_INTERVALS_RATIO)
    for s in (1, 2, 3):
        for key in ("depth", "photometric_confidence"):
            assert rel_l1(out["stage%d" % s][key], ref["stage%d" % s][key]) < E2E_TOL, (s, key)


@pytest.mark.parametrize("D", [40, 272])
def test_cost_reg_net_2d_module_at_any_width(hip, O, D):
    from ada_mvs_amd.models.adamvs import CostRegNet2D
    net = CostRegNet2D(D)
    sd = synth.seeded_state_dict(net, seed=2)
    net.load_state_dict(sd)
    x = torch.randn(2, D, 8, 16, generator=torch.Generator().manual_seed(D)) * 0.5
    out = net.cuda()(dev(x))
    assert out.shape == x.shape and rel_l1(out, O.cost_reg_net_2d(x, sd, "")) < OP_TOL


def test_cost_reg_weights_of_another_layout_are_refused(hip):
    """Advisor (round 3): a blob of the 11-block layout at a width that carries the F(2x2, 3x3) blocks was read out of bounds.
    The blob's length is part of the call now."""
    from ada_mvs_amd import _lib, packing
    from ada_mvs_amd.models.adamvs import CostRegNet2D
    net = CostRegNet2D(64)
    net.load_state_dict(synth.seeded_state_dict(net, seed=2))
    wpk = packing.pack_cost_reg_net_2d(net.state_dict(), "", "fp32").cuda()
    lib = _lib.load()
    assert wpk.numel() == lib.adamvs_cost_reg_net_2d_weight_floats(64, 0) == 11 * (9 * 64 * 64 + 64) + 5 * 16 * 64 * 64
    x = torch.zeros(1, 64, 64, device="cuda")
    hip.cost_reg_net_2d(x, wpk, 8, 8)
    with pytest.raises(_lib.AdaMVSHipError, match="wpk holds"):
        hip.cost_reg_net_2d(x, wpk[:11 * (9 * 64 * 64 + 64)].contiguous(), 8, 8)


@pytest.mark.parametrize("C,S", [(32, 10), (16, 9), (8, 13), (8, 17)])
def test_sweep_with_more_than_eight_source_views(hip, O, C, S):
    """The reference loops over any number of source views (adamvs.py:501); the sweep takes them in groups of eight, the later
    groups adding to the first one's sums (csrc/sweep.hip).  Aggregated similarity and conv1 against the oracle."""
    import torch.nn.functional as F
    from ada_mvs_amd import packing
    B, D, h, w = 2, 11, 12, 30
    feats = [synth.smooth_features(B, C, h, w, seed=40 + v) for v in range(S + 1)]
    proj = synth.rig_projections(S + 1, 4 * h, 4 * w, batch=B, baseline=6.0)["stage1"]
    g = torch.Generator().manual_seed(100 * C + S)
    lo = 380 + 40 * torch.rand(B, 1, h, w, generator=g)
    step = (200 + 40 * torch.rand(B, 1, h, w, generator=g)) / (D - 1)
    planes = (lo + step * torch.arange(D, dtype=torch.float32).reshape(1, D, 1, 1)).contiguous()
    vw = torch.rand(S, B, h, w, generator=g)
    w1 = torch.randn(8, C, 3, 3, generator=g) * 0.1
    c1, sim = hip.aggregate_conv1(hip.pack_features(dev(torch.stack(feats, 0).reshape(-1, C, h, w))), hip.relative_transforms(dev(proj)),
                                  dev(planes), dev(vw), packing.pack_conv1_two_row(w1).cuda(), B, S, C, D, h, w, return_similarity=True)
    c1, sim = c1.cpu(), sim.cpu()
    Rs, ts = zip(*[O.relative_transform(proj[:, s + 1], proj[:, 0]) for s in range(S)])
    for d in (0, 7, 8, 10):
        ref = O.aggregate_similarity(feats[0], feats[1:], Rs, ts, planes[:, d], [vw[s].unsqueeze(1) for s in range(S)])
        assert rel_l1(sim[d].reshape(B, h, w, C).permute(0, 3, 1, 2), ref) < OP_TOL, "plane %d" % d
        assert rel_l1(c1[d].reshape(B, h, w, 8).permute(0, 3, 1, 2), F.relu(F.conv2d(ref, w1, None, 1, 1))) < OP_TOL, "plane %d" % d


def test_eleven_views_end_to_end(hip, O):
    """--view_num 11 (10 source views): forward() and the train/test twin against the oracle."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    cfg = dict(views=11, H=64, W=96, ndepths=[16, 8, 4], num_depth=16)
    m = Infer_AdaMVSNet(cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=11, baseline=5.0)
    with torch.no_grad():
        out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
        ref = O.infer_adamvs_forward(imgs, proj, dv, sd, cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    assert len(out["stage1"]["pair_result"]) == 10
    for s in (1, 2, 3):
        for key in ("depth", "photometric_confidence"):
            assert rel_l1(out["stage%d" % s][key], ref["stage%d" % s][key]) < E2E_TOL, (s, key)


@pytest.mark.parametrize("precision", [0, 1])
def test_conv_layers_on_more_images_than_one_grid_takes(hip, precision):
    """The direct CostRegNet2D kernels take the image from blockIdx.z (65535 at most; a quarter of it in the transposed
    split-bf16 layers): more images run as sub-batches.  70 000 maps of 8 x 8 (1 x 2 for the transposed layer): the first,
    the last and the ones around the seam must equal the same maps run by themselves."""
    from ada_mvs_amd import packing
    D, N = 32, 70000
    g = torch.Generator().manual_seed(5)
    w = torch.randn(D, D, 3, 3, generator=g) * 0.05
    layer = (packing.pack_reg_layer if precision == 0 else packing.pack_reg_layer_bf16x3)(w, torch.ones(D), torch.randn(D, generator=g) * 0.1, False).cuda()
    for mode, (hi, wi) in ((0, (8, 8)), (2, (1, 2))):
        x = torch.randn(N, hi * wi, D, generator=g).cuda()
        out = hip.conv3x3_dd(x, layer[:9 * D * D], layer[9 * D * D:], None, N, D, hi, wi, mode, 1, precision=precision)
        idx = torch.tensor([0, 1, 16382, 16383, 16384, 65534, 65535, 65536, N - 1], device="cuda")
        part = hip.conv3x3_dd(x[idx].contiguous(), layer[:9 * D * D], layer[9 * D * D:], None, idx.numel(), D, hi, wi, mode, 1, precision=precision)
        assert torch.allclose(out[idx], part, rtol=1e-4, atol=1e-5), mode      # (small batches may take another tiling: not bit for bit)


@pytest.mark.parametrize("recur,mask", [("0", "1"), ("0", "2"), ("0", "4"), ("0", "8"), ("0", "15"), ("1", "7")])
@pytest.mark.parametrize("cfg,batch", [("tiny", 3), ("cfg1", 2)])
def test_gru_convolutions_in_the_minimal_filtering_form(hip, set_option, cfg, batch, recur, mask):
    """With one role per launch (option recur_mode = 0: what large stages run) the gate convolutions of both ConvGRU levels and
    the level-2 candidate run in the form F(2x2, 3x3) (csrc/slice_roles_wino.h; option gru_wino selects which, default 7):
    16 instead of 36 products, fp32 throughout, so the maps agree with the direct kernels' to rounding (asserted: 2e-5 through
    the whole cascade; measured ~1e-6) and with the reference's fixture where there is one.  Image sizes that are no multiple
    of the 8 x 32 tile (tiny: 16 x 24 at stage 1) exercise the edge paths."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS[cfg]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=batch, seed=21)
    args = (dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
    set_option("recur_mode", int(recur))      # 1: three launches per hypothesis, the same roles sharing launches (mask 7 only)
    outs = {}
    for mk in ("0", mask):
        set_option("gru_wino", int(mk))
        with torch.no_grad():
            outs[mk] = m(*args)
        torch.cuda.synchronize()
    differs = False
    for s in ("stage1", "stage2", "stage3"):
        for key in ("depth", "photometric_confidence"):
            assert rel_l1(outs[mask][s][key], outs["0"][s][key]) < 2e-5, (s, key)
            differs = differs or not torch.equal(outs[mask][s][key], outs["0"][s][key])
    assert differs, "the switch did not select another kernel"


def test_drop_in_forward_with_one_role_per_launch(hip, set_option):
    """The reference's fixtures through the schedule large stages run (one role per launch, GRU convolutions in the F(2x2, 3x3)
    form): the small shapes of the fixtures would otherwise only meet the pipelined schedules."""
    set_option("recur_mode", 0)
    set_option("gru_wino", 15)
    for cfg in ("tiny", "cfg1"):
        g = load_golden("e2e_" + cfg)
        m, _ = _model(cfg)
        imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=0)
        with torch.no_grad():
            out = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
        _check_against_golden(cfg, g, out, E2E_TOL)


@pytest.mark.parametrize("recur", ["0", "1"])
@pytest.mark.parametrize("stage,h,w", [(1, 22, 38), (2, 26, 50), (0, 8, 40), (1, 4, 6), (2, 70, 34)])
def test_minimal_filtering_roles_on_ragged_stage_sizes(hip, O, set_option, recur, stage, h, w):
    """The F(2x2, 3x3) roles (8 x 32 tiles of 2 x 2 output tiles) on maps no tile divides -- level-2 maps of 11 x 19, 13 x 25, 2 x 3,
    35 x 17 pixels: odd sizes cut through the 2 x 2 tiles --, as their own launches and sharing launches, against the CPU oracle."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    B, V, D = 2, 3, 34
    m = Infer_AdaMVSNet(48, [48, 32, 8], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    net = m.DepthNet[stage]
    C = (32, 16, 8)[stage]
    feats = [synth.smooth_features(B, C, h, w, seed=60 + v) for v in range(V)]
    proj = synth.rig_projections(V, 4 * h, 4 * w, batch=B)["stage1"]
    g = torch.Generator().manual_seed(9)
    near = 420.0 + 20.0 * torch.rand(B, 1, h, w, generator=g)
    D_ = 48 if stage == 0 else D
    planes = (near + 4.0 * torch.arange(D_, dtype=torch.float32).view(1, D_, 1, 1)).contiguous()
    prev = None if stage == 0 else [torch.rand(B, 1, h // 2, w // 2, generator=g) for _ in range(V - 1)]
    set_option("recur_mode", int(recur))
    set_option("gru_wino", 7)
    with torch.no_grad():
        got = net([dev(f) for f in feats], dev(proj), dev(planes), D_, None if prev is None else [dev(c) for c in prev])
        ref = O.infer_depth_stage(feats, proj, planes, sd, "DepthNet.%d." % stage, net.in_up, prev)
    for key in ("depth", "photometric_confidence"):
        assert got[key].shape == ref[key].shape
        assert rel_l1(got[key], ref[key]) < E2E_TOL, (key, stage, h, w, recur)


# --------------------------------------------------------------------------- stride-2 layers, minimal filtering along x (adamvs.py:206-211)
@pytest.mark.parametrize("N,D,h,w,relu", [(2, 192, 24, 64, 1), (1, 192, 14, 128, 1), (3, 192, 10, 64, 0), (1, 192, 2, 64, 1), (1, 384, 12, 128, 1),
                                          (40, 192, 24, 64, 1), (1, 192, 6, 520, 1), (2, 192, 14, 66, 1)])
def test_stride_two_layer_in_the_pair_form(hip, set_option, N, D, h, w, relu):
    """k_conv_dd_s2p (csrc/costreg2d.hip: output pairs of a row share their middle input column; five products per pair, kernel row
    and channel pair instead of six) against a float64 convolution : output widths of one, two and four blocks (the launcher takes
    this form where 32-column blocks divide a row, or from 256 columns: 260 = eight blocks and a ragged one with an odd pair), row counts
    no multiple of the block's 3 (7, 5, 1), D = 384 (two launches), 40 maps; the last case (33 columns) stays on the direct kernel.  Option conv_rows2 = 0 keeps these small maps off the 2-row kernel, which the
    launcher would otherwise choose for them."""
    from ada_mvs_amd import packing
    set_option("conv_rows2", 0)
    g = torch.Generator().manual_seed(N * 100 + D + h + w)
    x = torch.randn(N, D, h, w, generator=g)
    wt = torch.randn(D, D, 3, 3, generator=g) / (3 * D ** 0.5)
    scale, shift = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double(), (wt * scale.reshape(-1, 1, 1, 1)).double(), shift.double(), stride=2, padding=1)
    ref = torch.relu(ref) if relu else ref
    ho, wo = ref.shape[-2:]
    x_cl = dev(x.permute(0, 2, 3, 1).reshape(N, h * w, D).contiguous())
    pk = dev(packing.pack_reg_layer(wt, scale, shift, False))
    out = hip.conv3x3_dd(x_cl, pk[:9 * D * D], pk[9 * D * D:], None, N, D, h, w, 1, relu)
    torch.cuda.synchronize()
    got = out.cpu().double().reshape(N, ho, wo, D).permute(0, 3, 1, 2)
    assert rel_l1(got, ref) < 2e-6, rel_l1(got, ref)
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())


# --------------------------------------------------------------------------- one captured graph per input shape (predict_whu.py:100-112)
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_graphed_forward_equals_eager_for_every_depth_range(hip, precision):
    """ada_mvs_amd/graphed.py: the forward of a shape captured once and replayed for every later sample of that shape -- other images,
    other rigs and OTHER DEPTH RANGES (the half span of the window planes is read from device memory by the captured kernels,
    adamvs_stage_desc.half_span_dev; depth_min / depth_max are read from the caller's host tensor, reference adamvs.py:569-571), two
    shapes interleaved (each graph owns its workspace), depth_values handed over on the host and on the device.  Bit for bit the
    eager forward."""
    from ada_mvs_amd.graphed import GraphedForward
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    m = Infer_AdaMVSNet(16, [16, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8], precision=precision)
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    fwd = GraphedForward(m)
    small = dict(views=3, H=64, W=96, ndepths=[16, 8, 4], num_depth=16)
    large = dict(views=3, H=96, W=160, ndepths=[16, 8, 4], num_depth=16)
    cases = [(small, 1, 0, (400.0, 600.0)), (small, 1, 1, (380.0, 640.0)), (large, 2, 2, (400.0, 600.0)), (small, 1, 3, (420.0, 520.0)),
             (large, 2, 4, (350.0, 700.0)), (small, 1, 0, (400.0, 600.0))]
    for i, (cfg, batch, seed, (lo, hi)) in enumerate(cases):
        imgs, proj, _ = synth.tile_inputs(cfg, batch=batch, seed=seed, baseline=8.0 + seed)
        dv = torch.tensor([[lo, hi]] * batch, dtype=torch.float32)
        args = (dev(imgs), {k: dev(v) for k, v in proj.items()})
        with torch.no_grad():
            want = m(*args, dev(dv))
            want = {s: {k: want[s][k].clone() for k in ("depth", "photometric_confidence")} for s in ("stage1", "stage2", "stage3")}
            got = fwd(*args, dv if i % 2 == 0 else dev(dv))
        torch.cuda.synchronize()
        for s in ("stage1", "stage2", "stage3"):
            for k in ("depth", "photometric_confidence"):
                assert torch.equal(got[s][k], want[s][k]), (i, s, k)
        assert torch.equal(got["depth"], want["stage3"]["depth"])
    assert fwd.captures == 2 and len(fwd.cache) == 2
    third = dict(views=3, H=64, W=128, ndepths=[16, 8, 4], num_depth=16)      # a third shape evicts the least recently used graph
    imgs, proj, dv = synth.tile_inputs(third, batch=1, seed=5)
    with torch.no_grad():
        want = m(dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))["depth"].clone()
        got = fwd(dev(imgs), {k: dev(v) for k, v in proj.items()}, dv)["depth"]
    assert torch.equal(got, want) and fwd.captures == 3 and len(fwd.cache) == 2


def test_a_captured_forward_contains_kernel_nodes_only(hip):
    """A memset node inside a captured hipGraph ran out of order with the kernel nodes around it on this stack (round 6,
    profiles/r06_graph_memset_node.txt): a replayed stage then started its recurrence from the previous replay's states.  The library
    zeroes and copies with kernels for that reason (csrc/api.hip zero_floats / copy_floats).  Here: the forward captured the way
    ada_mvs_amd/graphed.py captures it; the node types of the captured graph, read back through hipGraphGetNodes /
    hipGraphNodeGetType, must be kernel nodes (and the empty nodes a capture may insert) -- no memset, no memcpy."""
    import collections
    import ctypes
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    m = Infer_AdaMVSNet(16, [16, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    args = (dev(imgs), {k: dev(v) for k, v in proj.items()}, dev(dv))
    spans = dev(torch.tensor([1.0, 1.0, 1.0]))
    try:
        g = torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError as e:
        pytest.skip("CUDAGraph(keep_graph=True) unavailable: %s" % e)
    with torch.no_grad():
        m(*args)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            feats_cl, shapes = m.extract_features(args[0])
            m.infer_from_features(feats_cl, shapes, args[1], args[2], 0.0, span_dev=spans)
    rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))      # the runtime torch itself uses
    raw = ctypes.c_void_p(g.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert rt.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0 and n.value > 50, n.value
    nodes = (ctypes.c_void_p * n.value)()
    assert rt.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
    kinds = collections.Counter()
    for node in nodes:
        t = ctypes.c_int(-1)
        assert rt.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)) == 0
        kinds[t.value] += 1
    KERNEL, MEMCPY, MEMSET, EMPTY = 0, 1, 2, 5          # hipGraphNodeType
    assert kinds[KERNEL] > 50 and kinds[MEMSET] == 0 and kinds[MEMCPY] == 0 and set(kinds) <= {KERNEL, EMPTY}, dict(kinds)
