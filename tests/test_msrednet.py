"""MS-REDNet inference: oracle/msrednet_oracle.py against fixtures the reference's own
models/msrednet.py produced (tools/gen_golden_msred.py), and -- on the GPU -- the HIP path against both."""
import os
import sys

import numpy as np
import pytest
import torch

import ada_mvs_amd  # noqa: F401
from ada_mvs_amd import synth
from oracle import msrednet_oracle as mo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OP_TOL = 2e-5          # relative L1 per op / network (fp32 on both sides)
E2E_TOL = 1e-3         # BASELINE.json: depth and confidence maps within 1e-3 relative L1


def gold(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}


def rel_l1(a, b):
    return float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-12))


class _Cell(torch.nn.Module):
    def __init__(self, cin, hc):
        super().__init__()
        self.gate_conv = torch.nn.Conv2d(cin + hc, 2 * hc, 3, padding=1)
        self.reset_gate_norm = torch.nn.GroupNorm(1, hc, 1e-5, True)
        self.update_gate_norm = torch.nn.GroupNorm(1, hc, 1e-5, True)
        self.output_conv = torch.nn.Conv2d(cin + hc, hc, 3, padding=1)
        self.output_norm = torch.nn.GroupNorm(1, hc, 1e-5, True)


def cell_state_dict(seed=2):
    return synth.seeded_state_dict(_Cell(16, 16), seed=seed)


def test_oracle_gru_cell2_matches_reference():
    g = gold("msred_gru_cell")
    out = mo.conv_gru_cell2(g["x"], g["h"], cell_state_dict(), "")
    assert rel_l1(out, g["out"]) < 1e-6


def slice_state_dict(C=32, seed=3):
    from ada_mvs_amd.models.msrednet import slice_RED_Regularization
    return synth.seeded_state_dict(slice_RED_Regularization(C, 8), seed=seed)


def test_oracle_slice_step_matches_reference():
    g = gold("msred_slice_step")
    sd = slice_state_dict()
    B, _, h, w = g["cost0"].shape
    states = [torch.zeros(B, 8 << k, h >> k, w >> k) for k in range(4)]
    for step in range(2):
        reg, states = mo.slice_red_step(g["cost%d" % step], states, sd, "")
        assert rel_l1(reg, g["reg%d" % step]) < 1e-5
        for k in range(4):
            assert rel_l1(states[k], g["state%d_%d" % (k + 1, step)]) < 1e-5


def tiny_model_state():
    from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet
    c = synth.CONFIGS["tiny"]
    m = Infer_CascadeREDNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    return m, synth.seeded_state_dict(m, seed=0)


def test_oracle_end_to_end_matches_reference():
    g = gold("msred_e2e_tiny")
    _, sd = tiny_model_state()
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    c = synth.CONFIGS["tiny"]
    with mo.ao.use_grid_sample():
        out = mo.infer_cascade_rednet_forward(imgs, proj, dv, sd, c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    assert rel_l1(out["stage1"]["depth"], g["depth_stage1"]) < 1e-5
    assert rel_l1(out["stage2"]["depth"], g["depth_stage2"]) < 1e-5
    assert rel_l1(out["depth"], g["depth"]) < 1e-5
    assert rel_l1(out["photometric_confidence"], g["photometric_confidence"]) < 1e-4


def test_oracle_feature_net_fpn_matches_reference():
    """FeatureNet(arch_mode="fpn") (reference msrednet.py:74-91, 115-125; no model class selects it, the constructor and
    forward exist): the oracle against a run of the reference's class on seeded weights."""
    g = gold("msred_featnet_fpn")
    from ada_mvs_amd.models.msrednet import FeatureNet
    sd = synth.seeded_state_dict(FeatureNet(8, 3, 4, "fpn"), seed=4)
    out = mo.feature_net_fpn(g["x"], sd, "")
    for k in ("stage1", "stage2", "stage3"):
        assert out[k].shape == g[k].shape
        assert rel_l1(out[k], g[k]) < 1e-6, k


def test_state_dict_keys_match_reference_layout():
    m, sd = tiny_model_state()
    keys = set(m.state_dict())
    assert set(sd) == keys
    for k in ("feature.conv0.0.conv.weight", "feature.deconv1.deconv.conv.weight", "feature.out3.weight",
              "cost_regularization.0.conv_gru4.gate_conv.bias", "cost_regularization.2.conv_gru1.output_norm.weight",
              "cost_regularization.1.upconv3.conv.weight", "cost_regularization.2.upconv2d.bias"):
        assert k in keys
    assert m.state_dict()["cost_regularization.0.conv_gru1.gate_conv.weight"].shape == (16, 40, 3, 3)
    assert m.state_dict()["cost_regularization.2.conv_gru1.gate_conv.weight"].shape == (16, 16, 3, 3)


# ---------------------------------------------------------------------------------------------------------------
# GPU: the HIP path (csrc/msred.hip + k_conv_dd) against the reference's fixtures and the oracle
# ---------------------------------------------------------------------------------------------------------------
def _cl(x):
    """[B,C,h,w] -> channel-last [B,h*w,C] on the GPU"""
    B, C, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, h * w, C).contiguous().cuda()


def _nchw(x_cl, h, w):
    B, _, C = x_cl.shape
    return x_cl.reshape(B, h, w, C).permute(0, 3, 1, 2).cpu()


@pytest.mark.gpu
@pytest.mark.parametrize("npix", [37 * 11, 640 * 480])         # 64 partial ranges / 150 of them
def test_group_stats_against_torch(npix):
    from ada_mvs_amd import hip_ops
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(3, npix, 48, generator=g) * 2 + 0.7).cuda()
    x1 = (torch.randn(3, npix, 48, generator=g) * 0.3 - 1.1).cuda()
    part = hip_ops.group_stats_workspace(3, 2, x0.device)
    hip_ops.group_stats_partial(x0, x1, 12, part)
    stats = hip_ops.group_stats_finish(part, 3, 2, npix, 12)
    again = hip_ops.group_stats_finish(part, 3, 2, npix, 12)
    assert torch.equal(stats, again)                            # fixed association: bit-identical
    for gi, x in enumerate((x0, x1)):
        sel = x[:, :, :12].double()
        mean = sel.mean(dim=(1, 2))
        rstd = 1.0 / torch.sqrt(sel.var(dim=(1, 2), unbiased=False) + 1e-5)
        assert torch.allclose(stats[:, gi, 0].double(), mean, rtol=1e-6, atol=1e-6)
        assert torch.allclose(stats[:, gi, 1].double(), rstd, rtol=1e-6)


@pytest.mark.gpu
def test_gru_cell2_against_reference_golden():
    """One ConvGRUCell2 (x 16, h 16 channels) through the split convolutions and the two fused epilogues."""
    from ada_mvs_amd import hip_ops, packing
    g = gold("msred_gru_cell")
    sd = cell_state_dict()
    x, h0 = g["x"], g["h"]
    B, _, hh, ww = x.shape
    dev = torch.device("cuda:0")
    wg, bg, wc, bc = sd["gate_conv.weight"], sd["gate_conv.bias"], sd["output_conv.weight"], sd["output_conv.bias"]
    pk = lambda w, b: packing.pack_padded_dd(w, b, 16).to(dev)          # noqa: E731
    conv = lambda inp, p, skip: hip_ops.conv3x3_dd(inp, p[:9 * 256], p[9 * 256:], skip, B, 16, hh, ww, 0, False)   # noqa: E731
    xm, state = _cl(x), _cl(h0)
    fr = conv(state, pk(wg[:16, 16:], None), conv(xm, pk(wg[:16, :16], bg[:16]), None))
    fu = conv(state, pk(wg[16:, 16:], None), conv(xm, pk(wg[16:, :16], bg[16:]), None))
    part = hip_ops.group_stats_workspace(B, 2, dev)
    hip_ops.group_stats_partial(fr, fu, 16, part)
    gn = torch.cat([sd[k].reshape(-1) for k in ("reset_gate_norm.weight", "reset_gate_norm.bias", "update_gate_norm.weight",
                                                "update_gate_norm.bias", "output_norm.weight", "output_norm.bias")]).to(dev)
    rh, u = torch.zeros_like(state), torch.zeros(B, hh * ww, 16, device=dev)
    hip_ops.gru2_gates_apply(fr, fu, part, gn, state, rh, u, 16)
    o = conv(rh, pk(wc[:, 16:], None), conv(xm, pk(wc[:, :16], bc), None))
    hip_ops.group_stats_partial(o, None, 16, part)
    out = torch.zeros(B, hh * ww, 32, device=dev)
    hip_ops.gru2_out_apply(o, part, gn[64:], u, state, out, 16)
    assert rel_l1(_nchw(state, hh, ww), g["out"]) < OP_TOL
    assert rel_l1(_nchw(out[:, :, :16].contiguous(), hh, ww), g["out"]) < OP_TOL and bool((out[:, :, 16:] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("baseline,B,h,w", [(8.0, 2, 24, 40), (150.0, 2, 24, 40),      # in bounds / out-of-bounds taps: plain kernel
                                            (8.0, 2, 176, 192), (150.0, 2, 176, 192)])   # >= 65536 pixels: the sweep form
def test_variance_cost_against_oracle(baseline, B, h, w):
    from ada_mvs_amd import hip_ops
    C, V = 16, 4
    feats = [synth.smooth_features(B, C, h, w, seed=10 + v) for v in range(V)]
    proj = synth.rig_projections(V, 4 * h, 4 * w, batch=B, baseline=baseline)["stage1"]
    planes = 400 + 200 * torch.rand(B, 3, h, w, generator=torch.Generator().manual_seed(1))
    rel = [mo.ao.relative_transform(proj[:, v], proj[:, 0]) for v in range(1, V)]
    feat_cl = torch.cat([_cl(f) for f in feats], 0)
    rt = hip_ops.relative_transforms(proj.cuda())
    a = torch.full((3 * B, h * w, 32), 7.0, device="cuda")
    b = torch.full((3 * B, h * w, 16), 7.0, device="cuda")
    hip_ops.red_variance_cost(feat_cl, rt, planes.reshape(B, 3, h * w).cuda(), a, b, B, V - 1, C, 3, h, w, negate=True)
    for d in range(3):                                             # maps are plane-major: index d * B + b
        want = mo.variance_cost(feats[0], feats[1:], [r[0] for r in rel], [r[1] for r in rel], planes[:, d:d + 1])
        assert rel_l1(-_nchw(a[d * B:(d + 1) * B, :, :C], h, w), want) < OP_TOL
        assert rel_l1(-_nchw(b[d * B:(d + 1) * B], h, w), want) < OP_TOL
    assert bool((a[:, :, C:] == 7.0).all())                      # the other channels are not touched
    # positive sign: the plain one-thread-per-plane kernel (the negated form above runs on the register-resident-tap sweep)
    hip_ops.red_variance_cost(feat_cl, rt, planes.reshape(B, 3, h * w).cuda(), a, None, B, V - 1, C, 3, h, w, negate=False)
    want = mo.variance_cost(feats[0], feats[1:], [r[0] for r in rel], [r[1] for r in rel], planes[:, 1:2])
    assert rel_l1(_nchw(a[B:2 * B, :, :C], h, w), want) < OP_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("concurrent", [False, True])
def test_slice_red_steps_against_reference_golden(concurrent):
    """Two consecutive planes through encoder / four recurrences / decoder: reg_cost and all four states."""
    from ada_mvs_amd.models.msrednet import slice_RED_Regularization
    g = gold("msred_slice_step")
    net = slice_RED_Regularization(32, 8)
    net.load_state_dict(slice_state_dict())
    net = net.cuda()
    net.concurrent_levels = concurrent
    dev = torch.device("cuda:0")
    net.packed(dev)
    B, C, h, w = g["cost0"].shape
    X0 = torch.zeros(2 * B, h * w, net.x_widths()[0], device=dev)       # what cost_maps() produces, from the fixture's cost
    for step in range(2):
        X0[step * B:(step + 1) * B, :, :C] = -_cl(g["cost%d" % step])
    fin, R = net.regularize_maps(X0, B, h, w)
    torch.cuda.synchronize()
    for step in range(2):
        reg = fin[step * B:(step + 1) * B, :, 0].reshape(B, 1, h, w).cpu()
        assert rel_l1(reg, g["reg%d" % step]) < OP_TOL
        for k in range(4):
            state = _nchw(R[k][step * B:(step + 1) * B, :, :net.HC[k]].contiguous(), h >> k, w >> k)
            assert rel_l1(state, g["state%d_%d" % (k + 1, step)]) < OP_TOL, "state %d step %d" % (k + 1, step)


@pytest.mark.gpu
def test_feature_net_unet_against_oracle():
    m, sd = tiny_model_state()
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(4))
    want = mo.feature_net_unet(x, {k[len("feature."):]: v for k, v in sd.items() if k.startswith("feature.")}, "")
    with torch.no_grad():
        got = m.feature(x.cuda())
    for k in ("stage1", "stage2", "stage3"):
        assert rel_l1(got[k].cpu(), want[k]) < 5e-5, k


@pytest.mark.gpu
def test_feature_net_fpn_against_reference_golden():
    """adamvs_feature_net_fpn through the module mirror: the reference fixture (2 images, 32 x 64) and, at a size with
    several tiles per map, the oracle; state-dict keys as the reference's class."""
    from ada_mvs_amd.models.msrednet import FeatureNet
    g = gold("msred_featnet_fpn")
    net = FeatureNet(8, 3, 4, "fpn")
    sd = synth.seeded_state_dict(net, seed=4)
    net.load_state_dict(sd)
    assert {"inner1.weight", "inner1.bias", "inner2.bias", "out2.weight", "out3.weight"} <= set(sd) and "deconv1.conv.conv.weight" not in sd
    net = net.cuda().eval()
    with torch.no_grad():
        got = net(g["x"].cuda())
    for k in ("stage1", "stage2", "stage3"):
        assert rel_l1(got[k].cpu(), g[k]) < OP_TOL, k
    x = torch.randn(3, 3, 96, 160, generator=torch.Generator().manual_seed(2))
    want = mo.feature_net_fpn(x, sd, "")
    with torch.no_grad():
        got = net(x.cuda())
        cl = net.forward_cl(x.cuda())
    for i, k in enumerate(("stage1", "stage2", "stage3")):
        assert rel_l1(got[k].cpu(), want[k]) < OP_TOL, k
        assert torch.equal(cl[i].reshape(3, want[k].shape[2], want[k].shape[3], -1).permute(0, 3, 1, 2), got[k])


@pytest.mark.gpu
def test_end_to_end_against_reference_golden_and_oracle():
    g = gold("msred_e2e_tiny")
    m, sd = tiny_model_state()
    m.load_state_dict(sd)
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    with torch.no_grad():
        out = m(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
    assert rel_l1(out["stage1"]["depth"].cpu(), g["depth_stage1"]) < E2E_TOL
    assert rel_l1(out["stage2"]["depth"].cpu(), g["depth_stage2"]) < E2E_TOL
    assert rel_l1(out["depth"].cpu(), g["depth"]) < E2E_TOL
    assert rel_l1(out["photometric_confidence"].cpu(), g["photometric_confidence"]) < E2E_TOL
    assert out["depth"].shape == (1, 64, 96) and out["stage1"]["depth"].shape == (1, 16, 24)


@pytest.mark.gpu
def test_end_to_end_batch_of_two_against_oracle():
    """B = 2 with different depth ranges per sample (the interval comes from sample 0, as in the reference)."""
    m, sd = tiny_model_state()
    m.load_state_dict(sd)
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs("tiny", batch=2, seed=3)
    dv[1] = torch.tensor([380.0, 640.0])
    c = synth.CONFIGS["tiny"]
    with mo.ao.use_grid_sample():
        want = mo.infer_cascade_rednet_forward(imgs, proj, dv, sd, c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    with torch.no_grad():
        out = m(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
    assert rel_l1(out["depth"].cpu(), want["depth"]) < E2E_TOL
    assert rel_l1(out["photometric_confidence"].cpu(), want["photometric_confidence"]) < E2E_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("fold,small_grid", [("0", "1024"), ("0", "0")])
def test_unfolded_recurrence_paths_in_a_child_process(fold, small_grid):
    """At one or two samples the GRU levels run with their elementwise kernels folded into the convolutions (two dependent
    launches per plane, csrc/msred.hip); the four-launch form (epilogue partial sums, one launch for both gate
    convolutions) serves larger batches, and the seven-launch form (k_gn_partial, plain convolutions) maps whose partial sums
    would not fit.  A child process, its options seeded from the environment (ADAMVS_<NAME>, read once when the option table is
    first touched: include/adamvs_hip.h "OPTIONS"), runs the end-to-end and slice-step cases with
    folding off, and with the small-grid kernels off as well (ADAMVS_CONV_SMALL_GRID=0: every convolution on the generic
    kernel, every reduction a launch of its own)."""
    import subprocess
    env = dict(os.environ, ADAMVS_RED_FOLD_APPLIES=fold, ADAMVS_CONV_SMALL_GRID=small_grid)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "end_to_end_against_reference_golden_and_oracle or end_to_end_batch_of_two or slice_red_steps"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_cpu_tensors_raise():
    m, _ = tiny_model_state()
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    from ada_mvs_amd._lib import AdaMVSHipError
    with pytest.raises(AdaMVSHipError, match="MI355X"):
        m(imgs, proj, dv)


@pytest.mark.gpu
@pytest.mark.parametrize("D,h,w,relu", [(64, 12, 24, False), (64, 20, 36, True), (32, 6, 10, False), (32, 48, 96, True),
                                        (64, 48, 96, False)])
def test_conv3x3_dd_small_grid_form_against_torch(D, h, w, relu):
    """adamvs_conv3x3_dd on few workgroups takes the register/LDS-resident kernel: same results as the streaming one."""
    from ada_mvs_amd import hip_ops, packing
    g = torch.Generator().manual_seed(D + h)
    N = 2
    wt = torch.randn(D, D, 3, 3, generator=g) * (2.0 / (9 * D)) ** 0.5
    bias = torch.randn(D, generator=g) * 0.1
    x = torch.randn(N, D, h, w, generator=g)
    skip = torch.randn(N, D, h, w, generator=g)
    want = torch.nn.functional.conv2d(x, wt, bias, padding=1)
    want = (torch.relu(want) if relu else want) + skip
    pk = packing.pack_reg_layer(wt, torch.ones(D), bias, False).cuda()
    got = hip_ops.conv3x3_dd(_cl(x), pk[:9 * D * D], pk[9 * D * D:], _cl(skip), N, D, h, w, 0, relu)
    assert rel_l1(_nchw(got, h, w), want) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("CA,CB,cout,h,w", [(32, 8, 16, 16, 24), (16, 8, 8, 22, 38), (8, 8, 16, 64, 96), (16, 16, 32, 12, 20),
                                             (16, 16, 16, 30, 50)])
def test_conv3x3_pair_against_torch(CA, CB, cout, h, w):
    """conv3x3(cat(a, b)) + bias with register-resident weights (levels 1, 2 of the MS-REDNet cells), ragged sizes too."""
    from ada_mvs_amd import hip_ops, packing
    g = torch.Generator().manual_seed(CA + cout + h)
    B = 2
    wt = torch.randn(cout, CA + CB, 3, 3, generator=g) * (2.0 / (9 * (CA + CB))) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    a, b = torch.randn(B, CA, h, w, generator=g), torch.randn(B, CB, h, w, generator=g)
    want = torch.nn.functional.conv2d(torch.cat((a, b), 1), wt, bias, padding=1)
    rows = packing.pad16(cout)
    got = hip_ops.conv3x3_pair(_cl(a), _cl(b), packing.pack_small_conv(wt).cuda(), packing.pad_bias(bias, rows).cuda(), cout, h, w)
    assert got.shape == (B, h * w, cout) and rel_l1(_nchw(got, h, w), want) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("views", [2, 6])
def test_end_to_end_other_view_counts_against_oracle(views):
    """One source view (variance of two samples) and five: the cost kernel's view loop, against the oracle."""
    c = dict(synth.CONFIGS["tiny"], views=views)
    from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet
    m = Infer_CascadeREDNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=1)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(c, batch=1, seed=2)
    with mo.ao.use_grid_sample():
        want = mo.infer_cascade_rednet_forward(imgs, proj, dv, sd, c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    with torch.no_grad():
        out = m(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
    assert rel_l1(out["depth"].cpu(), want["depth"]) < E2E_TOL
    assert rel_l1(out["photometric_confidence"].cpu(), want["photometric_confidence"]) < E2E_TOL


@pytest.mark.gpu
def test_error_behaviour_of_the_new_entry_points():
    from ada_mvs_amd import hip_ops
    from ada_mvs_amd._lib import AdaMVSHipError
    from ada_mvs_amd.models.msrednet import slice_RED_Regularization
    dev = torch.device("cuda:0")
    net = slice_RED_Regularization(32, 8)
    net.load_state_dict(slice_state_dict())
    net = net.cuda()
    with pytest.raises(AdaMVSHipError, match="multiple of 8"):           # three stride-2 levels
        net.regularize_maps(torch.zeros(1, 12 * 20, 32, device=dev), 1, 12, 20)
    a, b = torch.zeros(1, 64, 24, device=dev), torch.zeros(1, 64, 8, device=dev)
    with pytest.raises(AdaMVSHipError, match="conv3x3_pair"):             # 24 + 8 channels: not a supported pairing
        hip_ops.conv3x3_pair(a, b, torch.zeros(9 * 32 * 16, device=dev), torch.zeros(16, device=dev), 16, 8, 8)
    x = torch.zeros(2, 16, 30, device=dev)
    with pytest.raises(AdaMVSHipError, match="group_stats_partial"):      # 30-wide rows are not float4-addressable
        hip_ops.group_stats_partial(x, None, 8, hip_ops.group_stats_workspace(2, 1, dev))
    with pytest.raises(AdaMVSHipError, match="GPU tensor"):
        hip_ops.red_variance_cost(torch.zeros(2, 4, 8), torch.zeros(1, 1, 12, device=dev), torch.zeros(1, 1, 4, device=dev),
                                  torch.zeros(1, 4, 8, device=dev), None, 1, 1, 8, 1, 2, 2)
    from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet
    with pytest.raises(AdaMVSHipError, match="share_cr"):
        Infer_CascadeREDNet(16, [16, 8, 4], [4, 2, 1], share_cr=True)
    from ada_mvs_amd.models.msrednet import FeatureNet
    fpn = FeatureNet(8, 3, 4, "fpn")
    fpn.load_state_dict(synth.seeded_state_dict(fpn, seed=4))
    fpn = fpn.cuda().eval()
    with pytest.raises(AdaMVSHipError, match="multiples of 32"):           # the C entry point itself; the module falls back to torch ops
        hip_ops.feature_net_fpn(torch.zeros(1, 3, 40, 64, device=dev), fpn.packed(dev))
    with pytest.raises(AdaMVSHipError, match="feature_net_fpn"):           # 4 channels
        hip_ops.feature_net_fpn(torch.zeros(1, 4, 32, 64, device=dev), fpn.packed(dev))
    with pytest.raises(AssertionError):
        FeatureNet(8, 3, 4, "resnet")


def test_regulariser_packing_layout():
    """pack_red_regularization: every block at a 64-float offset; the padded D x D blocks hold the real weights at the
    documented fragment positions and zeros elsewhere; the stride-1 ConvTranspose2d enters as its mirrored convolution."""
    from ada_mvs_amd import packing
    sd = slice_state_dict(C=16, seed=5)
    flat, off = packing.pack_red_regularization(sd, "", 16)
    assert all(o % 64 == 0 for o, _ in off.values())
    for name in ("gxr3", "ghu4", "cx3", "conv2", "upconv3", "upconv2d", "gp1", "cp2", "gn4"):
        assert name in off

    def frag(name, tap, co, ci):                       # value of W[co][ci][tap] in a padded block
        o, D = off[name]
        tile, r, kc, k4 = co // 16, co % 16, ci // 4, ci % 4
        return float(flat[o + ((tap * (D // 4) + kc) * (D // 16) + tile) * 64 + k4 * 16 + r])

    wg = sd["conv_gru3.gate_conv.weight"]              # [64][32 + 32][3][3]: reset rows 0-31, x columns 0-31
    assert off["gxr3"][1] == 32 and off["ghr3"][1] == 32
    assert frag("gxr3", 4, 5, 7) == float(wg[5, 7, 1, 1]) and frag("ghr3", 2, 30, 3) == float(wg[30, 32 + 3, 0, 2])
    assert frag("gxu3", 0, 1, 2) == float(wg[32 + 1, 2, 0, 0])
    o, D = off["gxr3"]
    assert torch.equal(flat[o + 9 * D * D:o + 9 * D * D + D], sd["conv_gru3.gate_conv.bias"][:32])        # bias with the x half
    o, D = off["ghr3"]
    assert bool((flat[o + 9 * D * D:o + 9 * D * D + D] == 0).all())                                      # none with the h half
    w1 = sd["conv1.conv.weight"]                       # [16][16][3][3] at width 16
    assert off["conv1"][1] == 16 and frag("conv1", 8, 15, 15) == float(w1[15, 15, 2, 2])
    wt = sd["upconv2d.weight"]                         # ConvTranspose2d [cin 8][cout 1][3][3] -> conv [1][8] with mirrored taps
    assert frag("upconv2d", 0, 0, 3) == float(wt[3, 0, 2, 2]) and frag("upconv2d", 5, 0, 7) == float(wt[7, 0, 1, 0])
    assert frag("upconv2d", 4, 1, 0) == 0.0 and frag("upconv2d", 4, 0, 8) == 0.0                          # padding
    o, D = off["upconv2d"]
    assert float(flat[o + 9 * D * D]) == float(sd["upconv2d.bias"][0])
    gn_o, hc = off["gn2"]
    assert hc == 16 and torch.equal(flat[gn_o + 4 * hc:gn_o + 5 * hc], sd["conv_gru2.output_norm.weight"])


@pytest.mark.gpu
@pytest.mark.parametrize("recipe", ["default", "sharp"])
def test_end_to_end_at_the_benchmark_shape_against_oracle(recipe):
    """One tile at the shape bench.py --model msrednet is quoted on (5 views, 768 x 384, hypotheses 192/64/8: 264 recurrent
    planes through four GRU levels, 96 x 192 ... 12 x 24 maps at stage 1) against oracle/msrednet_oracle.py (itself pinned by
    the reference-run fixtures above), evaluated twice: in the reference's fp32 and in float64.

    What this case shows (round 4): the confidence maps of this network are ill-conditioned at this size with the seeded weights.
    The fp32 CPU path -- the reference's own arithmetic -- is 2.5e-4 / 8.6e-4 / 3.5e-3 away from the float64 evaluation on the
    three stages' confidence maps (depth maps: 2-5e-5), although a different thread count or form of the warp moves it by
    8e-5 ... 2e-4 only; the HIP path, with other summation orders in every convolution, is 2.6e-4 / 8.8e-4 / 3.6e-3 from the
    fp32 run and 1.7e-4 / 5.8e-4 / 2.3e-3 from float64: three evaluations of the same network, pairwise a few 1e-3 apart on the
    last confidence map, the HIP one the closest to exact arithmetic.  The bar (1e-3 relative L1, BASELINE.json) is asserted,
    against both oracles, on every map the reference's own fp32 arithmetic holds to 1e-3 of float64 (all but the stage-3
    confidence map on the seeded weights); the others are "reference-limited" and held to a recorded distance from float64
    (the block at the end of the test); every map must be no farther from float64 than the reference's fp32 arithmetic is.

    recipe "sharp" (round 5; synth.LOGIT_GAINS: gain 15 instead of 3 on upconv2d, the layer in front of the unstabilised exp): the same
    three evaluations on weights with a trained network's dynamic range -- whether the 3.6e-3 of the last confidence map belongs to
    the seeded recipe or to the network is read off the printed rows (INTEGRATION.md quotes them)."""
    from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet
    c = synth.CONFIGS["cfg3"]
    m = Infer_CascadeREDNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=0, recipe=recipe)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs("cfg3", batch=1, seed=0)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))

    def oracle(dt):
        sd_ = {k: (v.cpu().to(dt) if v.is_floating_point() else v.cpu()) for k, v in sd.items()}
        with torch.no_grad(), mo.ao.use_grid_sample():
            return mo.infer_cascade_rednet_forward(imgs.to(dt), {k: v.to(dt) for k, v in proj.items()}, dv.to(dt), sd_, c["num_depth"],
                                                   c["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    want32, want64 = oracle(torch.float32), oracle(torch.float64)
    with torch.no_grad():
        out = m(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
    rows = {}
    for s in ("stage1", "stage2", "stage3"):
        for key in ("depth", "photometric_confidence"):
            assert out[s][key].shape == want32[s][key].shape
            rows[s + "." + key] = (rel_l1(out[s][key].cpu(), want32[s][key]), rel_l1(out[s][key].cpu().double(), want64[s][key]),
                                   rel_l1(want32[s][key].double(), want64[s][key]))
    print("msrednet cfg3-shape parity (hip vs fp32 oracle, hip vs float64 oracle, fp32 oracle vs float64):",
          {k: "%.2e %.2e %.2e" % v for k, v in rows.items()})
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                               "parity_msrednet_full_size%s.json" % ("" if recipe == "default" else "_" + recipe)), "w") as f:
            import json
            json.dump({k: {"hip_vs_fp32_oracle": v[0], "hip_vs_float64_oracle": v[1], "fp32_oracle_vs_float64": v[2]} for k, v in rows.items()}, f, indent=1)
    except OSError:
        pass
    # Every map is held to an ABSOLUTE bar (round 6).  Where the reference's own fp32 arithmetic is well-conditioned -- its CPU evaluation
    # within 1e-3 of float64 -- that bar is the north star's: 1e-3 against the fp32 oracle AND against float64.  Where it is not (the
    # reference's fp32 run is itself further than 1e-3 from exact arithmetic: "reference-limited"), no evaluation of the network in
    # fp32 can be held to 1e-3 of another; there the HIP path is held to a RECORDED distance from float64 (1.3 x what round 5 measured,
    # profiles/r05_parity_msrednet_full_size*.json) and must not be farther from float64 than the reference's fp32 evaluation.
    REFERENCE_LIMITED = {"default": {"stage3.photometric_confidence": 3.0e-3},
                         "sharp": {"stage2.depth": 1.3e-3, "stage2.photometric_confidence": 6.8e-3, "stage3.depth": 4.3e-3,
                                   "stage3.photometric_confidence": 0.22}}[recipe]
    for k, (e32, e64, o) in rows.items():
        three = "%s: hip vs fp32 oracle %.2e, hip vs float64 %.2e, fp32 oracle vs float64 %.2e" % (k, e32, e64, o)
        if o < 1e-3:
            assert k not in REFERENCE_LIMITED, "the reference is well-conditioned here now: hold the map to 1e-3 (" + three + ")"
            assert e32 < 1e-3 and e64 < 1e-3, three          # the bar, against the reference's fp32 and against exact arithmetic
        else:
            assert k in REFERENCE_LIMITED, "reference-limited map without a recorded bound (" + three + ")"
            assert e64 < REFERENCE_LIMITED[k], "reference-limited; " + three
        assert e64 <= 1.05 * o, three                        # no farther from exact arithmetic than the reference's fp32 evaluation is
        assert e32 < 2.0 * o + 1e-4, three                   # and from the reference no farther than two such distances
