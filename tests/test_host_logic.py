"""CPU-only checks of the host side: the C-ABI library loads and exports every
symbol include/adamvs_hip.h declares, weight packing layouts, state-dict
contract, loud failure without a GPU.  No compute calls."""
import ctypes
import os
import re

import pytest
import torch

import ada_mvs_amd  # noqa: F401
from ada_mvs_amd import _lib, packing, synth
from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "adamvs_hip.h")).read()
    declared = set(re.findall(r"\b(adamvs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.adamvs_version() == _lib.ABI_VERSION
    assert ctypes.sizeof(_lib.FuseWeights) == 17 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.StageDesc) == 72      # 14 ints + half_span (60), padded to the pointer half_span_dev (64 + 8)


def test_every_option_is_documented_in_the_header_with_its_default():
    """The option table (csrc/options.h) against the OPTIONS section of include/adamvs_hip.h: same names, same defaults, in both
    directions; set / get round trip; an unknown name is an argument error; the compute path has one place that reads the
    environment for them (the seeding loop) plus the recurrence's two tuning tables."""
    hdr = open(os.path.join(ROOT, "include", "adamvs_hip.h")).read()
    sec = hdr[hdr.index("---- OPTIONS"):hdr.index("int adamvs_option_count")]
    documented = {m.group(1): int(m.group(2)) for m in re.finditer(r"^ \*   ([a-z0-9_]+)\s+(-?\d+)\s{2,}\S", sec, re.M)}
    lib = _lib.load()
    table = {}
    for name in _lib.option_names():
        v = ctypes.c_int(0)
        assert lib.adamvs_option_default(name.encode(), ctypes.byref(v)) == 0
        table[name] = v.value
    assert lib.adamvs_option_name(lib.adamvs_option_count()) is None
    assert documented == table, (set(documented) ^ set(table), {k: (documented[k], table[k]) for k in documented if k in table and documented[k] != table[k]})
    with _lib.options(gru_wino=5, recur_mode=3):
        assert _lib.get_option("gru_wino") == 5 and lib.adamvs_gru_wino_mask() == 5
        assert lib.adamvs_recurrence_schedule(0, 10 ** 9) == 3
    assert _lib.get_option("gru_wino") == table["gru_wino"] and lib.adamvs_recurrence_schedule(0, 10 ** 9) == 0
    with pytest.raises(_lib.AdaMVSHipError, match="unknown option 'no_such'"):
        _lib.set_option("no_such", 1)
    csrc = os.path.join(ROOT, "ada-mvs_amd", "csrc")
    sites = sum(open(os.path.join(csrc, f)).read().count("getenv(") for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    assert sites <= 3, sites


def test_argument_errors_surface_as_exceptions_without_a_gpu():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    rc = lib.adamvs_pack_features(null, null, 1, 6, 4, 4, null)      # rejected before any launch
    assert rc < 0
    with pytest.raises(_lib.AdaMVSHipError, match="pack_features"):
        _lib.check(rc, "pack_features")
    desc = _lib.StageDesc(1, 2, 12, 8, 8, 16, 1, 1, 0, 0, 0, 0, 0)         # C=12 unsupported
    assert lib.adamvs_depth_stage_workspace_bytes(ctypes.byref(desc)) == 0
    assert b"C=12" in lib.adamvs_last_error_string()
    desc = _lib.StageDesc(8, 4, 32, 96, 192, 192, 1, 1, 0, 0, 0, 0, 0)     # cfg2 stage 1, 8 tiles
    assert lib.adamvs_depth_stage_workspace_bytes(ctypes.byref(desc)) > (1 << 30)


def test_cpu_tensors_are_rejected_not_silently_computed():
    c = synth.CONFIGS["tiny"]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8]).eval()
    imgs, proj, dv = synth.tile_inputs("tiny")
    with pytest.raises(_lib.AdaMVSHipError, match="no CPU fallback"):
        m(imgs, proj, dv)
    from ada_mvs_amd.models.module import homo_warping_float
    with pytest.raises(_lib.AdaMVSHipError):
        homo_warping_float(torch.zeros(1, 8, 4, 4), torch.eye(4)[None], torch.eye(4)[None], torch.ones(1, 1, 4, 4))


def test_state_dict_contract():
    m = Infer_AdaMVSNet(192, [48, 32, 8], [4, 2, 1], False, [8, 8, 8])
    sd = m.state_dict()
    assert len(sd) == 339
    assert sd["DepthNet.0.reg.conv7.0.weight"].shape == (48, 48, 3, 3)
    assert sd["DepthNet.1.reg_fuse.conv1.conv.weight"].shape == (8, 16, 3, 3)
    assert sd["DepthNet.2.reg_fuse.upconv2d.weight"].shape == (1, 8, 3, 3)
    assert sd["DepthNet.0.reg_fuse.upconv2d.weight"].shape == (8, 1, 3, 3)
    assert sd["DepthNet.0.reg_fuse.conv_gru2.conv_gates.0.weight"].shape == (32, 32, 3, 3)
    # 'module.'-prefixed checkpoints (predict_whu.py:82-89) load through DataParallel
    dp = torch.nn.DataParallel(m)
    dp.load_state_dict({"module." + k: v for k, v in synth.seeded_state_dict(m, 0).items()})


def test_fragment_packing_layout():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(24, 8, 3, 3, generator=g)                        # 24 couts -> 2 tiles, second half padded
    pk = packing.pack_small_conv(w).reshape(2, 9, 2, 64)
    for nt, t, kc, lane in ((0, 0, 0, 0), (0, 4, 1, 37), (1, 8, 1, 7), (1, 3, 0, 63)):
        co, ci = nt * 16 + (lane & 15), 4 * kc + (lane >> 4)
        expect = w[co, ci, t // 3, t % 3] if co < 24 else 0.0
        assert float(pk[nt, t, kc, lane]) == float(expect)
    wt = torch.randn(16, 8, 3, 3, generator=g)                       # ConvTranspose2d layout [cin][cout]
    pkt = packing.pack_small_conv(wt, transposed=True).reshape(1, 9, 4, 64)
    assert float(pkt[0, 5, 2, 21]) == float(wt[4 * 2 + 1, 5, 1, 2])
    # two-row conv1: rows 0-7 = output row y (ky = rr), rows 8-15 = output row y+1 (ky = rr-1)
    w1 = torch.randn(8, 16, 3, 3, generator=g)
    p2 = packing.pack_conv1_two_row(w1).reshape(4, 3, 4, 64)
    assert float(p2[1, 2, 3, 2 * 16 + 5]) == float(w1[5, 14, 1, 2])         # lane: k4=2, row 5  -> cin 3*4+2
    assert float(p2[1, 2, 3, 2 * 16 + 13]) == float(w1[5, 14, 0, 2])        # row 13 = channel 5 of row y+1, ky = 0
    assert float(p2[0, 0, 0, 8]) == 0.0 and float(p2[3, 1, 1, 3]) == 0.0
    # CostRegNet2D layer: [tap][kc][tile][lane], BN scale folded
    D = 32
    wl = torch.randn(D, D, 3, 3, generator=g)
    scale = torch.rand(D, generator=g) + 0.5
    pl = packing.pack_reg_layer(wl, scale, torch.zeros(D), False)
    frag = pl[:9 * D * D].reshape(9, D // 4, D // 16, 64)
    assert torch.isclose(frag[7, 5, 1, 50], wl[16 + (50 & 15), 20 + (50 >> 4), 2, 1] * scale[16 + (50 & 15)])
    assert pl.numel() == 9 * D * D + D


def test_split_bf16_packing_layout():
    g = torch.Generator().manual_seed(1)
    D = 64
    w = torch.randn(D, D, 3, 3, generator=g)
    scale = torch.rand(D, generator=g) + 0.5
    pk = packing.pack_reg_layer_bf16x3(w, scale, torch.zeros(D), False)
    assert pk.numel() == 9 * D * D + D                                  # same size as the fp32 packing
    frag = pk[:9 * D * D].view(torch.bfloat16).reshape(2, 9, D // 32, D // 16, 64, 8)
    ws = w * scale.reshape(-1, 1, 1, 1)
    hi, lo = packing.split_bf16(ws)
    tap, kb, tile, lane, j = 5, 1, 2, 37, 6
    co, ci = 16 * tile + (lane & 15), 32 * kb + 8 * (lane >> 4) + j
    assert frag[0, tap, kb, tile, lane, j] == hi[co, ci, tap // 3, tap % 3]
    assert frag[1, tap, kb, tile, lane, j] == lo[co, ci, tap // 3, tap % 3]
    assert float((hi.float() + lo.float() - ws).abs().max() / ws.abs().max()) < 2 ** -15


def test_winograd_packing_evaluates_the_convolution():
    """pack_reg_layer_wino: U = G w G^T in the fragment order of include/adamvs_hip.h; evaluating Y = At[(U . V)]A with
    V = Bt d B from the PACKED array reproduces the 3x3 convolution (the arithmetic of csrc/costreg2d_wino.hip, on the CPU)."""
    D, h, w = 64, 6, 8
    g = torch.Generator().manual_seed(5)
    wt = torch.randn(D, D, 3, 3, generator=g, dtype=torch.float64)
    scale = torch.rand(D, generator=g, dtype=torch.float64) + 0.5
    x = torch.randn(1, D, h, w, generator=g, dtype=torch.float64)
    pk = packing.pack_reg_layer_wino(wt.float(), scale.float())
    assert pk.numel() == 16 * D * D
    frag = pk.reshape(D // 4, 4, D // 16, 64, 4).double()                 # [kc][i][tile][lane][j]
    u = torch.zeros(4, 4, D, D, dtype=torch.float64)                        # [i][j][co][ci]
    for lane in range(64):
        co16, k4 = lane & 15, lane >> 4
        u[:, :, co16::16, k4::4] = frag[:, :, :, lane, :].permute(1, 3, 2, 0)   # [i][j][tile][kc]
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    xp = torch.nn.functional.pad(x[0], (1, 1, 1, 1))
    out = torch.zeros(D, h, w, dtype=torch.float64)
    for ty in range(h // 2):
        for tx in range(w // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                 # [ci][4][4]
            v = torch.einsum("ik,ckl,jl->ijc", Bt, d, Bt)
            m = torch.einsum("ijoc,ijc->ijo", u, v)
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = torch.einsum("ai,ijo,bj->oab", At, m, At)
    ref = torch.nn.functional.conv2d(x, wt * scale.reshape(-1, 1, 1, 1), padding=1)[0]
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-6         # U is rounded to fp32 once
    # a network blob carries the five stride-1 layers in this form behind the 11 direct blocks (fp32, supported widths only)
    from ada_mvs_amd.models.adamvs import CostRegNet2D
    sd = synth.seeded_state_dict(CostRegNet2D(64), 0)
    assert packing.pack_cost_reg_net_2d(sd, "").numel() == 11 * (9 * 64 * 64 + 64) + 5 * 16 * 64 * 64
    assert packing.pack_cost_reg_net_2d(sd, "", "bf16x3").numel() == 11 * (9 * 64 * 64 + 64)


def test_packed_network_sizes_match_the_header():
    m = Infer_AdaMVSNet(48, [48, 32, 8], [4, 2, 1], False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, 0)
    reg = packing.pack_cost_reg_net_2d(sd, "DepthNet.0.reg.")
    assert reg.numel() == 11 * (9 * 48 * 48 + 48)
    flat, off = packing.pack_slice_reg_net(sd, "DepthNet.0.reg_fuse.")
    assert list(off) == list(packing.FUSE_FIELDS)
    assert all(o % 64 == 0 for o in off.values())
    assert off["gates1"] - off["conv1"] == 12 * 8 * 64               # two-row conv1, C=32 -> KC=8
    assert flat.numel() >= off["final_w"] + 73
    # cand1 (8 outputs from cat(c1, r*h) = 16 channels) in the same two-row form: [12 = (rr, kx)][KC = 4][64]
    assert off["cand1_b"] - off["cand1"] == 12 * 4 * 64
    wc = sd["DepthNet.0.reg_fuse.conv_gru1.convc.0.weight"]          # [8][16][3][3]
    frag = flat[off["cand1"]:off["cand1"] + 12 * 4 * 64].reshape(4, 3, 4, 4, 16)      # [rr][kx][kc][k4][row]
    assert frag[1, 2, 3, 1, 5] == wc[5, 13, 1, 2]                     # rows 0-7: output row y, ky = rr
    assert frag[1, 2, 3, 1, 8 + 5] == wc[5, 13, 0, 2]                 # rows 8-15: output row y+1, ky = rr - 1
    assert frag[3, 0, 0, 0, 2] == 0 and frag[0, 0, 0, 0, 8 + 2] == 0  # no tap for (row y, rr = 3) and (row y+1, rr = 0)
    flat_b, off_b = packing.pack_slice_reg_net(sd, "DepthNet.0.reg_fuse.", "bf16x3")
    # split bf16 (ABI 15): cand1 two-row as well, k = (rr * 3 + kx) * 16 + cin = 192 = 6 k-blocks of [hi|lo][64 lanes][8 bf16]
    assert off_b["cand1_b"] - off_b["cand1"] == 2 * 6 * 64 * 8 // 2
    # last layer: tap-major (a channel pair of one tap is one 64-bit scalar operand of the packed FMA), bias at [72]
    w_up = sd["DepthNet.0.reg_fuse.upconv2d.weight"]                  # ConvTranspose2d: [8][1][3][3]
    fw = flat[off["final_w"]:off["final_w"] + 73]
    assert fw[(2 * 3 + 1) * 8 + 5] == w_up[5, 0, 2, 1] and fw[72] == sd["DepthNet.0.reg_fuse.upconv2d.bias"][0]
    flat3, off3 = packing.pack_slice_reg_net(sd, "DepthNet.2.reg_fuse.")
    w_fl = sd["DepthNet.2.reg_fuse.upconv2d.weight"]                  # Conv2d: [1][8][3][3]
    assert flat3[off3["final_w"] + (0 * 3 + 2) * 8 + 3] == w_fl[0, 3, 0, 2]


def test_feature_net_packing_layout():
    """FeatureNet0 packing (include/adamvs_hip.h: adamvs_feature_weights): BatchNorm folded, fragment order, the
    transposed layers stored per output parity class, the output convolutions split into feature and context columns."""
    from ada_mvs_amd.models.adamvs import FeatureNet0
    net = FeatureNet0(8)
    sd = synth.seeded_state_dict(net, 3)
    flat, off = packing.pack_feature_net(sd, "")
    assert all(o % 64 == 0 for o in off.values())
    assert set(off) == {n + s for n in packing.FEATURE_CONVS for s in (".w", ".b")} | \
        {n + s for n in packing.FEATURE_BRANCHES for s in (".w1", ".b1", ".w2")}
    # conv1.0: 5x5 stride 2, 8 -> 16: fragment (tap, kc), lane l = W[l & 15][4 kc + (l >> 4)][tap] * bn scale
    w = sd["conv1.0.conv.weight"]
    scale = sd["conv1.0.bn.weight"] / torch.sqrt(sd["conv1.0.bn.running_var"] + packing.BN_EPS)
    tap, kc, lane = 7, 1, 37
    co, ci = lane & 15, 4 * kc + (lane >> 4)
    got = flat[off["conv1_0.w"] + (tap * 2 + kc) * 64 + lane]
    assert torch.allclose(got, w[co, ci, tap // 5, tap % 5] * scale[co], rtol=1e-6, atol=0)
    shift = sd["conv1.0.bn.bias"] - sd["conv1.0.bn.running_mean"] * scale
    assert torch.allclose(flat[off["conv1_0.b"] + 5], shift[5], rtol=1e-6, atol=1e-8)
    # conv0.0 pads RGB to 4 input channels: the lanes of the fourth k-row hold zeros
    assert float(flat[off["conv0_0.w"]:off["conv0_0.w"] + 9 * 64].reshape(9, 4, 16)[:, 3].abs().max()) == 0.0
    # deconv2.deconv (ConvTranspose2d 16 -> 8): class (1,1) = fragments 5..8, tap (ty,tx) = kernel index (ty ? 0 : 2, tx ? 0 : 2)
    wt = sd["deconv2.deconv.conv.weight"]                              # [cin 16][cout 8][3][3]
    sc = sd["deconv2.deconv.bn.weight"] / torch.sqrt(sd["deconv2.deconv.bn.running_var"] + packing.BN_EPS)
    frag, kc, lane = 5 + (1 * 2 + 0), 2, 19                            # class 11, (ty, tx) = (1, 0) -> (ky, kx) = (0, 2)
    co, ci = lane & 15, 4 * kc + (lane >> 4)
    got = flat[off["deconv2_t.w"] + (frag * 4 + kc) * 64 + lane]
    assert torch.allclose(got, wt[ci, co, 0, 2] * sc[co], rtol=1e-6, atol=0)
    # out3 = [8][16][1][1]: columns 0-3 branch 1, 4-7 branch 2, 8-15 the feature map
    wo = sd["out3.weight"].reshape(8, 16)
    assert torch.equal(flat[off["br3_2.w2"]:off["br3_2.w2"] + 32].reshape(8, 4), wo[:, 4:8])
    assert flat[off["out3.w"] + (0 * 2 + 1) * 64 + 16 * 2 + 3] == wo[3, 8 + 4 * 1 + 2]


def test_synthetic_recipes_are_deterministic():
    a = synth.tile_inputs("tiny", batch=2, seed=3)
    b = synth.tile_inputs("tiny", batch=2, seed=3)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1]["stage2"], b[1]["stage2"])
    p = a[1]
    assert torch.allclose(p["stage1"][:, :, :2] * 4, p["stage3"][:, :, :2]) and torch.equal(p["stage1"][:, :, 2:], p["stage3"][:, :, 2:])


def test_reference_style_import_path():
    """predict_whu.py:72 does `from models.adamvs import Infer_AdaMVSNet`: works with ada-mvs_amd/ on sys.path."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import ada_mvs_amd; "
            "from models.adamvs import Infer_AdaMVSNet; from models.module import homo_warping_float, depth_regression; "
            "m = Infer_AdaMVSNet(num_depth=192, ndepths=[48, 32, 8], depth_intervals_ratio=[4.0, 2.0, 1.0], share_cr=False, "
            "cr_base_chs=[8, 8, 8]); print(len(m.state_dict()))") % (ROOT, os.path.join(ROOT, "ada-mvs_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith("339")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No libadamvs_hip.so -> AdaMVSHipError naming the build command; nothing falls back to PyTorch or the oracle."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libadamvs_hip.so"))
    with pytest.raises(_lib.AdaMVSHipError, match="not built"):
        _lib.load()
    from ada_mvs_amd import hip_ops
    with pytest.raises(_lib.AdaMVSHipError):                    # every op goes through load(): same failure
        hip_ops.feature_net0_workspace_bytes(1, 64, 96)


def _emulated_replica(model):
    """What torch.nn.parallel.replicate() hands nn.DataParallel for one device (torch/nn/parallel/replicate.py): every
    module shallow-copied, `_parameters` emptied, the per-device parameter copies re-attached as plain attributes and
    listed in `_former_parameters`.  On CPU the 'device copy' is a clone."""
    from collections import OrderedDict
    modules = list(model.modules())
    copies = {m: m._replicate_for_data_parallel() for m in modules}
    for m, r in copies.items():
        r._former_parameters = OrderedDict()
    for m, r in copies.items():
        for key, child in m._modules.items():
            if child is not None:
                setattr(r, key, copies[child])
        for key, p in m._parameters.items():
            if p is not None:
                c = p.detach().clone()
                setattr(r, key, c)
                r._former_parameters[key] = c
        for key, b in m._buffers.items():
            if b is not None:
                setattr(r, key, b.clone())
    return copies[model]


def test_packing_works_on_data_parallel_replicas():
    """reference predict_whu.py:82 wraps the model in nn.DataParallel; with two or more visible GPUs every forward
    runs on replicas whose state_dict() holds buffers only.  Packing must read the replica's weights all the same,
    give the very same packed stream as the source module, and land in the cache shared with the source module."""
    from ada_mvs_amd.models.module import module_state
    m = Infer_AdaMVSNet(16, [16, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    rep = _emulated_replica(m)
    assert len(rep.state_dict()) < len(sd)                                   # what broke round 1's packed(): buffers only
    got = module_state(rep)
    assert set(got) == set(sd)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    ref_reg = packing.pack_cost_reg_net_2d(sd, "DepthNet.0.reg.", "fp32")
    assert torch.equal(packing.pack_cost_reg_net_2d(module_state(rep.DepthNet[0].reg), "", "fp32"), ref_reg)
    flat_ref, _ = packing.pack_slice_reg_net(sd, "DepthNet.1.reg_fuse.", "fp32")
    flat_rep, _ = packing.pack_slice_reg_net(module_state(rep.DepthNet[1].reg_fuse), "", "fp32")
    assert torch.equal(flat_rep, flat_ref)
    f_ref, _ = packing.pack_feature_net(sd, "feature.")
    f_rep, _ = packing.pack_feature_net(module_state(rep.feature), "")
    assert torch.equal(f_rep, f_ref)
    # one cache and one workspace table per source module, shared by its replicas and keyed by device
    assert rep.DepthNet[0].reg._cache is m.DepthNet[0].reg._cache
    assert rep.DepthNet[0]._workspace is m.DepthNet[0]._workspace
    key = (torch.device("cpu"), "fp32")
    rep.DepthNet[0].reg.cached(key, lambda: "packed on the replica")
    assert m.DepthNet[0].reg._cache[key] == "packed on the replica"
    m.load_state_dict(sd)                                                    # loading drops every cache
    assert not m.DepthNet[0].reg._cache and not m.feature._cache


def test_inference_model_refuses_train_mode():
    m = Infer_AdaMVSNet(16, [16, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    assert m.training
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    with pytest.raises(_lib.AdaMVSHipError):
        m(imgs, proj, dv)


def test_share_cr_and_cr_base_chs_are_accepted_and_ignored():
    """Quirk Q8 (reference adamvs.py:537-565): `share_cr` and `cr_base_chs` are constructor arguments of Infer_AdaMVSNet
    that change nothing -- three DepthNets with GRU widths 8 / 16 whatever is passed; the mirror keeps the same
    state-dict layout so that reference checkpoints load either way."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    a = Infer_AdaMVSNet(48, [48, 32, 8], [4.0, 2.0, 1.0], False, [8, 8, 8]).state_dict()
    b = Infer_AdaMVSNet(48, [48, 32, 8], [4.0, 2.0, 1.0], True, [4, 16, 2]).state_dict()
    assert list(a) == list(b) and len(a) == 339
    assert all(a[k].shape == b[k].shape for k in a)


def test_bench_refuses_a_world_that_is_not_gpus():
    """bench.py --gpus 2 inside a 1-rank world (WORLD_SIZE=1 set by some launcher) must exit 2, not print an n_gpus = 1 line;
    and with a rank that fails (no GPU here) the self-launcher must pass the failure on."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2 but 1 rank" in r.stderr and not r.stdout.strip()


def test_cost_reg_width_table_and_padded_packing():
    """packing.REG_WIDTHS is the library's table (csrc/costreg2d.hip::costreg_width); a network of D hypotheses is packed at the
    next width with zero filters and PAD_SCORE on `prob`'s pad channels, and the blob has the length the library expects."""
    from ada_mvs_amd.models.adamvs import CostRegNet2D
    lib = _lib.load()
    for prec, code in (("fp32", 0), ("bf16x3", 1)):
        for d in list(range(1, 530, 7)) + [16, 32, 48, 64, 96, 128, 192, 256, 384, 385, 512, 513]:
            want = lib.adamvs_cost_reg_width(d, code)
            if d <= 512:
                assert packing.reg_width(d, prec) == want and want >= d, (d, prec)
            else:
                assert want == 0
                with pytest.raises(ValueError):
                    packing.reg_width(d, prec)
    for d, prec, code in ((40, "fp32", 0), (80, "fp32", 0), (64, "fp32", 0), (160, "bf16x3", 1)):
        net = CostRegNet2D(d)
        sd = synth.seeded_state_dict(net, seed=1)
        blob = packing.pack_cost_reg_net_2d(sd, "", prec)
        dr = packing.reg_width(d, prec)
        assert blob.numel() == lib.adamvs_cost_reg_net_2d_weight_floats(dr, code) > 0, (d, prec)
        lw = 9 * dr * dr + dr
        prob_bias = blob[10 * lw + 9 * dr * dr:11 * lw]
        assert torch.equal(prob_bias[:d], sd["prob.bias"]) and bool((prob_bias[d:] == packing.PAD_SCORE).all())
        conv0_bias = blob[9 * dr * dr:lw]
        assert bool((conv0_bias[d:] == 0).all())
    assert lib.adamvs_cost_reg_net_2d_weight_floats(40, 0) == 0      # not a width


def test_gru_prescaled_fields_follow_the_header():
    """include/adamvs_hip.h, "PRE-SCALED FIELDS" of adamvs_fuse_weights: recomputed here from the header's text (element
    addresses of the fragment layouts, the factors, the gates1 row order), not through packing.py's helpers."""
    import math
    L2E = math.log2(math.e)
    m = Infer_AdaMVSNet(16, [16, 8, 4], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=3)
    pre = "DepthNet.1.reg_fuse."
    header_rows = [0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15]
    for q in range(4):
        for e in range(4):
            assert header_rows[4 * q + e] == (2 * q + e if e < 2 else 8 + 2 * q + e - 2)

    # ---- bf16x3: split-bf16 fragments [cout tile][hi|lo][k/32][lane][8], k = tap * cin_total + cin
    blob, off = packing.pack_slice_reg_net(sd, pre, "bf16x3")

    def bx3_elem(field, nt_total, K, tile, lane, kb, j):
        nkb = (K + 31) // 32
        frag = blob[off[field]:off[field] + nt_total * 2 * nkb * 64 * 4].view(torch.bfloat16).reshape(nt_total, 2, nkb, 64, 8)
        return float(frag[tile, 0, kb, lane, j]) + float(frag[tile, 1, kb, lane, j])

    for field, key, nt, s, rows in (("gates1", "conv_gru1.conv_gates.0", 1, -L2E, header_rows), ("cand1", "conv_gru1.convc.0", 1, 2 * L2E, "two-row"),
                                    ("gates2", "conv_gru2.conv_gates.0", 2, -L2E, None), ("cand2", "conv_gru2.convc.0", 1, 2 * L2E, None)):
        w, b = sd[pre + key + ".weight"].double(), sd[pre + key + ".bias"].double()
        cout, cin = w.shape[0], w.shape[1]
        two_row = rows == "two-row"                    # cand1 (ABI 15): positions (rr, kx), rows 0-7 output row y, 8-15 row y + 1
        npos = 12 if two_row else 9
        for tile, lane, kb, j in ((0, 5, 0, 3), (nt - 1, 38, 2, 7), (0, 63, (npos * cin) // 32 - 1, 0), (nt - 1, 16 + 9, 1, 4), (0, 12, 4, 2)):
            row = 16 * tile + (lane & 15)
            k = 32 * kb + 8 * (lane >> 4) + j
            tap, ci = k // cin, k % cin
            if two_row:
                rr, kx = tap // 3, tap % 3
                src, ky = (row, rr) if row < 8 else (row - 8, rr - 1)
                want = s * float(w[src, ci, ky, kx]) if 0 <= ky <= 2 else 0.0
            else:
                src = rows[row] if rows else row
                want = s * float(w[src, ci, tap // 3, tap % 3]) if src < cout and tap < 9 else 0.0
            got = bx3_elem(field, nt, npos * cin, tile, lane, kb, j)
            assert abs(got - want) <= 2 ** -15 * max(abs(want), 1e-6) + 1e-12, (field, tile, lane, kb, j, got, want)
        rows = None if two_row else rows
        bias = blob[off[field + "_b"]:off[field + "_b"] + 16 * nt]
        for row in range(16 * nt):
            src = rows[row] if rows else row
            want = s * float(b[src]) if src < cout else 0.0
            assert abs(float(bias[row]) - want) <= 1e-6 * max(abs(want), 1.0), (field, row)

    # ---- fp32: direct fields unscaled, *_w = G (s g) G^T as [cout tile][i][j][cin/4][64]
    blob, off = packing.pack_slice_reg_net(sd, pre, "fp32")
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    for field, key, s in (("gates1_w", "conv_gru1.conv_gates.0", -L2E), ("gates2_w", "conv_gru2.conv_gates.0", -L2E),
                          ("cand2_w", "conv_gru2.convc.0", 2 * L2E), ("cand1_w", "conv_gru1.convc.0", 2 * L2E)):
        w = sd[pre + key + ".weight"].double()
        cout, cin = w.shape[0], w.shape[1]
        nt, KC = (cout + 15) // 16, cin // 4
        frag = blob[off[field]:off[field] + nt * 16 * KC * 64].reshape(nt, 4, 4, KC, 64)
        for tile, i, j, kc, lane in ((0, 0, 0, 0, 0), (nt - 1, 3, 1, KC - 1, 63), (0, 2, 2, 1, 21), (nt - 1, 1, 3, 0, 40)):
            co, ci = 16 * tile + (lane & 15), 4 * kc + (lane >> 4)
            want = float((G @ (s * w[co, ci]) @ G.t())[i, j]) if co < cout else 0.0
            assert abs(float(frag[tile, i, j, kc, lane]) - want) <= 1e-6 * max(abs(want), 1e-3), (field, tile, i, j, kc, lane)
    # the direct fp32 gate fragments [tile][tap][cin/4][64] are the unscaled reference weights, and so is their bias
    wg = sd[pre + "conv_gru2.conv_gates.0.weight"]
    frag = blob[off["gates2"]:off["gates2"] + 2 * 9 * 8 * 64].reshape(2, 9, 8, 64)
    assert float(frag[1, 4, 3, 37]) == float(wg[16 + (37 & 15), 4 * 3 + (37 >> 4), 1, 1])
    assert torch.equal(blob[off["gates2_b"]:off["gates2_b"] + 32], sd[pre + "conv_gru2.conv_gates.0.bias"].float())


def test_graphed_forward_host_side():
    """ada_mvs_amd/graphed.py without a GPU: the half spans it hands the captured kernels are the reference's Python-float products
    (ndepth / 2 * ratio * depth_interval with depth_interval = (max - min) / num_depth from batch item 0, models/adamvs.py:569-571,
    models/module.py:632) -- from a host tensor, a [B,2] or longer depth_values row -- and CPU images raise instead of running anything."""
    import torch
    from ada_mvs_amd import hip_ops, synth
    from ada_mvs_amd.graphed import GraphedForward
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    m = Infer_AdaMVSNet(48, [48, 32, 8], [4.0, 2.0, 1.0], False, [8, 8, 8]).eval()
    fwd = GraphedForward(m)
    dv = torch.tensor([[425.5, 611.25], [1.0, 2.0]], dtype=torch.float32)            # item 0 only (quirk Q4)
    interval = (float(dv[0, 1]) - float(dv[0, 0])) / 48
    assert fwd._spans(dv) == [48 / 2.0 * (4.0 * interval), 32 / 2.0 * (2.0 * interval), 8 / 2.0 * (1.0 * interval)]
    assert fwd._spans(dv) == [hip_ops.half_span_of(n, r * interval) for n, r in zip([48, 32, 8], [4.0, 2.0, 1.0])]
    imgs, proj, dv1 = synth.tile_inputs("tiny", batch=1, seed=0)
    with pytest.raises(_lib.AdaMVSHipError, match="MI355X"):
        fwd(imgs, proj, dv1)
    assert fwd.captures == 0 and len(fwd.cache) == 0
