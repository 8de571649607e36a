"""Host-side weight repacking for the gfx950 kernels (done once per checkpoint).

Convolution weights become MFMA A-fragment streams for v_mfma_f32_16x16x4_f32
(lane l of a fragment holds W[cout = 16*tile + (l & 15)][cin = 4*kc + (l >> 4)]
of one tap), eval-mode BatchNorm is folded into the CostRegNet2D weights.
Layouts are specified in include/adamvs_hip.h.  Pure tensor shuffling on the
CPU; no arithmetic of the hot path happens here.
"""
import os

import torch

BN_EPS = 1e-5

REG_LAYERS = ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11", "prob")
REG_TRANSPOSED = ("conv7", "conv9", "conv11")
REG_WINOGRAD = ("conv0", "conv2", "conv4", "conv6", "prob")      # stride-1 layers: also packed in the F(2x2, 3x3) form (fp32)
WINO_WIDTHS = (64, 128, 192, 256, 384, 512)              # the widths of REG_WIDTHS that csrc/costreg2d_wino.hip takes (multiples of 64)
# The widths CostRegNet2D's kernels are built for (csrc/costreg2d.hip::costreg_width, costreg_width_bf16x3; held equal to the
# library's table by tests/test_host_logic.py).  The reference builds the network for any number of hypotheses
# (models/adamvs.py:198-228): other D run at the next width, see pack_cost_reg_net_2d.
REG_WIDTHS = {"fp32": (16, 32, 48, 64, 96, 128, 192, 256, 384, 512), "bf16x3": (32, 64, 96, 128, 192, 256, 384, 512)}
PAD_SCORE = -1e30          # bias of `prob`'s pad channels: exp(PAD_SCORE - max) = 0 exactly, and finite (no inf - inf in the merges)


def reg_width(d, precision="fp32"):
    """The width CostRegNet2D runs at for d hypotheses."""
    for w in REG_WIDTHS[precision]:
        if d <= w:
            return w
    raise ValueError("CostRegNet2D: %d hypotheses (at most %d)" % (d, REG_WIDTHS[precision][-1]))


def _as_cout_cin_tap(w, transposed):
    """-> [cout][cin][9] fp32."""
    w = w.detach().to(torch.float32).cpu()
    if transposed:                      # ConvTranspose2d stores [cin][cout][ky][kx]
        w = w.permute(1, 0, 2, 3)
    return w.reshape(w.shape[0], w.shape[1], 9).contiguous()


def split_bf16(x):
    """fp32 -> (hi, lo) bf16 with hi + lo = x to 16 significant bits."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
    return hi, lo


def pack_small_conv(w, transposed=False):
    """[cout][cin][3][3] -> A fragments [NT][9][cin/4][64], cout zero-padded to 16*NT."""
    w = _as_cout_cin_tap(w, transposed)
    cout, cin = w.shape[0], w.shape[1]
    assert cin % 4 == 0
    nt = (cout + 15) // 16
    wp = torch.zeros(nt * 16, cin, 9)
    wp[:cout] = w
    # (nt, co16, kc, k4, tap) -> (nt, tap, kc, k4, co16): lane = k4*16 + co16
    return wp.reshape(nt, 16, cin // 4, 4, 9).permute(0, 4, 2, 3, 1).contiguous().reshape(-1)


def pack_conv1_two_row(w):
    """conv1 [8][cin][3][3] -> [12 = (rr 0..3) x (kx 0..2)][cin/4][64] two-row A fragments (include/adamvs_hip.h):
    MFMA rows 0-7 produce output row y (tap ky = rr), rows 8-15 output row y+1 (tap ky = rr-1)."""
    w = w.detach().to(torch.float32).cpu()
    cout, cin = w.shape[0], w.shape[1]
    assert cout == 8 and cin % 4 == 0
    wp = torch.zeros(4, 3, 16, cin)                       # [rr][kx][row16][cin]
    for rr in range(4):
        if rr <= 2:
            wp[rr, :, :8] = w[:, :, rr, :].permute(2, 0, 1)       # [kx][co][cin]
        if rr >= 1:
            wp[rr, :, 8:] = w[:, :, rr - 1, :].permute(2, 0, 1)
    # (rr, kx, row16, kc, k4) -> (rr, kx, kc, k4, row16): lane = k4*16 + row16
    return wp.reshape(4, 3, 16, cin // 4, 4).permute(0, 1, 3, 4, 2).contiguous().reshape(-1)


def _pack_rows_bf16x3(wm):
    """[16*NT rows][K] fp32 matrix -> split-bf16 A fragments [NT][hi|lo][K/32][64][8] (K zero-padded to 32),
    returned as a float32 view of the bf16 stream (include/adamvs_hip.h)."""
    rows, k = wm.shape
    nt, nkb = rows // 16, (k + 31) // 32
    wp = torch.zeros(rows, nkb * 32)
    wp[:, :k] = wm
    parts = []
    for part in split_bf16(wp):
        # (nt, co16, kb, kg4, j8) -> (nt, kb, kg4, co16, j8): lane = kg4*16 + co16
        parts.append(part.reshape(nt, 16, nkb, 4, 8).permute(0, 2, 3, 1, 4).contiguous())
    frag = torch.stack(parts, 1).contiguous().reshape(-1)           # [nt][hi|lo][kb][lane][8]
    if frag.numel() % 2:
        frag = torch.cat([frag, torch.zeros(1, dtype=torch.bfloat16)])
    return frag.view(torch.float32)


def pack_small_conv_bf16x3(w):
    """[cout][cin][3][3] -> split-bf16 fragments with k = (ky*3+kx)*cin_total + cin, cout zero-padded to 16*NT."""
    w = _as_cout_cin_tap(w, False)                      # [cout][cin][9]
    cout, cin = w.shape[0], w.shape[1]
    assert cin % 8 == 0
    nt = (cout + 15) // 16
    wm = torch.zeros(nt * 16, 9 * cin)
    wm[:cout] = w.permute(0, 2, 1).reshape(cout, 9 * cin)           # [cout][tap][cin]
    return _pack_rows_bf16x3(wm)


def pack_conv1_two_row_bf16x3(w):
    """conv1 [8][cin][3][3] in two-row form (rows 0-7: output row y, ky = rr; rows 8-15: row y+1, ky = rr-1),
    k = (rr*3+kx)*cin + cin_idx, rr = 0..3."""
    w = w.detach().to(torch.float32).cpu()
    cout, cin = w.shape[0], w.shape[1]
    assert cout == 8 and cin % 8 == 0
    wm = torch.zeros(16, 4, 3, cin)                                 # [row][rr][kx][cin]
    for rr in range(4):
        if rr <= 2:
            wm[:8, rr] = w[:, :, rr, :].permute(0, 2, 1)            # [co][kx][cin]
        if rr >= 1:
            wm[8:, rr] = w[:, :, rr - 1, :].permute(0, 2, 1)
    return _pack_rows_bf16x3(wm.reshape(16, 12 * cin))


def pad_bias(b, n):
    out = torch.zeros(n)
    out[:b.numel()] = b.detach().to(torch.float32).cpu().reshape(-1)
    return out


def pack_reg_layer(w, scale, shift, transposed):
    """One CostRegNet2D layer -> [9][D/4][D/16][64] fragment floats + [D] bias floats."""
    w = _as_cout_cin_tap(w, transposed) * scale.reshape(-1, 1, 1)
    d = w.shape[0]
    assert w.shape[1] == d and d % 16 == 0, "CostRegNet2D width must be a multiple of 16"
    # (tile, co16, kc, k4, tap) -> (tap, kc, tile, k4, co16)
    frag = w.reshape(d // 16, 16, d // 4, 4, 9).permute(4, 2, 0, 3, 1).contiguous().reshape(-1)
    return torch.cat([frag, shift.detach().to(torch.float32).cpu().reshape(-1)])


WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)


def pack_reg_layer_wino(w, scale):
    """A stride-1 CostRegNet2D layer for the F(2x2, 3x3) kernel: U = G w G^T (formed in double precision, rounded once),
    as A fragments [D/4][4 = patch row i][D/16][64][4 = patch column j] (include/adamvs_hip.h)."""
    w = (w.detach().to(torch.float64).cpu() * scale.detach().to(torch.float64).cpu().reshape(-1, 1, 1, 1))   # [co][ci][3][3]
    d = w.shape[0]
    assert w.shape[1] == d and d % 16 == 0
    u = torch.einsum("ik,ockl,jl->ijoc", WINO_G, w, WINO_G).to(torch.float32)                               # [i][j][co][ci]
    # (i, j, tile, co16, kc, k4) -> (kc, i, tile, k4, co16, j): lane = k4*16 + co16
    return u.reshape(4, 4, d // 16, 16, d // 4, 4).permute(4, 0, 2, 5, 3, 1).contiguous().reshape(-1)


def pack_reg_layer_bf16x3(w, scale, shift, transposed):
    """One CostRegNet2D layer for the split-bf16 kernels: 9*D*D floats worth of bf16 fragments
    [hi|lo][tap][cin/32][cout/16][lane][8] (include/adamvs_hip.h) + [D] fp32 bias; same size as the fp32 packing."""
    w = _as_cout_cin_tap(w, transposed) * scale.reshape(-1, 1, 1)
    d = w.shape[0]
    assert w.shape[1] == d and d % 32 == 0, "bf16x3 CostRegNet2D width must be a multiple of 32"
    halves = []
    for part in split_bf16(w):
        # (tile, co16, kb, kg4, j8, tap) -> (tap, kb, tile, kg4, co16, j8): lane = kg4*16 + co16, 8 bf16 per lane
        halves.append(part.reshape(d // 16, 16, d // 32, 4, 8, 9).permute(5, 2, 0, 3, 1, 4).contiguous().reshape(-1))
    frag = torch.cat(halves).view(torch.float32)          # reinterpret the bf16 stream as floats: 9*d*d of them
    assert frag.numel() == 9 * d * d
    return torch.cat([frag, shift.detach().to(torch.float32).cpu().reshape(-1)])


def _pad_square(w, transposed, dr):
    """[D][D][3][3] conv weights (ConvTranspose2d: [in][out]) zero-padded to [dr][dr][3][3]."""
    d = w.shape[0]
    if d == dr:
        return w
    out = torch.zeros(dr, dr, 3, 3, dtype=w.dtype)
    out[:d, :d] = w.detach().cpu()
    return out


def pack_cost_reg_net_2d(sd, pre, precision="fp32"):
    """All 11 layers of `pre` (e.g. 'DepthNet.0.reg.') from a reference-keyed state dict; fp32 at the widths the F(2x2, 3x3)
    kernel supports: followed by the five stride-1 layers in that form (16*D*D floats each, include/adamvs_hip.h).
    A network of D hypotheses is packed at Dr = reg_width(D) channels: zero filters for the pad channels (their activations are
    ReLU(0 + 0) = 0 in every layer and no real channel reads them), PAD_SCORE as the bias of `prob`'s pad channels."""
    pack_layer = pack_reg_layer if precision == "fp32" else pack_reg_layer_bf16x3
    d = sd[pre + "prob.weight"].shape[0]
    dr = reg_width(d, precision)

    def pad_vec(v, fill):
        out = torch.full((dr,), float(fill))
        out[:d] = v.detach().float().cpu()
        return out

    chunks, wino = [], []
    for name in REG_LAYERS:
        if name == "prob":
            w = sd[pre + "prob.weight"]
            scale = torch.ones(w.shape[0])
            shift = sd[pre + "prob.bias"]
            transposed = False
        else:
            transposed = name in REG_TRANSPOSED
            wkey, bnpre = (pre + name + ".0.weight", pre + name + ".1.") if transposed else \
                          (pre + name + ".conv.weight", pre + name + ".bn.")
            w = sd[wkey]
            g = sd[bnpre + "weight"].detach().float().cpu()
            b = sd[bnpre + "bias"].detach().float().cpu()
            mu = sd[bnpre + "running_mean"].detach().float().cpu()
            var = sd[bnpre + "running_var"].detach().float().cpu()
            scale = g / torch.sqrt(var + BN_EPS)
            shift = b - mu * scale
        w = _pad_square(w, transposed, dr)
        scale, shift = pad_vec(scale, 1.0), pad_vec(shift, PAD_SCORE if (name == "prob" and dr > d) else 0.0)
        chunks.append(pack_layer(w, scale, shift, transposed))
        if precision == "fp32" and name in REG_WINOGRAD and dr in WINO_WIDTHS:
            wino.append(pack_reg_layer_wino(w, scale))
    return torch.cat(chunks + wino)


FUSE_FIELDS = ("conv1", "gates1", "gates1_b", "cand1", "cand1_b", "conv2", "gates2", "gates2_b",
               "cand2", "cand2_b", "upconv1", "upconv1_b", "final_w", "gates1_w", "gates2_w", "cand2_w", "cand1_w")


def pack_small_conv_wino(w, scale=1.0):
    """[cout][cin][3][3] -> the transformed filters U = G (scale w) G^T (double precision, rounded once) of the F(2x2, 3x3) form
    as A fragments [NT][4 = patch row i][4 = patch column j][cin/4][64], cout zero-padded to 16*NT (csrc/slice_roles_wino.h).
    scale: -log2(e) for the gate convolutions, 2 log2(e) for the candidates -- the kernels feed the result to v_exp_f32."""
    w = w.detach().to(torch.float64).cpu() * scale
    cout, cin = w.shape[0], w.shape[1]
    assert cin % 4 == 0
    nt = (cout + 15) // 16
    u = torch.zeros(4, 4, nt * 16, cin, dtype=torch.float32)
    u[:, :, :cout] = torch.einsum("ik,ockl,jl->ijoc", WINO_G, w, WINO_G).to(torch.float32)
    # (i, j, nt, co16, kc, k4) -> (nt, i, j, kc, k4, co16): lane = k4*16 + co16
    return u.reshape(4, 4, nt, 16, cin // 4, 4).permute(2, 0, 1, 4, 5, 3).contiguous().reshape(-1)


def pack_slice_reg_net(sd, pre, precision="fp32"):
    """SliceCostRegNetRED of `pre` (e.g. 'DepthNet.0.reg_fuse.') -> (flat fp32 tensor, {field: offset}).
    precision "bf16x3": conv1 / gates / cand / conv2 as split-bf16 fragments (upconv1, final layer, biases stay fp32)."""
    pack_small, pack_c1 = (pack_small_conv, pack_conv1_two_row) if precision == "fp32" else \
                          (pack_small_conv_bf16x3, pack_conv1_two_row_bf16x3)
    wg1, bg1 = sd[pre + "conv_gru1.conv_gates.0.weight"], sd[pre + "conv_gru1.conv_gates.0.bias"]
    LOG2E = 1.4426950408889634
    # bf16x3: gate convolutions scaled by -log2(e), candidate convolutions by 2 log2(e): the fused GRU kernels feed their
    # accumulators to v_exp_f32 (2^x) directly (csrc/slice_roles_bx3.h: sigmoid_pre, tanh_pre)
    gs, cs = (1.0, 1.0) if precision == "fp32" else (-LOG2E, 2.0 * LOG2E)
    scaled = lambda key, s: sd[pre + key].detach().cpu().double().mul(s).float()       # noqa: E731
    if precision != "fp32":
        wg1, bg1 = scaled("conv_gru1.conv_gates.0.weight", gs), scaled("conv_gru1.conv_gates.0.bias", gs)
        # the fused level-1 kernel of the split-bf16 mode (csrc/slice_roles_bx3.h, Gru1FusedBx3Role) wants the gate rows
        # interleaved: MFMA row 4q + e = reset-gate channel 2q + e (e = 0, 1) | update-gate channel 2q + e - 2 (e = 2, 3)
        perm = torch.tensor([2 * (m // 4) + (m % 4) if m % 4 < 2 else 8 + 2 * (m // 4) + (m % 4 - 2) for m in range(16)])
        wg1, bg1 = wg1.detach().cpu()[perm], bg1.detach().cpu()[perm]
    parts = {
        "conv1": pack_c1(sd[pre + "conv1.conv.weight"]),
        "gates1": pack_small(wg1),
        "gates1_b": pad_bias(bg1, 16),
        # 8 outputs fill half an MFMA tile, so the kernels take them in the two-row form of conv1 (bf16x3: since ABI 15;
        # ADAMVS_BX3_CAND_TWO_ROW=0 packs the one-row form for a library built with -DBX3_CAND_TWO_ROW=0: A/B runs only)
        "cand1": (pack_c1 if precision == "fp32" or os.environ.get("ADAMVS_BX3_CAND_TWO_ROW", "1") != "0" else pack_small)(
            scaled("conv_gru1.convc.0.weight", cs)),
        "cand1_b": pad_bias(scaled("conv_gru1.convc.0.bias", cs), 16),
        "conv2": pack_small(sd[pre + "conv2.conv.weight"]),
        "gates2": pack_small(scaled("conv_gru2.conv_gates.0.weight", gs)),
        "gates2_b": pad_bias(scaled("conv_gru2.conv_gates.0.bias", gs), 32),
        "cand2": pack_small(scaled("conv_gru2.convc.0.weight", cs)),
        "cand2_b": pad_bias(scaled("conv_gru2.convc.0.bias", cs), 16),
        "upconv1": pack_small_conv(sd[pre + "upconv1.weight"], transposed=True),
        "upconv1_b": pad_bias(sd[pre + "upconv1.bias"], 16),
        # [8][1][3][3] (transposed, stages 1-2) and [1][8][3][3] (stage 3) both flatten to c*9 + tap; stored
        # tap-major, w[tap*8 + c], so that a channel pair of one tap is one 64-bit scalar operand
        "final_w": pad_bias(torch.cat([sd[pre + "upconv2d.weight"].detach().float().cpu().reshape(8, 9).t().reshape(-1),
                                       sd[pre + "upconv2d.bias"].detach().float().cpu().reshape(-1)]), 76),
    }
    if precision == "fp32":
        parts["gates1_w"] = pack_small_conv_wino(sd[pre + "conv_gru1.conv_gates.0.weight"], -LOG2E)
        parts["gates2_w"] = pack_small_conv_wino(sd[pre + "conv_gru2.conv_gates.0.weight"], -LOG2E)
        parts["cand2_w"] = pack_small_conv_wino(sd[pre + "conv_gru2.convc.0.weight"], 2.0 * LOG2E)
        parts["cand1_w"] = pack_small_conv_wino(sd[pre + "conv_gru1.convc.0.weight"], 2.0 * LOG2E)
    offsets, chunks, o = {}, [], 0
    for f in FUSE_FIELDS:
        if f not in parts:               # the F(2x2, 3x3) blocks exist in fp32 only: the field stays NULL
            continue
        t = parts[f]
        pad = (-t.numel()) % 64          # keep every field 256-byte aligned
        offsets[f] = o
        chunks.append(torch.cat([t, torch.zeros(pad)]))
        o += t.numel() + pad
    return torch.cat(chunks), offsets


# --------------------------------------------------------------------------------------------------------------
# FeatureNet0 (include/adamvs_hip.h: adamvs_feature_weights)
FEATURE_CONVS = ("conv0_0", "conv0_1", "conv1_0", "conv1_1", "conv1_2", "conv2_0", "conv2_1", "conv2_2",
                 "out1", "deconv1_t", "deconv1_c", "out2", "deconv2_t", "deconv2_c", "out3")
FEATURE_BRANCHES = ("br1_1", "br1_2", "br2_1", "br2_2", "br3_1", "br3_2")


def _bn_scale_shift(sd, pre):
    g, b = sd[pre + "weight"].detach().float().cpu(), sd[pre + "bias"].detach().float().cpu()
    m, v = sd[pre + "running_mean"].detach().float().cpu(), sd[pre + "running_var"].detach().float().cpu()
    scale = g / torch.sqrt(v + BN_EPS)
    return scale, b - m * scale


def pack_taps(w):
    """[cout][cin][ntaps] -> A fragments [NT][ntaps][cin/4][64] (cout zero-padded to 16*NT, cin to a multiple of 4)."""
    cout, cin, nt_ = w.shape
    cin4 = (cin + 3) // 4 * 4
    nt = (cout + 15) // 16
    wp = torch.zeros(nt * 16, cin4, nt_)
    wp[:cout, :cin] = w
    # (nt, co16, kc, k4, tap) -> (nt, tap, kc, k4, co16): lane = k4*16 + co16
    return wp.reshape(nt, 16, cin4 // 4, 4, nt_).permute(0, 4, 2, 3, 1).contiguous().reshape(-1)


def conv0_two_row(sd, pre, parts):
    """conv0.0 (3 -> 8; the input padded to RGB + a zero channel) and conv0.1 (8 -> 8) of both feature nets as two-row
    fragments [12][cin/4][64] (pack_conv1_two_row), BatchNorm scale folded in, shift -> bias: the layout of k_conv0_fused."""
    for name, key in (("conv0_0", "conv0.0."), ("conv0_1", "conv0.1.")):
        w = sd[pre + key + "conv.weight"].detach().float().cpu()
        scale, shift = _bn_scale_shift(sd, pre + key + "bn.")
        w = w * scale.reshape(-1, 1, 1, 1)
        if w.shape[1] % 4:
            w = torch.cat([w, torch.zeros(w.shape[0], 4 - w.shape[1] % 4, 3, 3)], 1)
        parts[name + ".w"] = pack_conv1_two_row(w)
        parts[name + ".b"] = pad_bias(shift, 16)


def pack_feature_net(sd, pre="feature.", context=True):
    """FeatureNet0 of `pre` -> (flat fp32 tensor, {field: offset}); fields 'name.w' / 'name.b' for the convolutions,
    'name.w1' / 'name.b1' / 'name.w2' for the pooled-context branches.  context=False: the plain U-Net `FeatureNet`
    of MS-REDNet (reference models/msrednet.py:29-127) -- same layers, out_k are [C][C] and there are no branches, so
    the branch weights are zeros (their contribution to out_k vanishes)."""
    parts = {}

    def conv_bn(name, key, taps):
        w = sd[pre + key + "conv.weight"].detach().float().cpu()
        scale, shift = _bn_scale_shift(sd, pre + key + "bn.")
        w = (w * scale.reshape(-1, 1, 1, 1)).reshape(w.shape[0], w.shape[1], taps)
        parts[name + ".w"] = pack_taps(w)
        parts[name + ".b"] = pad_bias(shift, (w.shape[0] + 15) // 16 * 16)

    conv0_two_row(sd, pre, parts)
    conv_bn("conv1_0", "conv1.0.", 25)
    conv_bn("conv1_1", "conv1.1.", 9)
    conv_bn("conv1_2", "conv1.2.", 9)
    conv_bn("conv2_0", "conv2.0.", 25)
    conv_bn("conv2_1", "conv2.1.", 9)
    conv_bn("conv2_2", "conv2.2.", 9)
    conv_bn("deconv1_c", "deconv1.conv.", 9)
    # deconv2.conv has 8 output channels: two-row fragments [12][4][64] (k_fconv_pair_two_row), like conv0
    w = sd[pre + "deconv2.conv.conv.weight"].detach().float().cpu()
    scale, shift = _bn_scale_shift(sd, pre + "deconv2.conv.bn.")
    parts["deconv2_c.w"] = pack_conv1_two_row(w * scale.reshape(-1, 1, 1, 1))
    parts["deconv2_c.b"] = pad_bias(shift, 16)
    for name, key in (("deconv1_t", "deconv1.deconv."), ("deconv2_t", "deconv2.deconv.")):
        w = sd[pre + key + "conv.weight"].detach().float().cpu().permute(1, 0, 2, 3)        # -> [cout][cin][ky][kx]
        scale, shift = _bn_scale_shift(sd, pre + key + "bn.")
        w = w * scale.reshape(-1, 1, 1, 1)
        classes = []
        for py in (0, 1):
            for px in (0, 1):
                taps = [w[:, :, (0 if ty else 2) if py else 1, (0 if tx else 2) if px else 1]
                        for ty in range(1 + py) for tx in range(1 + px)]
                classes.append(pack_taps(torch.stack(taps, 2)))
        parts[name + ".w"] = torch.cat(classes)
        parts[name + ".b"] = pad_bias(shift, 16)
    for k, C in ((1, 32), (2, 16), (3, 8)):
        if not context:
            wo = sd[pre + "out%d.weight" % k].detach().float().cpu().reshape(C, C)
            parts["out%d.w" % k] = pack_taps(wo.reshape(C, C, 1))
            parts["out%d.b" % k] = torch.zeros((C + 15) // 16 * 16)
            for j in (1, 2):
                parts["br%d_%d.w1" % (k, j)] = torch.zeros(C // 2 * C)
                parts["br%d_%d.b1" % (k, j)] = torch.zeros(C // 2)
                parts["br%d_%d.w2" % (k, j)] = torch.zeros(C * (C // 2))
            continue
        wo = sd[pre + "out%d.weight" % k].detach().float().cpu().reshape(C, 2 * C)
        parts["out%d.w" % k] = pack_taps(wo[:, C:].reshape(C, C, 1))
        parts["out%d.b" % k] = torch.zeros((C + 15) // 16 * 16)
        for j in (1, 2):
            key = "branch%d_%d.1." % (k, j)
            w1 = sd[pre + key + "conv.weight"].detach().float().cpu().reshape(C // 2, C)
            scale, shift = _bn_scale_shift(sd, pre + key + "bn.")
            parts["br%d_%d.w1" % (k, j)] = (w1 * scale.reshape(-1, 1)).reshape(-1)
            parts["br%d_%d.b1" % (k, j)] = shift.clone()
            parts["br%d_%d.w2" % (k, j)] = wo[:, (j - 1) * (C // 2):j * (C // 2)].contiguous().reshape(-1)
    offsets, chunks, o = {}, [], 0
    for f, t in parts.items():
        pad = (-t.numel()) % 64
        offsets[f] = o
        chunks.append(torch.cat([t.reshape(-1), torch.zeros(pad)]))
        o += t.numel() + pad
    return torch.cat(chunks), offsets


FEATURE_FPN_CONVS = ("conv0_0", "conv0_1", "conv1_0", "conv1_1", "conv1_2", "conv2_0", "conv2_1", "conv2_2",
                     "out1", "inner1", "out2", "inner2", "out3")


def pack_feature_net_fpn(sd, pre="feature."):
    """The FPN variant of MS-REDNet's FeatureNet (reference models/msrednet.py:74-91) -> (flat fp32 tensor, {field: offset})
    for adamvs_feature_fpn_weights: the encoder as in pack_feature_net; out1 / inner1 / inner2 1x1 (the inner ones with
    their bias), out2 / out3 3x3 without bias or BatchNorm."""
    parts = {}

    def conv_bn(name, key, taps):
        w = sd[pre + key + "conv.weight"].detach().float().cpu()
        scale, shift = _bn_scale_shift(sd, pre + key + "bn.")
        w = (w * scale.reshape(-1, 1, 1, 1)).reshape(w.shape[0], w.shape[1], taps)
        parts[name + ".w"] = pack_taps(w)
        parts[name + ".b"] = pad_bias(shift, (w.shape[0] + 15) // 16 * 16)

    conv0_two_row(sd, pre, parts)
    for name, key, taps in (("conv1_0", "conv1.0.", 25), ("conv1_1", "conv1.1.", 9),
                            ("conv1_2", "conv1.2.", 9), ("conv2_0", "conv2.0.", 25), ("conv2_1", "conv2.1.", 9), ("conv2_2", "conv2.2.", 9)):
        conv_bn(name, key, taps)
    for name in ("out1", "inner1", "out2", "inner2", "out3"):
        w = sd[pre + name + ".weight"].detach().float().cpu()
        cout = w.shape[0]
        parts[name + ".w"] = pack_taps(w.reshape(cout, w.shape[1], w.shape[2] * w.shape[3]))
        b = sd.get(pre + name + ".bias")
        b = torch.zeros(cout) if b is None else b.detach().float().cpu()
        parts[name + ".b"] = pad_bias(b, (cout + 15) // 16 * 16)
    offsets, chunks, o = {}, [], 0
    for f, t in parts.items():
        pad = (-t.numel()) % 64
        offsets[f] = o
        chunks.append(torch.cat([t.reshape(-1), torch.zeros(pad)]))
        o += t.numel() + pad
    return torch.cat(chunks), offsets


# ---- MS-REDNet regulariser (reference models/msrednet.py:330-366) on adamvs_conv3x3_dd -------------------------------
def pad16(n):
    return (n + 15) // 16 * 16


def pack_padded_dd(w, bias, D, transposed=False, flip=False, cin_at=0):
    """A 3x3 convolution [cout][cin][3][3] (ConvTranspose2d: [cin][cout][3][3]) zero-padded to D x D for
    adamvs_conv3x3_dd: 9*D*D fragment floats + D bias floats.  The real input channels sit at [cin_at, cin_at + cin)
    of the D-wide map, the real outputs at [0, cout).  flip: a stride-1 ConvTranspose2d (k3 p1) as the equivalent
    convolution -- taps mirrored, channels swapped (reference msrednet.py:345, 364)."""
    w = w.detach().to(torch.float32).cpu()
    if flip:
        w = w.permute(1, 0, 2, 3).flip(2, 3)
    elif transposed:
        w = w.permute(1, 0, 2, 3)              # -> [cout][cin][ky][kx]; the kernel's transposed mode takes it from here
    cout, cin = w.shape[:2]
    assert cout <= D and cin_at + cin <= D and D % 16 == 0
    full = torch.zeros(D, D, 3, 3)
    full[:cout, cin_at:cin_at + cin] = w
    shift = torch.zeros(D)
    if bias is not None:
        shift[:cout] = bias.detach().to(torch.float32).cpu()
    if transposed and not flip:                # pack_reg_layer permutes [cin][cout] -> [cout][cin] itself
        full = full.permute(1, 0, 2, 3).contiguous()
    return pack_reg_layer(full, torch.ones(D), shift, transposed and not flip)


def pack_red_regularization(sd, pre, C):
    """slice_RED_Regularization of `pre` for stage feature width C -> (flat fp32 tensor, {name: (offset, D)}).

    Level k = 1..4: x widths (C, 16, 32, 64) held in maps of width XW = (max(pad16(C), 16), 32, 64, 64), states
    (8, 16, 32, 64) in maps of width HW = (16, 16, 32, 64).  gate_conv / output_conv act on cat(x, h) and are split by
    linearity into an x half (+ bias; applied to all planes at once) and an h half (applied per plane):
    gxr / gxu / cx (width XW) and ghr / ghu / ch (width HW), r = reset rows, u = update rows of gate_conv.
    conv1-3 at the width of their input map; decoder: upconv3 64, upconv2 32, upconv1 16, upconv2d 16.
    gp / cp (levels 1, 2): gate_conv / output_conv whole, as register-resident A fragments for adamvs_conv3x3_pair,
    followed by the bias padded to 16 * tiles (the dict's second entry is that padded output count)."""
    xc, hc = (C, 16, 32, 64), (8, 16, 32, 64)
    XW, HW = (max(pad16(C), 16), 32, 64, 64), (16, 16, 32, 64)
    GW = (XW[0], XW[1], 32, 64)               # width of the x halves' input: level 3 reads conv2's own 32-wide output
    parts = {}
    for k in range(4):
        g = pre + "conv_gru%d." % (k + 1)
        wg, bg = sd[g + "gate_conv.weight"], sd[g + "gate_conv.bias"]
        wc, bc = sd[g + "output_conv.weight"], sd[g + "output_conv.bias"]
        cx, h = xc[k], hc[k]
        n = str(k + 1)
        parts["gxr" + n] = (pack_padded_dd(wg[:h, :cx], bg[:h], GW[k]), GW[k])
        parts["gxu" + n] = (pack_padded_dd(wg[h:, :cx], bg[h:], GW[k]), GW[k])
        parts["cx" + n] = (pack_padded_dd(wc[:, :cx], bc, GW[k]), GW[k])
        parts["ghr" + n] = (pack_padded_dd(wg[:h, cx:], None, HW[k]), HW[k])
        parts["ghu" + n] = (pack_padded_dd(wg[h:, cx:], None, HW[k]), HW[k])
        parts["ch" + n] = (pack_padded_dd(wc[:, cx:], None, HW[k]), HW[k])
        gn = torch.cat([sd[g + m].detach().float().cpu().reshape(-1) for m in (
            "reset_gate_norm.weight", "reset_gate_norm.bias", "update_gate_norm.weight", "update_gate_norm.bias",
            "output_norm.weight", "output_norm.bias")])
        parts["gn" + n] = (gn, h)
    for k in range(2):                          # the two shallow levels: both convolutions of the cell on cat(x, h) directly
        g = pre + "conv_gru%d." % (k + 1)
        for name, key in (("gp", "gate_conv"), ("cp", "output_conv")):
            wt = sd[g + key + ".weight"]
            nt16 = pad16(wt.shape[0])
            parts["%s%d" % (name, k + 1)] = (torch.cat([pack_small_conv(wt), pad_bias(sd[g + key + ".bias"], nt16)]), nt16)
    for k in range(3):                          # conv_{k+1}: level k+1 -> level k+2, stride 2
        parts["conv%d" % (k + 1)] = (pack_padded_dd(sd[pre + "conv%d.conv.weight" % (k + 1)], None, XW[k]), XW[k])
    for name, D in (("upconv3", 64), ("upconv2", 32), ("upconv1", 16)):
        parts[name] = (pack_padded_dd(sd[pre + name + ".conv.weight"], None, D, transposed=True), D)
    parts["upconv2d"] = (pack_padded_dd(sd[pre + "upconv2d.weight"], sd[pre + "upconv2d.bias"], 16, flip=True), 16)
    offsets, chunks, o = {}, [], 0
    for name, (t, D) in parts.items():
        pad = (-t.numel()) % 64
        offsets[name] = (o, D)
        chunks.append(torch.cat([t.reshape(-1), torch.zeros(pad)]))
        o += t.numel() + pad
    return torch.cat(chunks), offsets
