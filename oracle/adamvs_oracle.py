"""CPU oracle for the Ada-MVS depth-inference hot path.  TEST INFRASTRUCTURE.

This file is a plain, unfused CPU restatement (PyTorch CPU ops + explicit index
arithmetic, fp32) of the reference algorithm behind
`models/adamvs.py::Infer_AdaMVSNet.forward`.
It is the checker for the HIP path and the "port" CPU baseline of bench.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it; the product path (ada-mvs_amd/) never does.

Parity status: the reference has no tests or golden vectors of its own
(SURVEY.md section 4).  This oracle is PINNED against outputs of the real
reference, generated in the build container by tools/gen_golden.py (which
imports /root/reference) and committed under tests/golden/;
tests/test_oracle_golden.py checks every function below against them.

All `file:line` citations are relative to the reference repository root.
Weights arrive as a flat state dict using the reference's key names.
"""

import torch
import torch.nn.functional as F

__all__ = [
    "relative_transform", "warp_plane", "warp_plane_grid_sample", "use_grid_sample", "depth_range_samples", "upsample2x",
    "pair_similarity_volume", "cost_reg_net_2d", "softmax_max_regress",
    "aggregate_similarity", "conv_gru_cell", "slice_reg_step", "infer_depth_stage",
    "feature_net", "infer_adamvs_forward",
]


# ----------------------------------------------------------------------------
# a3  homo_warping_float, models/module.py:527-568
# ----------------------------------------------------------------------------
def relative_transform(src_proj, ref_proj):
    """T = P_src . P_ref^-1 -> (R [B,3,3], t [B,3]).  module.py:539-541."""
    proj = torch.matmul(src_proj, torch.inverse(ref_proj))
    return proj[:, :3, :3].contiguous(), proj[:, :3, 3].contiguous()


def warp_plane(src_fea, R, t, depth):
    """Warp src_fea [B,C,h,w] onto one hypothesis plane depth [B,h,w].

    X = R.[x,y,1]^T . d + t; (u,v) = (X0/X2, X1/X2)  (module.py:549-553);
    the reference then normalises by (w-1)/2,(h-1)/2 (554-555) and calls
    grid_sample(bilinear, zeros, align_corners=True) (563-564), which maps the
    normalised coordinate straight back to pixels: the fp32 round trip is kept
    here, the bilinear gather is written out (4 taps, out-of-range taps add 0).
    """
    B, C, h, w = src_fea.shape
    dev = src_fea.device
    dt = src_fea.dtype        # fp32 as the reference; float64 when a test asks for the exact-arithmetic version of a path
    y, x = torch.meshgrid(torch.arange(h, dtype=dt, device=dev), torch.arange(w, dtype=dt, device=dev), indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(h * w, dtype=dt, device=dev)))  # [3,hw]
    rot_xyz = torch.matmul(R.to(dt), xyz.unsqueeze(0).expand(B, 3, h * w))           # [B,3,hw]
    p = rot_xyz * depth.reshape(B, 1, h * w).to(dt) + t.to(dt).reshape(B, 3, 1)
    u = p[:, 0] / p[:, 2]
    v = p[:, 1] / p[:, 2]
    gx = u / ((w - 1) / 2) - 1
    gy = v / ((h - 1) / 2) - 1
    ix = ((gx + 1) / 2) * (w - 1)
    iy = ((gy + 1) / 2) * (h - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    wnw = (x1 - ix) * (y1 - iy)
    wne = (ix - x0) * (y1 - iy)
    wsw = (x1 - ix) * (iy - y0)
    wse = (ix - x0) * (iy - y0)
    flat = src_fea.reshape(B, C, h * w)
    out = torch.zeros(B, C, h * w, dtype=src_fea.dtype, device=dev)
    for xx, yy, ww in ((x0, y0, wnw), (x1, y0, wne), (x0, y1, wsw), (x1, y1, wse)):
        ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
        idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long()
        idx = torch.where(ok, idx, torch.zeros_like(idx))
        g = torch.gather(flat, 2, idx.unsqueeze(1).expand(B, C, h * w))
        out += g * (ww * ok.to(ww.dtype)).unsqueeze(1)
    return out.reshape(B, C, h, w)


def warp_plane_grid_sample(src_fea, R, t, depth):
    """Same warp through F.grid_sample, the way the reference issues it (module.py:554-564).  Used by the timed
    CPU baseline (ATen's vectorised sampler instead of the explicit gather above); tests hold the two equal."""
    B, C, h, w = src_fea.shape
    dev = src_fea.device
    dt = src_fea.dtype
    y, x = torch.meshgrid(torch.arange(h, dtype=dt, device=dev), torch.arange(w, dtype=dt, device=dev), indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(h * w, dtype=dt, device=dev)))
    p = torch.matmul(R.to(dt), xyz.unsqueeze(0).expand(B, 3, h * w)) * depth.reshape(B, 1, h * w).to(dt) + t.to(dt).reshape(B, 3, 1)
    gx = (p[:, 0] / p[:, 2]) / ((w - 1) / 2) - 1
    gy = (p[:, 1] / p[:, 2]) / ((h - 1) / 2) - 1
    grid = torch.stack((gx, gy), dim=2).view(B, h, w, 2)
    return F.grid_sample(src_fea, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


_WARP = [warp_plane]          # the sampler the composite functions below use (swapped by use_grid_sample)


class use_grid_sample:
    """Context manager: run the composite oracle functions with the grid_sample form of the warp."""

    def __enter__(self):
        _WARP[0] = warp_plane_grid_sample

    def __exit__(self, *a):
        _WARP[0] = warp_plane


# ----------------------------------------------------------------------------
# a2  get_depth_range_samples, models/module.py:646-663 (+ 628-643)
# ----------------------------------------------------------------------------
def depth_range_samples(cur_depth, ndepth, depth_interval_pixel, shape):
    """Hypothesis planes [B,D,h,w].

    cur_depth [B,2]=[min,max]: D uniform samples over [min,max]; the interval
    argument is ignored (module.py:650-658).  cur_depth [B,h,w]: window
    cur -+ D/2.interval, D samples, NOT clamped (module.py:632-641).
    """
    B, h, w = shape
    if cur_depth.dim() == 2:
        dmin = cur_depth[:, 0]
        dmax = cur_depth[:, -1]
        step = (dmax - dmin) / (ndepth - 1)
        k = torch.arange(0, ndepth, dtype=cur_depth.dtype).reshape(1, -1)
        s = dmin.unsqueeze(1) + k * step.unsqueeze(1)
        return s.reshape(B, ndepth, 1, 1).repeat(1, 1, h, w)
    assert tuple(cur_depth.shape) == (B, h, w)
    lo = cur_depth - ndepth / 2 * depth_interval_pixel
    hi = cur_depth + ndepth / 2 * depth_interval_pixel
    step = (hi - lo) / (ndepth - 1)
    k = torch.arange(0, ndepth, dtype=cur_depth.dtype).reshape(1, -1, 1, 1)
    return lo.unsqueeze(1) + k * step.unsqueeze(1)


# ----------------------------------------------------------------------------
# a7  F.interpolate(..., bilinear, align_corners=False) at exactly 2x
# ----------------------------------------------------------------------------
def upsample2x(x):
    """2x bilinear upsample of [B,C,h,w], half-pixel centres, edge clamp:
    src = max((dst+0.5)/2-0.5, 0), hi index clamped (adamvs.py:505, 522)."""
    B, C, h, w = x.shape

    def taps(n_out, n_in):
        d = torch.arange(n_out, dtype=torch.float32)
        s = ((d + 0.5) * 0.5 - 0.5).clamp(min=0)
        i0 = s.floor().long()
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        l1 = s - i0.to(torch.float32)
        return i0, i1, 1.0 - l1, l1

    y0, y1, wy0, wy1 = taps(2 * h, h)
    x0, x1, wx0, wx1 = taps(2 * w, w)
    top = x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1
    bot = x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1
    return top * wy0.reshape(1, 1, -1, 1) + bot * wy1.reshape(1, 1, -1, 1)


# ----------------------------------------------------------------------------
# a4  pairwise similarity volume, models/adamvs.py:464-478
# ----------------------------------------------------------------------------
def pair_similarity_volume(ref_fea, src_fea, R, t, depth_values):
    """sim[:,d] = mean_c(ref[c] * warp_d(src)[c])  -> [B,D,h,w]."""
    D = depth_values.shape[1]
    sims = []
    for d in range(D):
        warped = _WARP[0](src_fea, R, t, depth_values[:, d])
        sims.append((ref_fea * warped).mean(dim=1))
    return torch.stack(sims, dim=1)


# ----------------------------------------------------------------------------
# a5  CostRegNet2D, models/adamvs.py:198-238; blocks module.py:254-261
# ----------------------------------------------------------------------------
def _bn(x, sd, pre):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"],
                        sd[pre + "weight"], sd[pre + "bias"], False, 0.0, 1e-5)


def _conv_bn_relu(x, sd, pre, stride=1):
    return F.relu(_bn(F.conv2d(x, sd[pre + "conv.weight"], None, stride, 1), sd, pre + "bn."))


def _convt_bn_relu(x, sd, pre):
    y = F.conv_transpose2d(x, sd[pre + "0.weight"], None, stride=2, padding=1, output_padding=1)
    return F.relu(_bn(y, sd, pre + "1."))


def cost_reg_net_2d(x, sd, pre):
    """Depth-as-channel hourglass on x [B,D,h,w]; `pre` e.g. 'DepthNet.0.reg.'."""
    conv0 = _conv_bn_relu(x, sd, pre + "conv0.")
    conv2 = _conv_bn_relu(_conv_bn_relu(conv0, sd, pre + "conv1.", 2), sd, pre + "conv2.")
    conv4 = _conv_bn_relu(_conv_bn_relu(conv2, sd, pre + "conv3.", 2), sd, pre + "conv4.")
    y = _conv_bn_relu(_conv_bn_relu(conv4, sd, pre + "conv5.", 2), sd, pre + "conv6.")
    y = conv4 + _convt_bn_relu(y, sd, pre + "conv7.")
    y = conv2 + _convt_bn_relu(y, sd, pre + "conv9.")
    y = conv0 + _convt_bn_relu(y, sd, pre + "conv11.")
    return F.conv2d(y, sd[pre + "prob.weight"], sd[pre + "prob.bias"], 1, 1)


# ----------------------------------------------------------------------------
# a6  softmax / max / depth_regression, adamvs.py:481-486, module.py:617-625
# ----------------------------------------------------------------------------
def softmax_max_regress(score, depth_values):
    """-> (view_weight [B,1,h,w], pair_depth [B,h,w])."""
    prob = F.softmax(score, dim=1)
    view_weight = prob.max(1)[0].unsqueeze(1)
    pair_depth = torch.sum(prob * depth_values, 1)
    return view_weight, pair_depth


# ----------------------------------------------------------------------------
# a8  confidence-weighted aggregation, models/adamvs.py:495-512
# ----------------------------------------------------------------------------
def aggregate_similarity(ref_fea, src_feas, Rs, ts, depth_plane, view_weights):
    """sim[c] = sum_i w_i.warp_i[c].ref[c] / (1e-5 + sum_i w_i)  -> [B,C,h,w]."""
    sim_sum = 0
    w_sum = 1e-5
    for src, R, t, wgt in zip(src_feas, Rs, ts, view_weights):
        warped = _WARP[0](src, R, t, depth_plane)
        sim_sum = sim_sum + (warped * ref_fea) * wgt
        w_sum = w_sum + wgt
    return sim_sum / w_sum


# ----------------------------------------------------------------------------
# a9  SliceCostRegNetRED (adamvs.py:415-424) and ConvGRUCell (module.py:24-52)
# ----------------------------------------------------------------------------
def conv_gru_cell(x, h, sd, pre):
    hc = h.shape[1]
    gates = F.conv2d(torch.cat((x, h), 1), sd[pre + "conv_gates.0.weight"], sd[pre + "conv_gates.0.bias"], 1, 1)
    r = torch.sigmoid(gates[:, :hc])
    u = torch.sigmoid(gates[:, hc:])
    c = torch.tanh(F.conv2d(torch.cat((x, r * h), 1), sd[pre + "convc.0.weight"], sd[pre + "convc.0.bias"], 1, 1))
    return u * h + (1 - u) * c


def slice_reg_step(cost, state1, state2, sd, pre, in_up):
    """One recurrent step; `pre` e.g. 'DepthNet.0.reg_fuse.'.
    -> (reg_cost [B,1,2h,2w] or [B,1,h,w], state1, state2)."""
    c1 = F.relu(F.conv2d(cost, sd[pre + "conv1.conv.weight"], None, 1, 1))
    state1 = conv_gru_cell(c1, state1, sd, pre + "conv_gru1.")
    c2 = F.relu(F.conv2d(state1, sd[pre + "conv2.conv.weight"], None, 2, 1))
    state2 = conv_gru_cell(c2, state2, sd, pre + "conv_gru2.")
    up1 = F.conv_transpose2d(state2, sd[pre + "upconv1.weight"], sd[pre + "upconv1.bias"],
                             stride=2, padding=1, output_padding=1)
    s = F.relu(up1 + state1)
    if in_up:
        reg = F.conv_transpose2d(s, sd[pre + "upconv2d.weight"], sd[pre + "upconv2d.bias"],
                                 stride=2, padding=1, output_padding=1)
    else:
        reg = F.conv2d(s, sd[pre + "upconv2d.weight"], sd[pre + "upconv2d.bias"], 1, 1)
    return reg, state1, state2


# ----------------------------------------------------------------------------
# InferDepthNet0.forward, models/adamvs.py:433-533  (a4-a10 in sequence)
# ----------------------------------------------------------------------------
def infer_depth_stage(features, proj_matrices, depth_values, sd, pre, in_up, confidence_map=None):
    """features: list of V [B,C,h,w]; proj_matrices [B,V,4,4]; depth_values
    [B,D,h,w]; confidence_map: list of S maps from the previous stage or None.

    Returns the reference's dict; `pair_confidence` holds only the S maps the
    next stage reads (the reference's list also carries S.D duplicates, quirk
    Q1 of SURVEY.md section 8a)."""
    projs = torch.unbind(proj_matrices, 1)
    ref_fea, src_feas = features[0], features[1:]
    B, C, h, w = ref_fea.shape
    D = depth_values.shape[1]
    Rs, ts = [], []
    for sp in projs[1:]:
        R, t = relative_transform(sp, projs[0])
        Rs.append(R)
        ts.append(t)

    pair_results = []
    if confidence_map is None:                                   # adamvs.py:462-490
        weights = []
        for src, R, t in zip(src_feas, Rs, ts):
            sim = pair_similarity_volume(ref_fea, src, R, t, depth_values)
            score = cost_reg_net_2d(sim, sd, pre + "reg.")
            vw, pd = softmax_max_regress(score, depth_values)
            weights.append(vw)
            pair_results.append(pd)
    else:                                                        # adamvs.py:505
        weights = [F.interpolate(c, [h, w], mode="bilinear", align_corners=False)
                   for c in confidence_map[:len(src_feas)]]

    state1 = torch.zeros(B, 8, h, w)
    state2 = torch.zeros(B, 16, h // 2, w // 2)
    Ho, Wo = (2 * h, 2 * w) if in_up else (h, w)
    exp_sum = torch.zeros(B, 1, Ho, Wo)
    depth_image = torch.zeros(B, 1, Ho, Wo)
    max_prob = torch.zeros(B, 1, Ho, Wo)
    for d in range(D):                                           # adamvs.py:495-527
        plane = depth_values[:, d:d + 1]
        sim = aggregate_similarity(ref_fea, src_feas, Rs, ts, plane[:, 0], weights)
        reg, state1, state2 = slice_reg_step(sim, state1, state2, sd, pre + "reg_fuse.", in_up)
        prob = reg.exp()                                         # no max-subtraction (Q5)
        flag = (max_prob < prob).float()
        max_prob = flag * prob + (1 - flag) * max_prob
        if in_up:
            plane = F.interpolate(plane, [Ho, Wo], mode="bilinear", align_corners=False)
        depth_image = plane * prob + depth_image
        exp_sum = exp_sum + prob
    denom = exp_sum + 1e-10                                      # adamvs.py:529-531
    return {"depth": (depth_image / denom).squeeze(1),
            "photometric_confidence": (max_prob / denom).squeeze(1),
            "pair_confidence": weights, "pair_result": pair_results}


# ----------------------------------------------------------------------------
# FeatureNet0, models/adamvs.py:49-152 (context: feeds the hot path)
# ----------------------------------------------------------------------------
def _f_conv(x, sd, pre, stride=1, pad=1):
    return F.relu(_bn(F.conv2d(x, sd[pre + "conv.weight"], None, stride, pad), sd, pre + "bn."))


def _f_deconv_fuse(x_pre, x, sd, pre):
    hh, ww = x.shape[2:]
    y = F.conv_transpose2d(x, sd[pre + "deconv.conv.weight"], None, stride=2, padding=1, output_padding=1)
    y = F.relu(_bn(y[:, :, :2 * hh, :2 * ww].contiguous(), sd, pre + "deconv.bn."))
    return _f_conv(torch.cat((y, x_pre), 1), sd, pre + "conv.")


def _f_branches(feat, sd, pre, k):
    size = feat.shape[2:]
    outs = []
    for j, pool in ((1, 4), (2, 8)):
        b = F.avg_pool2d(feat, pool, pool)
        b = _f_conv(b, sd, "%sbranch%d_%d.1." % (pre, k, j), 1, 0)
        outs.append(F.interpolate(b, size=size, mode="bilinear", align_corners=False))
    return torch.cat((outs[0], outs[1], feat), 1)


def feature_net(x, sd, pre="feature."):
    c0 = _f_conv(_f_conv(x, sd, pre + "conv0.0."), sd, pre + "conv0.1.")
    c1 = _f_conv(c0, sd, pre + "conv1.0.", 2, 2)
    c1 = _f_conv(_f_conv(c1, sd, pre + "conv1.1."), sd, pre + "conv1.2.")
    c2 = _f_conv(c1, sd, pre + "conv2.0.", 2, 2)
    c2 = _f_conv(_f_conv(c2, sd, pre + "conv2.1."), sd, pre + "conv2.2.")
    out = {}
    out["stage1"] = F.conv2d(_f_branches(c2, sd, pre, 1), sd[pre + "out1.weight"])
    f = _f_deconv_fuse(c1, c2, sd, pre + "deconv1.")
    out["stage2"] = F.conv2d(_f_branches(f, sd, pre, 2), sd[pre + "out2.weight"])
    f = _f_deconv_fuse(c0, f, sd, pre + "deconv2.")
    out["stage3"] = F.conv2d(_f_branches(f, sd, pre, 3), sd[pre + "out3.weight"])
    return out


# ----------------------------------------------------------------------------
# a1  Infer_AdaMVSNet.forward, models/adamvs.py:567-620
# ----------------------------------------------------------------------------
def infer_adamvs_forward(imgs, proj_matrices, depth_values, sd, num_depth, ndepths,
                         depth_intervals_ratio, features=None):
    """Whole forward on CPU.  `features` (list over views of stage dicts) may be
    passed in to time / check the hot path without FeatureNet0."""
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
    depth_min = float(depth_values[0, 0])
    depth_max = float(depth_values[0, -1])
    depth_interval = (depth_max - depth_min) / num_depth         # batch item 0 only (Q4)
    if features is None:
        features = [feature_net(imgs[:, v], sd) for v in range(imgs.shape[1])]
    B, H, W = imgs.shape[0], imgs.shape[3], imgs.shape[4]
    outputs = {}
    depth, conf = None, None
    for s in range(len(ndepths)):
        name = "stage%d" % (s + 1)
        feats = [f[name] for f in features]
        scale = (4, 2, 1)[s]
        if depth is not None:
            cur, shape = depth, [B, depth.shape[1], depth.shape[2]]
        else:
            cur, shape = depth_values, [B, H // scale, W // scale]
        planes = depth_range_samples(cur, ndepths[s], depth_intervals_ratio[s] * depth_interval, shape)
        st = infer_depth_stage(feats, proj_matrices[name], planes, sd, "DepthNet.%d." % s,
                               in_up=(s < 2), confidence_map=conf)
        depth, conf = st["depth"], st["pair_confidence"]
        outputs[name] = st
        outputs.update(st)
    return outputs
