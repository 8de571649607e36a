"""CPU restatement of the reference's second model, MS-REDNet inference (models/msrednet.py:330-526,
models/module.py:54-106) -- SURVEY.md section 8f row f3.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing
else); the product path (ada-mvs_amd/models/msrednet.py) never calls it.

Functional form over a reference-keyed state dict, like oracle/adamvs_oracle.py, whose warp and hypothesis-plane
functions it shares (same models/module.py code in the reference).  Pinned by tests/golden/msred_*.npz, which
tools/gen_golden_msred.py produced by running the reference's own classes.
"""
import torch
import torch.nn.functional as F

from . import adamvs_oracle as ao


# ----------------------------------------------------------------------------
# FeatureNet (arch_mode 'unet'), models/msrednet.py:29-127
# ----------------------------------------------------------------------------
def feature_net_unet(x, sd, pre="feature."):
    c0 = ao._f_conv(ao._f_conv(x, sd, pre + "conv0.0."), sd, pre + "conv0.1.")
    c1 = ao._f_conv(c0, sd, pre + "conv1.0.", 2, 2)
    c1 = ao._f_conv(ao._f_conv(c1, sd, pre + "conv1.1."), sd, pre + "conv1.2.")
    c2 = ao._f_conv(c1, sd, pre + "conv2.0.", 2, 2)
    c2 = ao._f_conv(ao._f_conv(c2, sd, pre + "conv2.1."), sd, pre + "conv2.2.")
    out = {"stage1": F.conv2d(c2, sd[pre + "out1.weight"])}
    f = ao._f_deconv_fuse(c1, c2, sd, pre + "deconv1.")
    out["stage2"] = F.conv2d(f, sd[pre + "out2.weight"])
    f = ao._f_deconv_fuse(c0, f, sd, pre + "deconv2.")
    out["stage3"] = F.conv2d(f, sd[pre + "out3.weight"])
    return out


# ----------------------------------------------------------------------------
# FeatureNet (arch_mode 'fpn'), models/msrednet.py:74-91 (layers), 115-125 (forward): lateral 1x1 convolutions with bias
# added to the nearest-neighbour 2x upsampling of the coarser map, 3x3 output convolutions without bias
# ----------------------------------------------------------------------------
def feature_net_fpn(x, sd, pre="feature."):
    c0 = ao._f_conv(ao._f_conv(x, sd, pre + "conv0.0."), sd, pre + "conv0.1.")
    c1 = ao._f_conv(c0, sd, pre + "conv1.0.", 2, 2)
    c1 = ao._f_conv(ao._f_conv(c1, sd, pre + "conv1.1."), sd, pre + "conv1.2.")
    c2 = ao._f_conv(c1, sd, pre + "conv2.0.", 2, 2)
    c2 = ao._f_conv(ao._f_conv(c2, sd, pre + "conv2.1."), sd, pre + "conv2.2.")
    out = {"stage1": F.conv2d(c2, sd[pre + "out1.weight"])}
    f = c2.repeat_interleave(2, 2).repeat_interleave(2, 3) + F.conv2d(c1, sd[pre + "inner1.weight"], sd[pre + "inner1.bias"])
    out["stage2"] = F.conv2d(f, sd[pre + "out2.weight"], padding=1)
    f = f.repeat_interleave(2, 2).repeat_interleave(2, 3) + F.conv2d(c0, sd[pre + "inner2.weight"], sd[pre + "inner2.bias"])
    out["stage3"] = F.conv2d(f, sd[pre + "out3.weight"], padding=1)
    return out


# ----------------------------------------------------------------------------
# ConvGRUCell2, models/module.py:54-106: GroupNorm(1 group) on both gates and on the candidate
# ----------------------------------------------------------------------------
def conv_gru_cell2(x, h, sd, pre):
    hc = h.shape[1]
    f = F.conv2d(torch.cat((x, h), 1), sd[pre + "gate_conv.weight"], sd[pre + "gate_conv.bias"], padding=1)
    r = torch.sigmoid(F.group_norm(f[:, :hc], 1, sd[pre + "reset_gate_norm.weight"], sd[pre + "reset_gate_norm.bias"], 1e-5))
    u = torch.sigmoid(F.group_norm(f[:, hc:], 1, sd[pre + "update_gate_norm.weight"], sd[pre + "update_gate_norm.bias"], 1e-5))
    o = F.conv2d(torch.cat((x, r * h), 1), sd[pre + "output_conv.weight"], sd[pre + "output_conv.bias"], padding=1)
    y = torch.tanh(F.group_norm(o, 1, sd[pre + "output_norm.weight"], sd[pre + "output_norm.bias"], 1e-5))
    return u * h + (1 - u) * y


# ----------------------------------------------------------------------------
# slice_RED_Regularization.forward, models/msrednet.py:349-366
# ----------------------------------------------------------------------------
def slice_red_step(cost, states, sd, pre):
    """cost [B,C,h,w] (the variance), states = [s1 (8 ch), s2 (16, /2), s3 (32, /4), s4 (64, /8)]
    -> (reg_cost [B,1,h,w], new states)."""
    def down(x, name):
        return F.relu(F.conv2d(x, sd[pre + name + ".conv.weight"], None, 2, 1))

    def up(x, name):
        return F.relu(F.conv_transpose2d(x, sd[pre + name + ".conv.weight"], None, stride=2, padding=1, output_padding=1))

    s1, s2, s3, s4 = states
    c1 = down(-cost, "conv1")
    c2 = down(c1, "conv2")
    c3 = down(c2, "conv3")
    s4 = conv_gru_cell2(c3, s4, sd, pre + "conv_gru4.")
    s3 = conv_gru_cell2(c2, s3, sd, pre + "conv_gru3.")
    u2 = up(up(s4, "upconv3") + s3, "upconv2")
    s2 = conv_gru_cell2(c1, s2, sd, pre + "conv_gru2.")
    u1 = up(u2 + s2, "upconv1")
    s1 = conv_gru_cell2(-cost, s1, sd, pre + "conv_gru1.")
    reg = F.conv_transpose2d(u1 + s1, sd[pre + "upconv2d.weight"], sd[pre + "upconv2d.bias"], stride=1, padding=1)
    return reg, [s1, s2, s3, s4]


def variance_cost(ref_fea, src_feas, Rs, ts, depth_plane):
    """models/msrednet.py:396-412: E[x^2] - E[x]^2 over the reference and the warped source features of one plane."""
    total, sq = ref_fea.clone(), ref_fea ** 2
    for src, R, t in zip(src_feas, Rs, ts):
        wv = ao._WARP[0](src, R, t, depth_plane)
        total = total + wv
        sq = sq + wv ** 2
    n = len(src_feas) + 1
    return sq / n - (total / n) ** 2


# ----------------------------------------------------------------------------
# InferDepthNet.forward, models/msrednet.py:373-436
# ----------------------------------------------------------------------------
def infer_depth_stage_red(features, proj_matrices, depth_values, sd, pre):
    """features: V x [B,C,h,w]; proj_matrices [B,V,4,4]; depth_values [B,D,h,w] -> depth, photometric_confidence [B,h,w]."""
    ref, srcs = features[0], features[1:]
    B, C, h, w = ref.shape
    rel = [ao.relative_transform(proj_matrices[:, v], proj_matrices[:, 0]) for v in range(1, len(features))]
    Rs, ts = [r[0] for r in rel], [r[1] for r in rel]
    states = [torch.zeros(B, 8 << k, h >> k, w >> k) for k in range(4)]
    exp_sum = torch.zeros(B, 1, h, w)
    depth_image = torch.zeros(B, 1, h, w)
    max_prob = torch.zeros(B, 1, h, w)
    for d in range(depth_values.shape[1]):
        plane = depth_values[:, d:d + 1]
        reg, states = slice_red_step(variance_cost(ref, srcs, Rs, ts, plane), states, sd, pre)
        prob = reg.exp()
        flag = (max_prob < prob).float()
        max_prob = flag * prob + (1 - flag) * max_prob
        depth_image = plane * prob + depth_image
        exp_sum = exp_sum + prob
    denom = exp_sum + 1e-10
    return {"depth": (depth_image / denom).squeeze(1), "photometric_confidence": (max_prob / denom).squeeze(1)}


# ----------------------------------------------------------------------------
# Infer_CascadeREDNet.forward, models/msrednet.py:474-526
# ----------------------------------------------------------------------------
def infer_cascade_rednet_forward(imgs, proj_matrices, depth_values, sd, num_depth, ndepths, depth_intervals_ratio):
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
    depth_min, depth_max = float(depth_values[0, 0]), float(depth_values[0, -1])
    depth_interval = (depth_max - depth_min) / num_depth
    features = [feature_net_unet(imgs[:, v], sd) for v in range(imgs.shape[1])]
    B, H, W = imgs.shape[0], imgs.shape[3], imgs.shape[4]
    outputs, depth = {}, None
    for s in range(len(ndepths)):
        name = "stage%d" % (s + 1)
        scale = (4, 2, 1)[s]
        if depth is not None:                                    # msrednet.py:495-501: to full resolution first
            cur = F.interpolate(depth.unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
        else:
            cur = depth_values
        planes = ao.depth_range_samples(cur, ndepths[s], depth_intervals_ratio[s] * depth_interval, [B, H, W])
        planes = F.interpolate(planes.unsqueeze(1), [ndepths[s], H // scale, W // scale], mode="trilinear",
                               align_corners=False).squeeze(1)    # msrednet.py:512-514: down to the stage's resolution
        st = infer_depth_stage_red([f[name] for f in features], proj_matrices[name], planes, sd,
                                   "cost_regularization.%d." % s)
        depth = st["depth"]
        outputs[name] = st
        outputs.update(st)
    return outputs
