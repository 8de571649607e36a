"""Thin PyTorch-tensor wrappers over the C ABI (include/adamvs_hip.h).

PyTorch is used for device memory and the current stream only; every
computation below happens in libadamvs_hip.so.  All functions require CUDA
(ROCm) fp32 tensors and raise otherwise -- there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import FuseWeights, StageDesc, check


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.AdaMVSHipError("%s must be a GPU tensor: the Ada-MVS hot path has no CPU fallback" % name)
    if t.dtype != torch.float32:
        raise _lib.AdaMVSHipError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def relative_transforms(proj):
    """proj [B,V,4,4] -> rt [B,V-1,12]   (module.py:539-541)"""
    proj = _dev(proj, "proj")
    B, V = proj.shape[:2]
    rt = torch.empty(B, V - 1, 12, device=proj.device, dtype=torch.float32)
    check(_lib.load().adamvs_relative_transforms(_p(proj), _p(rt), B, V, _stream()), "relative_transforms")
    return rt


def pack_features(x, out=None):
    """[N,C,h,w] -> channel-last [N,h*w,C]"""
    x = _dev(x, "features")
    N, C, h, w = x.shape
    if out is None:
        out = torch.empty(N, h * w, C, device=x.device, dtype=torch.float32)
    check(_lib.load().adamvs_pack_features(_p(x), _p(out), N, C, h, w, _stream()), "pack_features")
    return out


def unpack_features(x, h, w):
    """channel-last [N,h*w,C] -> [N,C,h,w]"""
    x = _dev(x, "features")
    N, hw, C = x.shape
    out = torch.empty(N, C, h, w, device=x.device, dtype=torch.float32)
    check(_lib.load().adamvs_unpack_features(_p(x), _p(out), N, C, h, w, _stream()), "unpack_features")
    return out


def depth_range_samples(cur_depth, ndepth, depth_interval_pixel, shape):
    """get_depth_range_samples (module.py:646-663) -> [B,D,h,w]"""
    cur_depth = _dev(cur_depth, "cur_depth")
    B, h, w = shape
    out = torch.empty(B, ndepth, h, w, device=cur_depth.device, dtype=torch.float32)
    lib = _lib.load()
    if cur_depth.dim() == 2:
        if cur_depth.shape[1] != 2:        # the reference reads [:,0] and [:,-1] only
            cur_depth = torch.stack((cur_depth[:, 0], cur_depth[:, -1]), 1).contiguous()
        check(lib.adamvs_depth_range_samples_uniform(_p(cur_depth), _p(out), B, ndepth, h, w, _stream()),
              "depth_range_samples_uniform")
    else:
        if tuple(cur_depth.shape) != (B, h, w):
            raise _lib.AdaMVSHipError("cur_depth:%s, input shape:%s" % (tuple(cur_depth.shape), shape))
        check(lib.adamvs_depth_range_samples_window(_p(cur_depth), float(depth_interval_pixel), _p(out), B, ndepth, h, w,
                                                    _stream()), "depth_range_samples_window")
    return out


def resize_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) for [N,1,h,w] / [N,h,w]"""
    x = _dev(x, "x")
    hi, wi = x.shape[-2:]
    ho, wo = size
    N = x.numel() // (hi * wi)
    out = torch.empty(x.shape[:-2] + (ho, wo), device=x.device, dtype=torch.float32)
    check(_lib.load().adamvs_resize_bilinear(_p(x), _p(out), N, hi, wi, ho, wo, _stream()), "resize_bilinear")
    return out


def depth_regression(p, depth_values):
    """module.py:617-625"""
    p = _dev(p, "p")
    depth_values = _dev(depth_values, "depth_values")
    B, D, h, w = p.shape
    out = torch.empty(B, h, w, device=p.device, dtype=torch.float32)
    hd, wd = (0, 0) if depth_values.dim() <= 2 else depth_values.shape[2:]
    check(_lib.load().adamvs_depth_regression(_p(p), _p(depth_values), _p(out), B, D, h, w, hd, wd, _stream()),
          "depth_regression")
    return out


def homo_warp(src_fea, rt, depth_values):
    """homo_warping_float body (module.py:543-566): src [B,C,h,w], rt [B,12], depth [B,Nd,h,w] -> [B,C,Nd,h,w]"""
    src_fea = _dev(src_fea, "src_fea")
    rt = _dev(rt, "rt")
    depth_values = _dev(depth_values, "depth_values")
    B, C, h, w = src_fea.shape
    Nd = depth_values.shape[1]
    out = torch.empty(B, C, Nd, h, w, device=src_fea.device, dtype=torch.float32)
    check(_lib.load().adamvs_homo_warp(_p(src_fea), _p(rt), _p(depth_values), _p(out), B, C, Nd, h, w, _stream()), "homo_warp")
    return out


def pair_similarity(feat, rt, planes, B, S, C, D, h, w):
    """feat [V*B,hw,C] (view-major), rt [B,S,12], planes [B,D,h,w] -> sim [S*B,hw,D]"""
    sim = torch.empty(S * B, h * w, D, device=feat.device, dtype=torch.float32)
    check(_lib.load().adamvs_pair_similarity(_p(_dev(feat, "feat")), _p(_dev(rt, "rt")), _p(_dev(planes, "planes")), _p(sim),
                                             B, S, C, D, h, w, _stream()), "pair_similarity")
    return sim


def cost_reg_net_2d(x_cl, wpk, h, w, precision=0):
    """x_cl [N,hw,D] channel-last -> score [N,hw,D]   (adamvs.py:229-238)"""
    x_cl = _dev(x_cl, "x")
    N, hw, D = x_cl.shape
    lib = _lib.load()
    nbytes = lib.adamvs_cost_reg_net_2d_workspace_bytes(N, D, h, w)
    ws = torch.empty(nbytes // 4, device=x_cl.device, dtype=torch.float32)
    score = torch.empty_like(x_cl)
    check(lib.adamvs_cost_reg_net_2d(_p(x_cl), _p(wpk), wpk.numel(), _p(score), N, D, h, w, int(precision), _p(ws), nbytes, _stream()),
          "cost_reg_net_2d")
    return score


def softmax_max_regress(score, planes, S, B, D, h, w):
    vw = torch.empty(S, B, h, w, device=score.device, dtype=torch.float32)
    pd = torch.empty(S, B, h, w, device=score.device, dtype=torch.float32)
    check(_lib.load().adamvs_softmax_max_regress(_p(_dev(score, "score")), _p(_dev(planes, "planes")), _p(vw), _p(pd),
                                                 S, B, D, h, w, _stream()), "softmax_max_regress")
    return vw, pd


def prob_softmax_regress(x_cl, wpk_layer, bias, planes, S, B, D, h, w, precision=0):
    """The `prob` layer of CostRegNet2D with softmax / max / depth regression in its epilogue (what the stage runs):
    x_cl [S*B, h*w, D], planes [B, D, h, w] -> (view_weight, pair_depth) [S, B, h, w]."""
    vw = torch.empty(S, B, h, w, device=x_cl.device, dtype=torch.float32)
    pd = torch.empty(S, B, h, w, device=x_cl.device, dtype=torch.float32)
    check(_lib.load().adamvs_prob_softmax_regress(_p(_dev(x_cl, "x")), _p(wpk_layer), _p(bias), _p(_dev(planes, "planes")), _p(vw), _p(pd),
                                                  S, B, D, h, w, int(precision), _stream()), "prob_softmax_regress")
    return vw, pd


def prob_softmax_regress_wino(x_cl, wino_layer, bias, depth_range, S, B, D, h, w):
    """The same with `prob` in the F(2x2, 3x3) form (wino_layer from packing.pack_reg_layer_wino): per-lane softmax partials and a
    merge kernel instead of the score volume -- what the fp32 stage runs at D a multiple of 64.  depth_range [B, 2]: first and
    last hypothesis plane of every tile (stage 1's uniform planes)."""
    vw = torch.empty(S, B, h, w, device=x_cl.device, dtype=torch.float32)
    pd = torch.empty(S, B, h, w, device=x_cl.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.adamvs_prob_softmax_regress_wino_workspace_bytes(S, B, D, h, w)
    ws = torch.empty(nbytes // 4, device=x_cl.device, dtype=torch.float32)
    check(lib.adamvs_prob_softmax_regress_wino(_p(_dev(x_cl, "x")), _p(wino_layer), _p(bias), _p(_dev(depth_range, "depth_range")), _p(vw), _p(pd),
                                               S, B, D, h, w, ctypes.c_void_p(ws.data_ptr()), nbytes, _stream()), "prob_softmax_regress_wino")
    return vw, pd


def aggregate_conv1(feat, rt, planes, view_weight, w1pk, B, S, C, D, h, w, precision=0, return_similarity=False):
    """-> c1 [D,B,hw,8]; return_similarity (D <= 32, one chunk): also the aggregated similarity [D,B,hw,C] the call left in
    its workspace (include/adamvs_hip.h: the workspace holds the last chunk's similarity on return)."""
    c1 = torch.empty(D, B, h * w, 8, device=feat.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.adamvs_aggregate_conv1_workspace_bytes(B, C, D, h, w)
    ws = torch.empty(max(nbytes // 4, 1), device=feat.device, dtype=torch.float32)
    check(_lib.load().adamvs_aggregate_conv1(_p(_dev(feat, "feat")), _p(_dev(rt, "rt")), _p(_dev(planes, "planes")),
                                             _p(_dev(view_weight, "view_weight")), _p(w1pk), _p(c1), B, S, C, D, h, w, int(precision),
                                             _p(ws), nbytes, _stream()), "aggregate_conv1")
    if return_similarity:
        if D > 32:
            raise _lib.AdaMVSHipError("aggregate_conv1: return_similarity needs D <= 32 (one chunk of planes in the workspace)")
        return c1, ws[:D * B * h * w * C].reshape(D, B, h * w, C)
    return c1


class PackedFuse:
    """Device copy of a packed SliceCostRegNetRED + its adamvs_fuse_weights struct."""

    def __init__(self, flat, offsets, device):
        self.buf = flat.to(device)
        base = self.buf.data_ptr()
        self.struct = FuseWeights(**{f: base + 4 * o for f, o in offsets.items()})

    def ptr(self):
        return ctypes.byref(self.struct)

    def field(self, name):
        return ctypes.c_void_p(getattr(self.struct, name))


class PackedFeature:
    """Device copy of a packed FeatureNet0 + its adamvs_feature_weights struct."""

    def __init__(self, flat, offsets, device):
        from ._lib import FeatureWeights, FConvWeights, ContextWeights
        self.buf = flat.to(device)
        base = self.buf.data_ptr()
        at = lambda f: base + 4 * offsets[f]
        kw = {n: FConvWeights(at(n + ".w"), at(n + ".b")) for n in packing_fields()[0]}
        kw.update({n: ContextWeights(at(n + ".w1"), at(n + ".b1"), at(n + ".w2")) for n in packing_fields()[1]})
        self.struct = FeatureWeights(**kw)

    def ptr(self):
        return ctypes.byref(self.struct)


def packing_fields():
    from . import packing
    return packing.FEATURE_CONVS, packing.FEATURE_BRANCHES


def feature_net0_workspace_bytes(N, H, W):
    return int(_lib.load().adamvs_feature_net0_workspace_bytes(int(N), int(H), int(W)))


def feature_net0(imgs, packed, workspace=None, out=None, views=None):
    """FeatureNet0.forward on [N,3,H,W] images -> channel-last (stage1 [N,hw/16,32], stage2 [N,hw/4,16], stage3 [N,hw,8]).
    out: the three (contiguous) result tensors, e.g. slices of the maps of a larger batch run in chunks.
    views = (n0, n): imgs is [B,V,3,H,W] as the reference's forward() receives it; images n0 .. n0+n-1 of the V*B images in
    view-major order (m = v*B + b) are computed, read in place (no transposed copy)."""
    lib = _lib.load()
    imgs = _dev(imgs, "imgs")
    if views is not None:
        Bv, Vv, c, H, W = imgs.shape
        n0, N = views
    else:
        N, c, H, W = imgs.shape
    if c != 3:
        check(-1, "feature_net0")
    dev = imgs.device
    if out is None:
        s1 = torch.empty(N, (H // 4) * (W // 4), 32, device=dev, dtype=torch.float32)
        s2 = torch.empty(N, (H // 2) * (W // 2), 16, device=dev, dtype=torch.float32)
        s3 = torch.empty(N, H * W, 8, device=dev, dtype=torch.float32)
    else:
        s1, s2, s3 = out
        want = ((N, (H // 4) * (W // 4), 32), (N, (H // 2) * (W // 2), 16), (N, H * W, 8))
        if any(tuple(t.shape) != w or not t.is_contiguous() or t.dtype != torch.float32 or t.device != dev for t, w in zip(out, want)):
            raise _lib.AdaMVSHipError("feature_net0: out tensors must be contiguous float32 %s on %s" % (want, dev))
    nbytes = lib.adamvs_feature_net0_workspace_bytes(N, H, W)
    if workspace is None or workspace.numel() * 4 < nbytes:
        workspace = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
    if views is not None:
        check(lib.adamvs_feature_net0_views(_p(imgs), packed.ptr(), _p(s1), _p(s2), _p(s3), Bv, Vv, n0, N, H, W, _p(workspace), nbytes,
                                            _stream()), "feature_net0_views")
    else:
        check(lib.adamvs_feature_net0(_p(imgs), packed.ptr(), _p(s1), _p(s2), _p(s3), N, H, W, _p(workspace), nbytes, _stream()),
              "feature_net0")
    return s1, s2, s3


class PackedFeatureFpn:
    """Device copy of a packed FPN FeatureNet + its adamvs_feature_fpn_weights struct."""

    def __init__(self, flat, offsets, device):
        from ._lib import FeatureFpnWeights, FConvWeights
        from . import packing
        self.buf = flat.to(device)
        base = self.buf.data_ptr()
        at = lambda f: base + 4 * offsets[f]
        self.struct = FeatureFpnWeights(**{n: FConvWeights(at(n + ".w"), at(n + ".b")) for n in packing.FEATURE_FPN_CONVS})

    def ptr(self):
        return ctypes.byref(self.struct)


def feature_net_fpn_workspace_bytes(N, H, W):
    return int(_lib.load().adamvs_feature_net_fpn_workspace_bytes(int(N), int(H), int(W)))


def feature_net_fpn(imgs, packed, workspace=None):
    """FeatureNet(arch_mode="fpn").forward (reference models/msrednet.py:115-125) on [N,3,H,W] images -> channel-last maps."""
    lib = _lib.load()
    imgs = _dev(imgs, "imgs")
    N, c, H, W = imgs.shape
    if c != 3:
        check(-1, "feature_net_fpn")
    dev = imgs.device
    s1 = torch.empty(N, (H // 4) * (W // 4), 32, device=dev, dtype=torch.float32)
    s2 = torch.empty(N, (H // 2) * (W // 2), 16, device=dev, dtype=torch.float32)
    s3 = torch.empty(N, H * W, 8, device=dev, dtype=torch.float32)
    nbytes = lib.adamvs_feature_net_fpn_workspace_bytes(N, H, W)
    if workspace is None or workspace.numel() * 4 < nbytes:
        workspace = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
    check(lib.adamvs_feature_net_fpn(_p(imgs), packed.ptr(), _p(s1), _p(s2), _p(s3), N, H, W, _p(workspace), nbytes, _stream()),
          "feature_net_fpn")
    return s1, s2, s3


def slice_reg_step(cost_cl, state1, state2, fuse, B, C, h, w, in_up, precision=0):
    """SliceCostRegNetRED.forward on channel-last maps; states updated in place. -> reg [B,1,Ho,Wo]"""
    lib = _lib.load()
    Ho, Wo = (2 * h, 2 * w) if in_up else (h, w)
    reg = torch.empty(B, 1, Ho, Wo, device=cost_cl.device, dtype=torch.float32)
    nbytes = lib.adamvs_slice_reg_step_scratch_bytes(B, h, w)
    scratch = torch.empty(nbytes // 4, device=cost_cl.device, dtype=torch.float32)
    check(lib.adamvs_slice_reg_step(_p(_dev(cost_cl, "cost")), _p(state1), _p(state2), fuse.ptr(), _p(reg), B, C, h, w,
                                    int(in_up), int(precision), _p(scratch), nbytes, _stream()), "slice_reg_step")
    return reg


def stage_desc(B, S, C, h, w, D, in_up, first_stage, prev_hw=(0, 0), precision=0, precision_fuse=0, eps_in_numerator=0,
               plane_mode=_lib.PLANES_EXPLICIT, half_span=0.0, half_span_dev=None):
    """half_span_dev: a one-element float32 device tensor holding the half span (window planes) -- read by the kernels when they run,
    so that a captured graph serves any depth range; the caller keeps it alive."""
    return StageDesc(B, S, C, h, w, D, int(in_up), int(first_stage), int(prev_hw[0]), int(prev_hw[1]), int(precision),
                     int(precision_fuse), int(eps_in_numerator), int(plane_mode), float(half_span),
                     ctypes.c_void_p(_dev(half_span_dev, "half_span_dev").data_ptr()) if half_span_dev is not None else None)


def half_span_of(ndepth, depth_interval_pixel):
    """ndepth / 2 * depth_inteval_pixel formed in Python floats, the product rounded to fp32 where it meets the map (module.py:632)."""
    return float(ndepth / 2.0 * float(depth_interval_pixel))


def plane_source(cur_depth, ndepth, depth_interval_pixel, shape, span_dev=None):
    """What get_depth_range_samples (module.py:646-663) would materialise, as (plane_mode, half_span, tensor[, span_dev]) for
    adamvs_depth_stage_forward: a 2-D cur_depth [B, >=2] gives uniform planes over [min, max] (the interval is ignored
    there, quirk Q3), a map [B,h,w] gives the per-pixel window cur -+ ndepth / 2 * depth_interval_pixel.  The planes are
    generated inside the kernels, bit-identical to depth_range_samples()."""
    cur_depth = _dev(cur_depth, "cur_depth")
    B, h, w = shape
    if cur_depth.dim() == 2:
        if cur_depth.shape[1] != 2:        # the reference reads [:,0] and [:,-1] only
            cur_depth = torch.stack((cur_depth[:, 0], cur_depth[:, -1]), 1).contiguous()
        return _lib.PLANES_UNIFORM, 0.0, cur_depth
    if tuple(cur_depth.shape) != (B, h, w):
        raise _lib.AdaMVSHipError("cur_depth:%s, input shape:%s" % (tuple(cur_depth.shape), shape))
    if span_dev is not None:           # the half span lives in device memory (graphed.py): nothing of the depth range is baked into the launch
        return _lib.PLANES_WINDOW, 0.0, cur_depth, span_dev
    return _lib.PLANES_WINDOW, half_span_of(ndepth, depth_interval_pixel), cur_depth


def depth_stage_workspace_bytes(desc):
    n = _lib.load().adamvs_depth_stage_workspace_bytes(ctypes.byref(desc))
    if n == 0:
        check(-1, "depth_stage_workspace_bytes")
    return n


def conv3x3_dd_wino(x_cl, wino_layer, bias, skip, N, D, h, w, relu, out=None):
    """A stride-1 CostRegNet2D layer in the F(2x2, 3x3) form; wino_layer from packing.pack_reg_layer_wino."""
    if out is None:
        out = torch.empty(N, h * w, D, device=x_cl.device, dtype=torch.float32)
    null = ctypes.c_void_p(0)
    check(_lib.load().adamvs_conv3x3_dd_wino(_p(_dev(x_cl, "x")), _p(wino_layer), _p(bias), _p(skip) if skip is not None else null,
                                             _p(out), N, D, h, w, int(relu), _stream()), "conv3x3_dd_wino")
    return out


def conv3x3_dd(x_cl, wpk_layer, bias, skip, N, D, hi, wi, mode, relu, out=None, precision=0, in2=None):
    """One CostRegNet2D layer on channel-last maps (mode 0 stride 1, 1 stride 2, 2 transposed stride 2).
    skip: added to the output after the ReLU; in2: added to the input (the layer convolves x_cl + in2; fp32 only)."""
    ho, wo = (hi // 2, wi // 2) if mode == 1 else ((2 * hi, 2 * wi) if mode == 2 else (hi, wi))
    if out is None:
        out = torch.empty(N, ho * wo, D, device=x_cl.device, dtype=torch.float32)
    null = ctypes.c_void_p(0)
    check(_lib.load().adamvs_conv3x3_dd(_p(x_cl), _p(_dev(in2, "in2")) if in2 is not None else null, _p(wpk_layer), _p(bias),
                                        _p(skip) if skip is not None else null,
                                        _p(out), N, D, hi, wi, mode, int(relu), int(precision), _stream()), "conv3x3_dd")
    return out


def depth_stage_forward(desc, feat, rt, planes, prev_conf, w_reg, fuse, workspace=None, phases=_lib.PHASE_ALL, outputs=None,
                        timing_only=False):
    """InferDepthNet0.forward (adamvs.py:433-533).  Returns (view_weight [S,B,h,w], pair_depth or None,
    depth [B,Ho,Wo], confidence [B,Ho,Wo]).  timing_only: adamvs_bench_stage_phase -- the selected phases for their duration,
    no maps promised (bench.py's phase table)."""
    dev = feat.device
    B, S, h, w = desc.B, desc.S, desc.h, desc.w
    Ho, Wo = (2 * h, 2 * w) if desc.in_up else (h, w)
    nbytes = depth_stage_workspace_bytes(desc)
    if workspace is None or workspace.numel() * 4 < nbytes:
        workspace = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
    if outputs is not None:                 # piecewise (phase-by-phase) calls reuse one set of outputs
        vw, pd, depth, conf = outputs
    else:
        vw = torch.empty(S, B, h, w, device=dev, dtype=torch.float32)
        pd = torch.empty(S, B, h, w, device=dev, dtype=torch.float32) if desc.first_stage else None
        depth = torch.empty(B, Ho, Wo, device=dev, dtype=torch.float32)
        conf = torch.empty(B, Ho, Wo, device=dev, dtype=torch.float32)
    null = ctypes.c_void_p(0)
    entry = _lib.load().adamvs_bench_stage_phase if timing_only else _lib.load().adamvs_depth_stage_forward
    check(entry(
        ctypes.byref(desc), _p(_dev(feat, "feat")), _p(_dev(rt, "rt")), _p(_dev(planes, "planes")),
        _p(_dev(prev_conf, "prev_conf")) if prev_conf is not None else null,
        _p(w_reg) if w_reg is not None else null, w_reg.numel() if w_reg is not None else 0, fuse.ptr(),
        _p(vw), _p(pd) if pd is not None else null, _p(depth), _p(conf), int(phases), _p(workspace), nbytes, _stream()),
        "depth_stage_forward")
    return vw, pd, depth, conf


# ---- MS-REDNet pieces (csrc/msred.hip; reference models/msrednet.py:373-436, models/module.py:54-106) ---------------
def red_variance_cost(feat, rt, planes, out_a, out_b, B, S, C, D, h, w, negate=True):
    """-variance of (reference, warped sources) for the D planes [B,D,h*w] into channels [0,C) of out_a [D*B,h*w,Da]
    (plane-major) and out_b."""
    check(_lib.load().adamvs_red_variance_cost(_p(_dev(feat, "feat")), _p(rt), _p(_dev(planes, "planes")), _p(out_a), out_a.shape[-1],
                                               _p(out_b) if out_b is not None else ctypes.c_void_p(0),
                                               out_b.shape[-1] if out_b is not None else 0, B, S, C, D, h, w, int(negate),
                                               _stream()), "red_variance_cost")


def channel_copy(src, src_c0, dst, dst_c0, n):
    """dst[b, p, dst_c0:dst_c0+n] = src[b, p, src_c0:src_c0+n] for channel-last maps [N, npix, D]."""
    N, npix = src.shape[0], src.shape[1]
    check(_lib.load().adamvs_channel_copy(_p(src), _p(dst), N, npix, n, src.stride(0), src.stride(1), src_c0, dst.stride(0),
                                          dst.stride(1), dst_c0, _stream()), "channel_copy")


def planes_to_volume(src, vol, B):
    """vol[b, d, p] = src[d * B + b, p, 0]: the single real channel of the last decoder layer into the slice volume."""
    N, npix = src.shape[0], src.shape[1]
    D = N // B
    for b in range(B):
        check(_lib.load().adamvs_channel_copy(ctypes.c_void_p(src.data_ptr() + 4 * b * src.stride(0)),
                                              ctypes.c_void_p(vol.data_ptr() + 4 * b * vol.stride(0)), D, npix, 1,
                                              B * src.stride(0), src.stride(1), 0, vol.stride(1), 1, 0, _stream()), "channel_copy")


def group_stats_workspace(N, ngroups, device):
    return torch.empty(_lib.load().adamvs_group_stats_workspace_bytes(N, ngroups) // 8, device=device, dtype=torch.float64)


def group_stats_partial(x0, x1, n, partials):
    """Partial sums for GroupNorm(1 group) over channels [0, n) of x0 (and x1; views into a wider map are fine: the
    pixel stride is what counts): consumed by the gru2_* epilogues."""
    N, npix, D = x0.shape[0], x0.shape[1], x0.stride(1)
    check(_lib.load().adamvs_group_stats_partial(_p(x0), _p(x1) if x1 is not None else ctypes.c_void_p(0), N, npix, D, n,
                                                 _p(partials), partials.numel() * 8, _stream()), "group_stats_partial")


def group_stats_finish(partials, N, ngroups, npix, n, eps=1e-5):
    stats = torch.empty(N, ngroups, 2, device=partials.device, dtype=torch.float32)
    check(_lib.load().adamvs_group_stats_finish(_p(partials), _p(stats), N, ngroups, npix, n, eps, _stream()), "group_stats_finish")
    return stats


def gru2_gates_apply(fr, fu, partials, gn, h, rh, u, HC, eps=1e-5):
    """fr / fu: two maps of one width, or views of the two halves of one map (their last-but-one stride is the width)."""
    N, npix, W = h.shape
    check(_lib.load().adamvs_gru2_gates_apply(_p(fr), _p(fu), fr.stride(1), _p(partials), _p(gn), _p(h), _p(rh), _p(u), N, npix, W,
                                              HC, eps, _stream()), "gru2_gates_apply")


def conv3x3_pair(a, b, wpk, bias, cout, h, w, out=None):
    """conv3x3(cat(a, b)) + bias on compact channel-last maps [B, h*w, CA], [B, h*w, CB] -> [B, h*w, cout]."""
    B = a.shape[0]
    if out is None:
        out = torch.empty(B, h * w, cout, device=a.device, dtype=torch.float32)
    check(_lib.load().adamvs_conv3x3_pair(_p(_dev(a, "a")), a.shape[-1], _p(_dev(b, "b")), b.shape[-1], _p(wpk), _p(bias), _p(out),
                                          cout, B, h, w, _stream()), "conv3x3_pair")
    return out


def gru2_out_apply(o, partials, gn, u, h, out, HC, eps=1e-5):
    N, npix, W = o.shape
    check(_lib.load().adamvs_gru2_out_apply(_p(o), _p(partials), _p(gn), _p(u), _p(h),
                                            _p(out) if out is not None else ctypes.c_void_p(0),
                                            out.shape[-1] if out is not None else 0, N, npix, W, HC, eps, _stream()),
          "gru2_out_apply")


def red_recur_pair(x, wg, bg, wc, bc, gn, R, B, h, w, HC, eps=1e-5):
    """ConvGRUCell2 of a shallow level over all planes: x [D*B, h*w, Cx] compact -> R[:, :, :HC]."""
    D, Cx = x.shape[0] // B, x.shape[-1]
    lib = _lib.load()
    nbytes = lib.adamvs_red_recur_workspace_bytes(B, h, w, HC, 2 * HC, HC)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    check(lib.adamvs_red_recur_pair(_p(_dev(x, "x")), Cx, _p(wg), _p(bg), _p(wc), _p(bc), _p(gn), _p(R), R.shape[-1], B, D, h, w, HC,
                                    eps, _p(ws), nbytes, _stream()), "red_recur_pair")


def red_recur_split(gxr, gxu, cx, w_ghr, w_ghu, w_ch, gn, R, B, h, w, HC, eps=1e-5):
    """ConvGRUCell2 of a deep level over all planes from the precomputed x halves [D*B, h*w, W] -> R[:, :, :HC].
    w_*: the h halves as contiguous (9 W W + W)-float blocks."""
    D, W = gxr.shape[0] // B, gxr.shape[-1]
    lib = _lib.load()
    nbytes = lib.adamvs_red_recur_workspace_bytes(B, h, w, W, W, HC)
    ws = torch.empty(nbytes // 4, device=gxr.device, dtype=torch.float32)
    check(lib.adamvs_red_recur_split(_p(_dev(gxr, "gxr")), _p(gxu), _p(cx), _p(w_ghr), _p(w_ghu), _p(w_ch), _p(gn), _p(R),
                                     R.shape[-1], B, D, h, w, W, HC, eps, _p(ws), nbytes, _stream()), "red_recur_split")


def soft_argmin(vol, planes, B, D, h, w):
    depth = torch.empty(B, h, w, device=vol.device, dtype=torch.float32)
    conf = torch.empty(B, h, w, device=vol.device, dtype=torch.float32)
    check(_lib.load().adamvs_soft_argmin(_p(vol), _p(_dev(planes, "planes")), _p(depth), _p(conf), B, D, h, w, _stream()),
          "soft_argmin")
    return depth, conf
