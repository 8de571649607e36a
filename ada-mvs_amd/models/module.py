"""Mirror of the hot-path functions and layer containers of reference
models/module.py, backed by the HIP library.

Functions keep the reference names, argument order and tensor layouts
(homo_warping_float, depth_regression, get_depth_range_samples,
get_cur_depth_range_samples).  The nn.Module blocks below exist to own
parameters under the reference's state-dict key names.  FeatureNet0 runs on
the hand-written kernels of csrc/featnet.hip (`FeatureNet0.forward_cl`,
SURVEY.md section 8f1); its blocks (`Conv2d`, `Deconv2d`, `DeConv2dFuse`)
keep a PyTorch `forward` only for `FeatureNet0.forward_torch`, the explicit
reference form used by the tests.  The hot-path blocks (`ConvReLU`,
`ConvBnReLU`, `ConvGRUCell`) have no forward of their own: their arithmetic
is fused into the HIP kernels of the enclosing network.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ada_mvs_amd import hip_ops


def module_state(module):
    """state_dict() of `module` that also works on an nn.DataParallel replica.  torch.nn.parallel.replicate() empties a
    replica's `_parameters` and re-attaches the per-device copies as plain attributes (kept in `_former_parameters`), so
    `replica.state_dict()` holds buffers only (reference predict_whu.py:82 wraps the model in nn.DataParallel, which
    replicates on every forward as soon as two GPUs are visible)."""
    out = {}
    for name, m in module.named_modules():
        pre = name + "." if name else ""
        for k, v in m._parameters.items():
            if v is not None:
                out[pre + k] = v
        for k, v in getattr(m, "_former_parameters", {}).items():
            out.setdefault(pre + k, v)
        for k, v in m._buffers.items():
            if v is not None and k not in m._non_persistent_buffers_set:
                out[pre + k] = v
    return out


class PackedCache:
    """Mixin for the networks whose forward runs on packed weights (BatchNorm folded, MFMA fragment order).

    The cache is ONE dict per source module, keyed by device (and precision): a replica's __dict__ is a shallow copy,
    so every nn.DataParallel replica of every forward finds the weights its device packed before, and the per-device
    threads never evict each other.  Moving the module (.cuda() / .to()) or loading a state dict drops the cache;
    weights edited in place afterwards need an explicit drop_cache()."""

    def _cache_init(self):
        self._cache = {}

    def cached(self, key, build):
        c = self._cache
        if key not in c:
            c[key] = build()
        return c[key]

    def drop_cache(self):
        self._cache.clear()

    def _apply(self, fn, *a, **k):
        self.drop_cache()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.drop_cache()
        return super()._load_from_state_dict(*a, **k)


# ---- functions (reference models/module.py:527-663) -------------------------
def homo_warping_float(src_fea, src_proj, ref_proj, depth_values):
    """src_fea [B,C,H,W], src_proj/ref_proj [B,4,4], depth_values [B,Nd,H,W] -> [B,C,Nd,H,W]
    (reference models/module.py:527-568; [B,Nd] depth values are broadcast like the reference's view)."""
    B, C, H, W = src_fea.shape
    if depth_values.dim() == 2:
        depth_values = depth_values.reshape(B, -1, 1, 1).expand(B, depth_values.shape[1], H, W)
    rt = hip_ops.relative_transforms(torch.stack((ref_proj, src_proj), 1))[:, 0]
    return hip_ops.homo_warp(src_fea, rt, depth_values)


def depth_regression(p, depth_values):
    """sum_d p * depth (reference models/module.py:617-625)."""
    return hip_ops.depth_regression(p, depth_values)


def get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape, max_depth=192.0, min_depth=0.0):
    """reference models/module.py:628-643 (max_depth / min_depth are unused there too)."""
    return hip_ops.depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, list(shape))


def get_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, device, dtype, shape, max_depth=192.0, min_depth=0.0):
    """reference models/module.py:646-663."""
    return hip_ops.depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, list(shape))


# ---- FeatureNet0 building blocks (reference models/module.py:164-251, 506-524) -
class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        y = self.conv(x)
        if self.bn is not None:
            y = self.bn(y)
        return F.relu(y, inplace=True) if self.relu else y


class Deconv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        assert stride in (1, 2)
        self.stride = stride
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        y = self.conv(x)
        if self.stride == 2:
            y = y[:, :, :2 * x.shape[2], :2 * x.shape[3]].contiguous()
        if self.bn is not None:
            y = self.bn(y)
        return F.relu(y, inplace=True) if self.relu else y


class DeConv2dFuse(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1):
        super().__init__()
        self.deconv = Deconv2d(in_channels, out_channels, kernel_size, stride=2, padding=1, output_padding=1,
                               bn=True, relu=relu, bn_momentum=bn_momentum)
        self.conv = Conv2d(2 * out_channels, out_channels, kernel_size, stride=1, padding=1, bn=bn, relu=relu,
                           bn_momentum=bn_momentum)

    def forward(self, x_pre, x):
        return self.conv(torch.cat((self.deconv(x), x_pre), dim=1))


# ---- hot-path parameter containers -------------------------------------------
class _FusedLayer(nn.Module):
    def forward(self, *args, **kwargs):
        raise RuntimeError("%s has no standalone forward: it is fused into the HIP kernels of its parent network "
                           "(libadamvs_hip.so)" % type(self).__name__)


class ConvBnReLU(_FusedLayer):
    """reference models/module.py:254-261"""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)


class ConvReLU(_FusedLayer):
    """reference models/module.py:264-270"""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)


class ConvGRUCell(_FusedLayer):
    """reference models/module.py:5-52"""

    def __init__(self, input_channels, hidden_channels, kernel_size=3):
        super().__init__()
        self.input_channels, self.hidden_channels, self.kernel_size = input_channels, hidden_channels, kernel_size
        pad = (kernel_size - 1) // 2
        self.conv_gates = nn.Sequential(nn.Conv2d(input_channels + hidden_channels, 2 * hidden_channels, kernel_size,
                                                  stride=1, padding=pad, bias=True))
        self.convc = nn.Sequential(nn.Conv2d(input_channels + hidden_channels, hidden_channels, kernel_size,
                                             stride=1, padding=pad, bias=True))
