"""Synthetic, seeded inputs for the Ada-MVS depth-inference path.

The reference ships no checkpoints and no test data, so parity
work, the bench and the golden fixtures all use the recipes below
(SURVEY.md section 8c).  Everything is drawn key-by-key from numpy PCG64
generators seeded from the parameter NAME, so the same weights are obtained
for any module that has the reference's state-dict keys -- the reference model
in the fixture generator, the oracle, and the HIP-backed drop-in class.

Nothing here touches a GPU or imports the oracle.
"""
import math
import zlib

import numpy as np
import torch

# BASELINE.json configs, as (views, H, W, ndepths, num_depth)
CONFIGS = {
    "tiny": dict(views=3, H=64, W=96, ndepths=[16, 8, 4], num_depth=16),
    "cfg1": dict(views=3, H=128, W=160, ndepths=[48, 32, 8], num_depth=48),
    "cfg2": dict(views=5, H=384, W=768, ndepths=[192], num_depth=192),
    "cfg3": dict(views=5, H=384, W=768, ndepths=[192, 64, 8], num_depth=192),
    "cfg5": dict(views=9, H=768, W=1536, ndepths=[256, 96, 16], num_depth=256),
}
DEPTH_INTERVALS_RATIO = [4.0, 2.0, 1.0]
DEPTH_RANGE = (400.0, 600.0)


def _rng(key, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(key.encode()), seed]))


def _is_transposed(module):
    return isinstance(module, torch.nn.ConvTranspose2d)


# Gains on the two logits layers (`reg.prob`: the scores of the stage-1 softmax, reference adamvs.py:481; `reg_fuse.upconv2d`:
# reg_cost, the argument of the UNSTABILISED exp of adamvs.py:516).
#   "default"  3 / 3: the recipe of SURVEY.md 8c -- soft probability volumes (pair confidence 0.4 - 1.0, |reg_cost| <= 13)
#   "sharp"    30 / 15: what a trained network's dynamic range looks like -- stage-1 softmaxes near one-hot (mean pair confidence
#              0.98), reg_cost up to +-60 (exp ~ 1e26; the running sums A = sum depth * exp reach 1e29 of fp32's 3e38)
#   "overflow" 30 / 75: reg_cost beyond 88.7 at part of the image: exp = inf, and the reference returns inf / NaN maps there
LOGIT_GAINS = {"default": (3.0, 3.0), "sharp": (30.0, 15.0), "overflow": (30.0, 75.0)}


def seeded_state_dict(model, seed=0, recipe="default"):
    """He-normal convs, randomised BN statistics, gain 3 on the two logits
    layers: gives non-degenerate probability volumes with random weights.
    recipe: LOGIT_GAINS above (the same draws, other gains on the two logits layers).

    `model` is any nn.Module carrying the reference's key names
    (reference models/adamvs.py:537-565).  Returns a new state dict.
    """
    sd = {}
    for mname, mod in model.named_modules():
        pre = mname + "." if mname else ""
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            w = mod.weight
            k = w.shape[2] * w.shape[3]
            if _is_transposed(mod):
                fan_in = w.shape[0] * k / float(mod.stride[0] * mod.stride[1])
            else:
                fan_in = w.shape[1] * k
            g = _rng(pre + "weight", seed)
            sd[pre + "weight"] = torch.from_numpy(
                (g.standard_normal(tuple(w.shape)) * math.sqrt(2.0 / fan_in)).astype(np.float32))
            if mod.bias is not None:
                g = _rng(pre + "bias", seed)
                sd[pre + "bias"] = torch.from_numpy(
                    (g.standard_normal(tuple(mod.bias.shape)) * 0.1).astype(np.float32))
        elif isinstance(mod, torch.nn.BatchNorm2d):
            n = mod.num_features
            sd[pre + "weight"] = torch.from_numpy(
                _rng(pre + "weight", seed).uniform(0.5, 1.5, n).astype(np.float32))
            sd[pre + "bias"] = torch.from_numpy(
                (_rng(pre + "bias", seed).standard_normal(n) * 0.1).astype(np.float32))
            sd[pre + "running_mean"] = torch.from_numpy(
                (_rng(pre + "running_mean", seed).standard_normal(n) * 0.1).astype(np.float32))
            sd[pre + "running_var"] = torch.from_numpy(
                _rng(pre + "running_var", seed).uniform(0.5, 1.5, n).astype(np.float32))
            sd[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        elif isinstance(mod, torch.nn.GroupNorm):
            n = mod.num_channels
            sd[pre + "weight"] = torch.from_numpy(
                _rng(pre + "weight", seed).uniform(0.5, 1.5, n).astype(np.float32))
            sd[pre + "bias"] = torch.from_numpy(
                (_rng(pre + "bias", seed).standard_normal(n) * 0.1).astype(np.float32))
    g_prob, g_up = LOGIT_GAINS[recipe]
    for key in list(sd):
        if key.endswith("reg.prob.weight"):
            sd[key] = sd[key] * g_prob
        elif key.endswith("upconv2d.weight"):
            sd[key] = sd[key] * g_up
    missing = set(model.state_dict().keys()) - set(sd.keys())
    if missing:
        raise RuntimeError("seeded_state_dict: keys without a recipe: %s" % sorted(missing)[:5])
    return sd


def rig_projections(views, H, W, batch=1, baseline=8.0, dtype=np.float32):
    """Synthetic pinhole rig -> {"stage1","stage2","stage3"}: [B,V,4,4].

    Each matrix is [K.[R|t]; 0 0 0 1] with rows 0-1 divided by 4 / 2 / 1, the
    way reference datasets/predict_oblique.py:154-177 builds them.
    """
    K = np.array([[1.2 * W, 0, W / 2.0], [0, 1.2 * W, H / 2.0], [0, 0, 1.0]], dtype=np.float64)
    full = np.zeros((batch, views, 4, 4), dtype=np.float64)
    for b in range(batch):
        for v in range(views):
            a = 0.02 * v * (1.0 + 0.05 * b)
            R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
            t = np.array([-baseline * v, 3.0 * (v % 2), 0.5 * v])
            P = np.eye(4)
            P[:3, :3] = K @ R
            P[:3, 3] = K @ t
            full[b, v] = P
    out = {}
    for name, s in (("stage1", 4.0), ("stage2", 2.0), ("stage3", 1.0)):
        m = full.copy()
        m[:, :, :2, :] = full[:, :, :2, :] / s
        out[name] = torch.from_numpy(m.astype(dtype))
    return out


def tile_inputs(cfg, batch=1, seed=0, baseline=8.0):
    """imgs [B,V,3,H,W] ~ N(0,1), proj_matrices dict, depth_values [B,2]."""
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(batch, c["views"], 3, c["H"], c["W"], generator=g, dtype=torch.float32)
    proj = rig_projections(c["views"], c["H"], c["W"], batch=batch, baseline=baseline)
    depth_values = torch.tensor([list(DEPTH_RANGE)] * batch, dtype=torch.float32)
    return imgs, proj, depth_values


def smooth_features(batch, C, h, w, seed=0):
    """Band-limited random feature maps [B,C,h,w] (op-level tests: a warp of
    white noise would make every sub-pixel error look large)."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(batch, C, max(h // 4, 2), max(w // 4, 2), generator=g)
    x = torch.nn.functional.interpolate(lo, size=(h, w), mode="bicubic", align_corners=False)
    return (x + 0.1 * torch.randn(batch, C, h, w, generator=g)).contiguous()
