"""Evaluation helpers of the reference's test mode (reference utils.py:29-66, 236-315): nested-container converters,
the running mean of scalar dicts and the three depth metrics `train_whu.py --mode test` reports.  Plain PyTorch on
whatever device the maps live on -- bookkeeping around the hot path, not part of it.
"""
import numpy as np
import torch


def _recursive(func):
    """Apply `func` to the leaves of nested lists / tuples / dicts (reference utils.py:30-41)."""
    def walk(v):
        if isinstance(v, list):
            return [walk(x) for x in v]
        if isinstance(v, tuple):
            return tuple(walk(x) for x in v)
        if isinstance(v, dict):
            return {k: walk(x) for k, x in v.items()}
        return func(v)
    return walk


@_recursive
def tensor2float(v):
    if isinstance(v, float):
        return v
    if isinstance(v, torch.Tensor):
        return v.data.item()
    raise NotImplementedError("invalid input type {} for tensor2float".format(type(v)))


@_recursive
def tensor2numpy(v):
    if isinstance(v, np.ndarray):
        return v
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy().copy()
    raise NotImplementedError("invalid input type {} for tensor2numpy".format(type(v)))


@_recursive
def tocuda(v):
    if isinstance(v, torch.Tensor):
        return v.cuda()
    if isinstance(v, str):
        return v
    raise NotImplementedError("invalid input type {} for tocuda".format(type(v)))


class DictAverageMeter:
    """Mean of dicts of floats over update() calls (reference utils.py:236-255)."""

    def __init__(self):
        self.data, self.count = {}, 0

    def update(self, new_input):
        self.count += 1
        for k, v in new_input.items():
            if not isinstance(v, float):
                raise NotImplementedError("invalid data {}: {}".format(k, type(v)))
            self.data[k] = self.data.get(k, 0.0) + v if self.count > 1 else v

    def mean(self):
        return {k: v / self.count for k, v in self.data.items()}


def _per_image_mean(values):
    return torch.stack(values).mean()


@torch.no_grad()
def Thres_metrics(depth_est, depth_gt, mask, thres):
    """Fraction of valid pixels with |est - gt| < thres, per image, then the mean over the batch (utils.py:286-293)."""
    assert isinstance(thres, (int, float))
    out = []
    for e, g, m in zip(depth_est, depth_gt, mask):
        out.append(((e[m] - g[m]).abs() < thres).float().mean())
    return _per_image_mean(out)


@torch.no_grad()
def Inter_metrics(depth_est, depth_gt, interval, mask, thres):
    """Same with the error measured in hypothesis intervals (utils.py:296-304).  `interval` broadcasts against the
    masked pixels exactly as in the reference (a one-element tensor or a float)."""
    assert isinstance(thres, (int, float))
    out = []
    for e, g, m in zip(depth_est, depth_gt, mask):
        out.append((((e[m] - g[m]).abs() / interval) < thres).float().mean())
    return _per_image_mean(out)


@torch.no_grad()
def AbsDepthError_metrics(depth_est, depth_gt, mask, depth_threshold):
    """Mean absolute error over the valid pixels whose error is below depth_threshold (utils.py:307-315); an image
    without such a pixel gives nan, as torch.mean of an empty tensor does in the reference."""
    out = []
    for e, g, m in zip(depth_est, depth_gt, mask):
        diff = (e[m] - g[m]).abs()
        out.append(diff[diff < depth_threshold].mean())
    return _per_image_mean(out)
