"""Resize / crop / normalise of the predict pipeline (reference datasets/preprocess.py:22-99), without cv2.

A camera is the reference's [2,4,4] array: cam[0] = extrinsic [Rcw|tcw], cam[1][:3,:3] = K,
cam[1][3] = (depth_min, interval, depth_num, depth_max).
"""
import math

import numpy as np


def scale_camera(cam, scale=1):
    """Focal lengths and principal point times `scale`; a copy (reference preprocess.py:22-34)."""
    out = np.copy(cam)
    for i, j in ((0, 0), (1, 1), (0, 2), (1, 2)):
        out[1][i][j] = cam[1][i][j] * scale
    return out


def scale_mvs_camera(cams, scale=1):
    """In place over a list of cameras (reference preprocess.py:37-41)."""
    for v in range(len(cams)):
        cams[v] = scale_camera(cams[v], scale=scale)
    return cams


def _axis_taps(n_in, n_out, step):
    """Bilinear taps of cv2.resize(INTER_LINEAR) called with fx/fy: sample centre (i + 0.5) * step - 0.5 with
    step = 1 / fx (not n_in / n_out), clamped at the border."""
    x = (np.arange(n_out, dtype=np.float64) + 0.5) * step - 0.5
    i0 = np.floor(x).astype(np.int64)
    f = x - i0
    below, above = i0 < 0, i0 >= n_in - 1
    f[below | above] = 0.0
    i0 = np.clip(i0, 0, n_in - 1)
    return i0, np.minimum(i0 + 1, n_in - 1), f


def scale_image(image, scale=1, interpolation="linear"):
    """cv2.resize(image, None, fx=scale, fy=scale, ...) (reference preprocess.py:44-54): output size
    round(size * scale) (ties to even, as cvRound), half-pixel centres.  'linear' is bilinear, 'biculic'
    is nearest neighbour exactly as in the reference.  8-bit results are rounded to nearest; cv2's 11-bit
    fixed-point weights can differ from this by one grey level on ties."""
    if scale == 1:
        return image
    h, w = image.shape[:2]
    nh, nw = int(np.round(h * scale)), int(np.round(w * scale))
    if interpolation == "biculic":
        ys = np.minimum(np.floor(np.arange(nh) * (1.0 / scale)).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(nw) * (1.0 / scale)).astype(np.int64), w - 1)
        return image[ys][:, xs]
    if interpolation != "linear":
        return None
    if scale == 0.5 and h % 2 == 0 and w % 2 == 0 and image.dtype.kind == "u" and image.dtype.itemsize <= 2:
        # the default predict setting: every sample centre falls between four pixels with weights 1/4 -- the 2 x 2 box,
        # in unsigned integer arithmetic (same rounding as the general path below: half up); signed images take that path
        acc = image[0::2, 0::2].astype(np.uint32)
        acc += image[0::2, 1::2]
        acc += image[1::2, 0::2]
        acc += image[1::2, 1::2]
        return ((acc + 2) >> 2).astype(image.dtype)
    y0, y1, fy = _axis_taps(h, nh, 1.0 / scale)
    x0, x1, fx = _axis_taps(w, nw, 1.0 / scale)
    src = image.astype(np.float64)
    bc = (slice(None), slice(None)) + (None,) * (image.ndim - 2)
    rows = src[y0] * (1.0 - fy)[:, None][bc] + src[y1] * fy[:, None][bc]
    out = rows[:, x0] * (1.0 - fx)[None, :][bc] + rows[:, x1] * fx[None, :][bc]
    if np.issubdtype(image.dtype, np.integer):
        info = np.iinfo(image.dtype)
        return np.clip(np.floor(out + 0.5), info.min, info.max).astype(image.dtype)
    return out.astype(image.dtype)


def scale_input(image, cam, depth_image=None, scale=1):
    """Image (and depth map) and camera resized together (reference preprocess.py:57-66)."""
    image = scale_image(image, scale=scale)
    cam = scale_camera(cam, scale=scale)
    if depth_image is None:
        return image, cam
    return image, cam, scale_image(depth_image, scale=scale, interpolation="linear")


def crop_input(image, cam, depth_image=None, max_h=384, max_w=768, resize_scale=1, base_image_size=32):
    """Top-left crop to at most (max_h, max_w) * resize_scale; a side below the limit is rounded UP to a multiple of
    `base_image_size` (the slice then simply ends at the image border, as in the reference).  The principal point
    moves by the crop origin, which is (0, 0).  `cam` is modified in place (reference preprocess.py:69-99)."""
    limit_h, limit_w = int(max_h * resize_scale), int(max_w * resize_scale)
    h, w = image.shape[:2]
    new_h = limit_h if h > limit_h else int(math.ceil(h / base_image_size) * base_image_size)
    new_w = limit_w if w > limit_w else int(math.ceil(w / base_image_size) * base_image_size)
    top, left = 0, 0
    image = image[top:top + new_h, left:left + new_w]
    cam[1][0][2] = cam[1][0][2] - left
    cam[1][1][2] = cam[1][1][2] - top
    if depth_image is None:
        return image, cam
    return image, cam, depth_image[top:top + new_h, left:left + new_w]


def center_image(img):
    """Per-channel zero mean / unit variance over the image, float32 (reference preprocess.py:102-112)."""
    x = np.array(img).astype(np.float32)
    var = np.var(x, axis=(0, 1), keepdims=True)
    mean = np.mean(x, axis=(0, 1), keepdims=True)
    return (x - mean) / (np.sqrt(var) + 0.00000001)
