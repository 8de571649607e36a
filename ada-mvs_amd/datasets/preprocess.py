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


def _cv_round(x):
    """cvRound / saturate_cast<int>(double): round half to even."""
    return int(np.rint(x))


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


# ---- OpenCV's 8-bit INTER_LINEAR, restated bit for bit (modules/imgproc/src/resize.cpp, generic C++ path) ----------
# cv::resize computes, per axis, float sample positions and 11-bit fixed-point weights (INTER_RESIZE_COEF_BITS = 11,
# weights saturate_cast<short>(w * 2048): round half to even), filters horizontally into int32 rows
# (HResizeLinear<uchar,int,short>: S[sx] * a0 + S[sx+1] * a1) and vertically with
# VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>:  dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
# Horizontally a tap outside the image gives weight (2048, 0) on the clamped pixel; vertically the row index is clamped
# and the two weights are kept.  This follows the published algorithm of OpenCV 4.x; cv2 is not installed in the
# build container, so it is NOT pinned against a cv2 run (IPP / vendor HAL builds of OpenCV may also differ in the
# last bit).  tests/test_predict_io.py holds it to hand-derived vectors.
_COEF_BITS = 11
_COEF_SCALE = np.float32(1 << _COEF_BITS)


def _cv_linear_coeffs(n_in, n_out, scale, clamp_weights):
    """-> (index of the first tap, index of the second tap, int weights [n_out, 2]) of one axis; scale = 1 / f."""
    pos = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)       # fx = (float)((dx+0.5)*scale_x - 0.5)
    i0 = np.floor(pos).astype(np.int64)                                                        # cvFloor
    f = (pos - i0.astype(np.float32)).astype(np.float32)                                       # fx -= sx, in float
    if clamp_weights:                                    # x axis: sx < 0 -> fx = 0, sx = 0; sx >= w - 1 -> fx = 0, sx = w - 1
        f = np.where((i0 < 0) | (i0 >= n_in - 1), np.float32(0), f).astype(np.float32)
    w1 = np.rint(f * _COEF_SCALE).astype(np.int64)                                            # saturate_cast<short>(fx * 2048)
    w0 = np.rint((np.float32(1) - f) * _COEF_SCALE).astype(np.int64)
    lo = np.clip(i0, 0, n_in - 1)
    hi = np.clip(i0 + 1, 0, n_in - 1)
    return lo, hi, np.stack((w0, w1), 1)


def _cv_resize_linear_u8(image, nh, nw, scale_y, scale_x):
    h, w = image.shape[:2]
    x0, x1, ax = _cv_linear_coeffs(w, nw, scale_x, True)
    y0, y1, by = _cv_linear_coeffs(h, nh, scale_y, False)
    src = image.astype(np.int64)
    bc = (slice(None), slice(None)) + (None,) * (image.ndim - 2)
    rows = src[:, x0] * ax[:, 0][None, :][bc] + src[:, x1] * ax[:, 1][None, :][bc]             # int32 rows, values * 2048
    s0, s1 = rows[y0] >> 4, rows[y1] >> 4
    b0, b1 = by[:, 0][:, None][bc], by[:, 1][:, None][bc]
    out = (((b0 * s0) >> 16) + ((b1 * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _cv_resize_area2_u8(image, nh, nw):
    """cv::resize turns INTER_LINEAR into INTER_AREA when both scale factors are exactly 1/2 (resize.cpp: `interpolation
    == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2`): full 2 x 2 blocks are (a + b + c + d + 2) >> 2
    (ResizeAreaFastVec), a block cut by the image border (odd sizes whose half rounds up) is the rounded mean of the
    pixels it still covers (ResizeAreaFast_Invoker: saturate_cast<uchar>((float)sum / count))."""
    h, w = image.shape[:2]
    src = image.astype(np.int64)
    pad = np.zeros((2 * nh, 2 * nw) + image.shape[2:], dtype=np.int64)
    cnt = np.zeros((2 * nh, 2 * nw) + (1,) * (image.ndim - 2), dtype=np.int64)
    hh, ww = min(h, 2 * nh), min(w, 2 * nw)
    pad[:hh, :ww] = src[:hh, :ww]
    cnt[:hh, :ww] = 1
    total = pad[0::2, 0::2] + pad[0::2, 1::2] + pad[1::2, 0::2] + pad[1::2, 1::2]
    n = cnt[0::2, 0::2] + cnt[0::2, 1::2] + cnt[1::2, 0::2] + cnt[1::2, 1::2]
    full = (total + 2) >> 2
    part = np.rint(total.astype(np.float32) / np.maximum(n, 1).astype(np.float32)).astype(np.int64)      # cvRound: half to even
    return np.clip(np.where(n == 4, full, part), 0, 255).astype(np.uint8)


def scale_image(image, scale=1, interpolation="linear"):
    """cv2.resize(image, None, fx=scale, fy=scale, ...) (reference preprocess.py:44-54): output size
    cvRound(size * scale) (ties to even), half-pixel centres.  'linear' is INTER_LINEAR -- for 8-bit images OpenCV's
    fixed-point arithmetic bit for bit (see above), for other types bilinear in float64 rounded to nearest --,
    'biculic' is nearest neighbour exactly as in the reference.  Like the reference, any scale (1 included) goes
    through the resize."""
    h, w = image.shape[:2]
    nh, nw = _cv_round(h * scale), _cv_round(w * scale)
    if interpolation == "biculic":
        ys = np.minimum(np.floor(np.arange(nh) * (1.0 / scale)).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(nw) * (1.0 / scale)).astype(np.int64), w - 1)
        return image[ys][:, xs]
    if interpolation != "linear":
        return None
    if scale == 1:
        return image                        # identity under both arithmetic paths (weights 2048 / 0)
    inv = 1.0 / scale                       # scale_x = 1. / inv_scale_x in cv::resize
    if image.dtype == np.uint8:
        if inv == 2.0:
            return _cv_resize_area2_u8(image, nh, nw)
        return _cv_resize_linear_u8(image, nh, nw, inv, inv)
    y0, y1, fy = _axis_taps(h, nh, inv)
    x0, x1, fx = _axis_taps(w, nw, inv)
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
