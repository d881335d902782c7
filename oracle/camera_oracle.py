"""TEST INFRASTRUCTURE ONLY - CPU restatement of the camera stage around the network.

The reference does this work with OpenCV on the CPU (src/unet_ros_node.py:296-311, src/unet.py:24-72):
    bgr8 sensor_msgs/Image -> cv2.warpPerspective(M, (1055, 685)) -> cv2.resize(same size, INTER_AREA)  [a copy]
    -> BGR2RGB -> cv2.resize((224, 224)) [bilinear] -> network -> threshold -> cv2.resize(back) [bilinear] -> mono8.
OpenCV (cv2) is a third-party dependency that is NOT installed here and the reference holds no fixture for this
stage, so this file restates OpenCV 4.x's published 8-bit algorithms (modules/imgproc/src/imgwarp.cpp,
resize.cpp) and the parity of the stage against cv2 itself is UNPINNED; the HIP kernels are checked bit-exactly
against this restatement.

Algorithms restated (all integer after the coordinate computation):
  * getPerspectiveTransform: the 8x8 linear system of cv::getPerspectiveTransform, solved in float64.
  * warpPerspective, INTER_LINEAR, BORDER_CONSTANT(0), 8-bit: for destination (x, y): with M^-1 (float64),
    W = INTER_TAB_SIZE / (m20 x + m21 y + m22) (0 if the denominator is 0), X = cvRound((m00 x + m01 y + m02) W),
    Y likewise (INTER_TAB_SIZE = 32); integer source pixel (X >> 5, Y >> 5), fractions a = X & 31, b = Y & 31;
    weights (32-b)(32-a)*32, (32-b)a*32, b(32-a)*32, b*a*32 (they sum to 2^15 exactly, so the table's sum
    correction never fires); value = (sum w*pixel + 2^14) >> 15, taps outside the source read 0.
  * resize, INTER_LINEAR, 8-bit: fx = float32((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx, clamped at the
    borders (sx < 0 -> sx = 0, fx = 0; sx >= w-1 -> sx = w-1, fx = 0); coefficients saturate_cast<short>(c * 2048)
    (cvRound); horizontal pass D = S[sx]*a0 + S[sx+1]*a1 (int32); vertical pass
    (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2  (the VResizeLinear<uchar,int,short> form).
    Equal sizes are a plain copy (cv::resize's early exit), which is also what the INTER_AREA call at scale 1 is.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform(src, dst) -> 3x3 float64 (reference call: src/unet_ros_node.py:255)."""
    src = np.asarray(src, dtype=np.float64).reshape(4, 2)
    dst = np.asarray(dst, dtype=np.float64).reshape(4, 2)
    a = np.zeros((8, 8), dtype=np.float64)
    b = np.zeros(8, dtype=np.float64)
    for i in range(4):
        x, y = src[i]
        u, v = dst[i]
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    h = np.linalg.solve(a, b)
    return np.append(h, 1.0).reshape(3, 3)


def _round_half_even(v):
    return np.rint(v)   # cvRound: lrint in the default rounding mode


def warp_coords(m, width, height, xs=None, ys=None):
    """Fixed-point source coordinates (X, Y) of destination pixels, as cv::warpPerspective computes them."""
    inv = np.linalg.inv(np.asarray(m, dtype=np.float64))
    if xs is None:
        ys, xs = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    xs = np.asarray(xs, dtype=np.float64)
    ys = np.asarray(ys, dtype=np.float64)
    w = inv[2, 0] * xs + inv[2, 1] * ys + inv[2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(w != 0.0, INTER_TAB_SIZE / w, 0.0)
    lim = 2147483647.0
    fx = np.clip((inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]) * w, -2147483648.0, lim)
    fy = np.clip((inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]) * w, -2147483648.0, lim)
    return _round_half_even(fx).astype(np.int64), _round_half_even(fy).astype(np.int64)


def _sample(img, X, Y):
    """8-bit bilinear sample at fixed-point coordinates, constant border 0; img (H,W,C) uint8."""
    h, w = img.shape[:2]
    sx, sy = X >> INTER_BITS, Y >> INTER_BITS
    a, b = (X & (INTER_TAB_SIZE - 1)).astype(np.int64), (Y & (INTER_TAB_SIZE - 1)).astype(np.int64)
    acc = np.zeros(X.shape + (img.shape[2],), dtype=np.int64)
    for dy, dx, wgt in ((0, 0, (32 - b) * (32 - a)), (0, 1, (32 - b) * a), (1, 0, b * (32 - a)), (1, 1, b * a)):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        pix = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        acc += np.where(ok[..., None], pix, 0) * (wgt * 32)[..., None]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def warp_perspective(img, m, width, height):
    """cv2.warpPerspective(img, m, (width, height)) for 8-bit (H,W,C) images (defaults: INTER_LINEAR, constant 0)."""
    img = np.asarray(img)
    if img.ndim == 2:
        return warp_perspective(img[..., None], m, width, height)[..., 0]
    X, Y = warp_coords(m, width, height)
    return _sample(img, X, Y)


def resize_coeffs(src, dst):
    """(source index, [c0, c1] as int16 2^11 fixed point) per destination index, cv::resize INTER_LINEAR."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    low, high = s < 0, s >= src - 1
    f = np.where(low | high, np.float32(0), f)
    s = np.where(low, 0, np.where(high, src - 1, s))
    c1 = np.rint(f.astype(np.float32) * np.float32(2048)).astype(np.int64)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    return s, np.stack([c0, c1], axis=1)


def resize_linear(img, width, height):
    """cv2.resize(img, (width, height)) for 8-bit images (default interpolation: bilinear)."""
    img = np.asarray(img)
    if img.ndim == 2:
        return resize_linear(img[..., None], width, height)[..., 0]
    h, w = img.shape[:2]
    if (h, w) == (height, width):
        return img.copy()
    sx, cx = resize_coeffs(w, width)
    sy, cy = resize_coeffs(h, height)
    src = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, w - 1)
    sy1 = np.minimum(sy + 1, h - 1)
    # horizontal pass on the two source rows of every destination row
    def hrow(rows):
        r = src[rows]                                     # (height, w, C)
        return r[:, sx] * cx[None, :, 0, None] + r[:, sx1] * cx[None, :, 1, None]
    d0, d1 = hrow(sy), hrow(sy1)
    b0, b1 = cy[:, 0][:, None, None], cy[:, 1][:, None, None]
    out = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def prestage(bgr, m, warp_w, warp_h, out_w=224, out_h=224, bgr_in=True):
    """Image callback up to the network input (src/unet_ros_node.py:299-311 + src/unet.py:33): (H,W,3) uint8 ->
    (out_h, out_w, 3) RGB uint8."""
    warped = warp_perspective(bgr, m, warp_w, warp_h)
    rgb = warped[..., ::-1] if bgr_in else warped
    return resize_linear(np.ascontiguousarray(rgb), out_w, out_h)


def poststage(mask, out_w, out_h):
    """postprocess_output's resize back to the warped size (src/unet.py:70): (h,w) uint8 -> (out_h,out_w) uint8."""
    return resize_linear(mask, out_w, out_h)
