"""Host image operations the hot path's callers need, restated without OpenCV
(``cv2`` is not installed on the MI355X hosts).

``resize_linear_u8`` restates ``cv2.resize(im, (W, H))`` with the default
``INTER_LINEAR`` for uint8 images (reference ``network.py:152``), following the
published OpenCV algorithm (imgproc/resize.cpp): half-pixel-centre source
coordinates, 11-bit fixed-point coefficients, horizontal pass into 32-bit rows,
vertical pass ``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``, and the
special case that an exact 2x2 down-scale is computed as a 2x2 box average.
Parity with a particular OpenCV build (IPP / HAL variants) is unpinned: OpenCV
cannot be installed here.
"""
from __future__ import annotations

import numpy as np

_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def _linear_coeffs(dsize: int, ssize: int, clamp_frac: bool):
    inv_scale = float(dsize) / float(ssize)
    scale = 1.0 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo = s < 0
        f[lo] = 0.0
        s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0.0
        s[hi] = ssize - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(_COEF_SCALE)).astype(np.int64)
    c1 = np.rint(f * np.float32(_COEF_SCALE)).astype(np.int64)
    c0 = np.clip(c0, -32768, 32767)
    c1 = np.clip(c1, -32768, 32767)
    return s, c0, c1, scale


def resize_linear_u8(im: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """``cv2.resize(im, (dst_w, dst_h))`` (INTER_LINEAR) for a uint8 HWC / HW image."""
    src = np.asarray(im)
    if src.dtype != np.uint8:
        raise TypeError("resize_linear_u8 expects uint8, got %s" % src.dtype)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    sh, sw, cn = src.shape
    if sh == dst_h and sw == dst_w:
        out = src.copy()
        return out[:, :, 0] if squeeze else out
    sx, a0, a1, scale_x = _linear_coeffs(dst_w, sw, clamp_frac=True)
    sy, b0, b1, scale_y = _linear_coeffs(dst_h, sh, clamp_frac=False)
    eps = np.finfo(np.float64).eps
    if (abs(scale_x - 2.0) < eps and abs(scale_y - 2.0) < eps):
        # INTER_LINEAR with an exact 2x2 down-scale is routed to the fast INTER_AREA kernel
        s = src.astype(np.int32)
        out = (s[0:2 * dst_h:2, 0:2 * dst_w:2] + s[0:2 * dst_h:2, 1:2 * dst_w:2] +
               s[1:2 * dst_h:2, 0:2 * dst_w:2] + s[1:2 * dst_h:2, 1:2 * dst_w:2] + 2) >> 2
        out = out.astype(np.uint8)
        return out[:, :, 0] if squeeze else out
    s = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    # horizontal pass for every source row that is needed
    rows = s[:, sx, :] * a0[None, :, None] + s[:, sx1, :] * a1[None, :, None]   # [sh, dst_w, cn]
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    r0 = rows[y0]
    r1 = rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out
