"""ORACLE -- test infrastructure only (imported by tests/ alone; the product never calls it).

Scalar, loop-per-pixel restatement of the caller-side image operations the reference performs before the forward pass
(reference network.py:137-146 ``center_crop``; network.py:152 ``cv2.resize(im, (im_side, im_side))``, default
INTER_LINEAR, uint8).  The arithmetic lives in a third-party dependency that is absent here -- OpenCV (``cv2``; the
reference pins no version, README "opencv-python") -- so this follows the published algorithm of imgproc/resize.cpp:
half-pixel-centre source coordinates in float32 from a double scale, 11-bit fixed-point coefficients (cvRound = round
half to even, saturated to short), 32-bit horizontal pass, vertical pass
(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
PARITY UNPINNED against a real OpenCV build (it cannot be installed; the reference holds no image fixtures): it pins
the two product implementations -- roomnet_amd/imageops.py (NumPy) and csrc/rn_imageops.hip (HIP) -- to each other and
to this independent third restatement, not to cv2 itself.  Pure Python loops: small cases only."""
import numpy as np


def center_crop(x):
    """reference network.py:137-146."""
    h, w, _ = x.shape
    offset = abs((w - h) // 2)
    if h < w:
        return x[:, offset:offset + h, :]
    if w < h:
        return x[offset:offset + w, :, :]
    return x


def resize_linear_u8_scalar(src, dw, dh):
    """cv2.resize(src, (dw, dh)), INTER_LINEAR, uint8 HWC (reference network.py:152); the exact-2x special case is not
    handled here (callers test it separately against a 2x2 box average)."""
    sh, sw, cn = src.shape
    sx_scale, sy_scale = 1.0 / (dw / sw), 1.0 / (dh / sh)
    out = np.zeros((dh, dw, cn), np.uint8)

    def coef(d, scale, ssize, clamp):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if clamp:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= ssize - 1:
                f, s = np.float32(0), ssize - 1
        c0 = int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048)))
        c1 = int(np.rint(f * np.float32(2048)))
        return s, c0, c1

    for dy in range(dh):
        sy, b0, b1 = coef(dy, sy_scale, sh, False)
        y0, y1 = min(max(sy, 0), sh - 1), min(max(sy + 1, 0), sh - 1)
        for dx in range(dw):
            sx, a0, a1 = coef(dx, sx_scale, sw, True)
            sx1 = min(sx + 1, sw - 1)
            for c in range(cn):
                r0 = int(src[y0, sx, c]) * a0 + int(src[y0, sx1, c]) * a1
                r1 = int(src[y1, sx, c]) * a0 + int(src[y1, sx1, c]) * a1
                out[dy, dx, c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
    return out
