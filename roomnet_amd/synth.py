"""Seeded synthetic image batches (there is no dataset and no network access).

``perf_batch``   uniform-noise uint8 images: what ``bench.py`` times.
``parity_batch`` a structured mix (solid colours, gradients, low-pass noise,
                 checkers/stripes, noise).  Uniform noise alone is degenerate for
                 parity work: every such image lands in the same class with about
                 the same logits, so the parity set adds images that reach other
                 classes and both clamps of the final ReLU6.
All images are BGR uint8 HWC, the layout ``cv2.imread`` hands the reference
(``infer.py:81``).
"""
from __future__ import annotations

import numpy as np


def perf_batch(n: int, side: int = 224, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, side, side, 3), dtype=np.uint8)


def _upsample_bilinear(coarse: np.ndarray, side: int) -> np.ndarray:
    """coarse [3,g,g] in [0,1] -> [side,side,3] smooth field (separable lerp)."""
    g = coarse.shape[1]
    pos = np.linspace(0.0, g - 1.0, side)
    lo = np.minimum(pos.astype(np.int64), g - 2)
    t = pos - lo
    rows = coarse[:, lo, :] * (1 - t)[None, :, None] + coarse[:, lo + 1, :] * t[None, :, None]
    out = rows[:, :, lo] * (1 - t)[None, None, :] + rows[:, :, lo + 1] * t[None, None, :]
    return np.transpose(out, (1, 2, 0))


def field_image(grid_u8: np.ndarray, side: int) -> np.ndarray:
    """A low-frequency colour field: coarse uint8 grid [3 (B, G, R), g, g] -> [side, side, 3] uint8, the same separable
    up-sampling as the seeded low-pass images below.  The class-covering parity images (tests/golden/class_fields.npz, found by
    tools/search_class_images.py) are stored as such grids and regenerated at 224 and 600."""
    field = _upsample_bilinear(np.asarray(grid_u8, np.float64), side)          # values in [0, 255]
    return np.ascontiguousarray(np.clip(field, 0, 255).astype(np.uint8))


def parity_set(side: int, fields_u8=None, seed: int = 1) -> np.ndarray:
    """The parity images of a side: the seeded batch below followed by one image per class-covering grid."""
    ims = parity_batch(side, seed)
    if fields_u8 is None or len(fields_u8) == 0:
        return ims
    return np.ascontiguousarray(np.concatenate([ims, np.stack([field_image(f, side) for f in fields_u8])], 0))


def parity_batch(side: int = 224, seed: int = 1) -> np.ndarray:
    rng = np.random.default_rng(seed)
    ims = []
    # (i) solid colours
    for col in ((0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255),
                (128, 128, 128), (30, 200, 120), (220, 40, 180)):
        ims.append(np.broadcast_to(np.array(col, np.uint8), (side, side, 3)).copy())
    # (ii) linear gradients
    ramp = np.linspace(0, 255, side)
    ims.append(np.stack([np.tile(ramp[None, :], (side, 1))] * 3, -1).astype(np.uint8))
    ims.append(np.stack([np.tile(ramp[:, None], (1, side))] * 3, -1).astype(np.uint8))
    diag = (ramp[:, None] + ramp[None, :]) / 2
    ims.append(np.stack([diag, 255 - diag, diag[::-1]], -1).astype(np.uint8))
    # (iii) low-pass noise at several scales
    for sc in (4, 8, 16, 32, 56):
        for _ in range(3):
            g = side // sc + 2
            field = _upsample_bilinear(rng.random((3, g, g)), side)
            ims.append(np.clip(field * 255.0, 0, 255).astype(np.uint8))
    # (iv) checkers / stripes
    yy, xx = np.mgrid[0:side, 0:side]
    for period in (2, 8, 32):
        chk = (((yy // period) + (xx // period)) & 1).astype(np.uint8) * 255
        ims.append(np.stack([chk, chk, chk], -1))
        stripe = ((xx // period) & 1).astype(np.uint8) * 255
        ims.append(np.stack([stripe, 255 - stripe, stripe], -1))
    # (v) uniform noise
    for _ in range(8):
        ims.append(rng.integers(0, 256, (side, side, 3), dtype=np.uint8))
    return np.ascontiguousarray(np.stack(ims, 0))
