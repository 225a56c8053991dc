"""Image file I/O for the directory driver, standing in for the three OpenCV calls the
reference makes (``cv2.imread`` ``infer.py:81``, ``cv2.putText`` ``infer.py:89-92``,
``cv2.imwrite`` ``infer.py:93``).  OpenCV is not installed on the MI355X hosts; Pillow is.

The overlay text is drawn with the same Hershey simplex strokes, anchor, scale rule and colours as
``cv2.putText`` (``roomnet_amd/hershey.py``); the anti-aliasing filter is not OpenCV's, so overlay pixel values are
close to, not equal to, OpenCV's -- pixels that never reach the network.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


_pa = None


def _pixels_zero_copy(im):
    """The RGB image's own memory as an (H, W, 4) uint8 view (Pillow stores RGB as RGBX), or None.  ``np.asarray(im)`` goes
    through ``im.tobytes()``, a Python loop that holds the GIL for ~1-2 ms per 1920 x 1080 image: with it the decode pool of the
    directory drivers levels off near 500 img/s however many threads it has (tools/bench_decode.py).  Pillow >= 11.2 exports the
    image memory through the Arrow C data interface without a copy; the channel swap that follows releases the GIL."""
    global _pa
    if _pa is False or not hasattr(im, "__arrow_c_array__"):
        return None
    try:
        if _pa is None:
            import pyarrow
            _pa = pyarrow
        w, h = im.size
        v = _pa.array(im).flatten().to_numpy(zero_copy_only=True)
        return v.reshape(h, w, 4) if v.size == h * w * 4 else None
    except ImportError:
        _pa = False
        return None
    except Exception:        # an image held in several memory blocks is not exportable: the plain path takes it
        return None


def imread(path: str) -> Optional[np.ndarray]:
    """BGR uint8 HWC like ``cv2.imread(path)`` (IMREAD_COLOR: 3 channels, alpha dropped,
    EXIF orientation applied); ``None`` when the file is not a readable image."""
    try:
        from PIL import Image, ImageOps
        with Image.open(path) as im:
            im = ImageOps.exif_transpose(im)
            if im.mode in ("I;16", "I;16B", "I;16L", "I"):
                # 16-bit grey: cv2.imread(IMREAD_COLOR) scales to 8 bits by >> 8 (Pillow's convert() would clip)
                a = np.asarray(im)
                a8 = (a.astype(np.int64).clip(0, 65535) >> 8).astype(np.uint8)    # libpng strip_16: the high byte
                return np.ascontiguousarray(np.repeat(a8[:, :, None], 3, axis=2))
            if im.mode in ("RGBA", "LA", "P"):
                im = im.convert("RGBA").convert("RGB") if im.mode != "P" else im.convert("RGB")
            elif im.mode != "RGB":
                im = im.convert("RGB")
            rgbx = _pixels_zero_copy(im)
            if rgbx is not None:
                return np.ascontiguousarray(rgbx[:, :, 2::-1])       # copied while `im` still owns the memory
            rgb = np.asarray(im, dtype=np.uint8)
    except Exception:
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def imwrite(path: str, im_bgr: np.ndarray) -> bool:
    """``cv2.imwrite``: format from the extension (JPEG quality 95 like OpenCV's default)."""
    from PIL import Image
    ext = os.path.splitext(path)[1].lower()
    img = Image.fromarray(np.ascontiguousarray(im_bgr[:, :, ::-1]))
    try:
        if ext in (".jpg", ".jpeg", ".jpe"):
            img.save(path, format="JPEG", quality=95)
        elif ext == ".png":
            img.save(path, format="PNG", compress_level=1)
        elif ext == ".bmp":
            img.save(path, format="BMP")
        else:
            img.save(path)
    except Exception:
        return False
    return True


def put_text(im_bgr: np.ndarray, text: str, org: Tuple[int, int], font_scale: float,
             color_bgr: Tuple[int, int, int]) -> None:
    """In-place overlay like ``cv2.putText(im, text, org, FONT_HERSHEY_SIMPLEX, font_scale,
    color, 1, LINE_AA)``: ``org`` is the bottom-left corner of the text on its baseline.  Drawn with the Hershey
    simplex strokes OpenCV uses (``roomnet_amd/hershey.py``)."""
    from . import hershey
    hershey.put_text(im_bgr, text, org, font_scale, color_bgr, 1)
