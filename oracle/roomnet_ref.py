"""ORACLE (test infrastructure, not product code) -- NumPy restatement of the
reference's forward pass.

PARITY UNPINNED: the arithmetic of the reference lives in TensorFlow 1.13.1
(un-vendored pip dependency; version recorded in final_model/roomnet.meta),
which cannot be installed here, and the reference ships no tests or golden
vectors.  This file restates the published op semantics of the TF-1.13 CPU
kernels that ``network.py`` calls, op by op and unfused, so that every graph
node can be tapped.  It is pinned only structurally (checkpoint CRCs, the
``_output_shapes`` recorded in roomnet.meta) and against two independent
restatements (oracle/tf_ops.c and a torch-CPU cross-check in tests/).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module; the product package never does.

Reference call sites restated here
  network.py:128-135  infer            -> preprocess_batch
  network.py:137-146  center_crop      -> center_crop
  network.py:148-156  infer_optimized  -> preprocess_batch (224 inputs skip the resize)
  network.py:172-208  conv_block       -> _conv_block
  network.py:210-223  dense_block      -> _dense_block
  network.py:225-237  init_nn_graph    -> forward
  network.py:44-45    softmax/argmax   -> forward (tail)
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon


# --------------------------------------------------------------- host pre-proc
def preprocess_batch(im_bgr_u8: np.ndarray) -> np.ndarray:
    """network.py:129 / :153 -- ``((im[..., [2,1,0]] / 255.) * 2) - 1`` evaluated
    in float64 by NumPy, then cast to float32 when fed to the placeholder."""
    im = ((im_bgr_u8[..., [2, 1, 0]] / 255.) * 2) - 1
    return im.astype(np.float32)


def center_crop(x: np.ndarray) -> np.ndarray:
    """network.py:137-146."""
    h, w, _ = x.shape
    offset = abs((w - h) // 2)
    if h < w:
        return x[:, offset:offset + h, :]
    if w < h:
        return x[offset:offset + w, :, :]
    return x.copy()


# ------------------------------------------------------------------ TF-1.13 ops
def conv2d_valid(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """tf.layers.conv2d(strides=1, padding=VALID, use_bias=False): NHWC x HWIO
    cross-correlation; out[n,y,x,o] = sum_{ky,kx,c} in[n,y+ky,x+kx,c] W[ky,kx,c,o]."""
    n, h, wd, c = x.shape
    kh, kw, _, o = w.shape
    ho, wo = h - kh + 1, wd - kw + 1
    wm = w.reshape(kh * kw * c, o)
    out = np.empty((n, ho, wo, o), dtype=x.dtype)
    for i in range(n):
        cols = np.empty((ho, wo, kh * kw * c), dtype=x.dtype)
        for ky in range(kh):
            for kx in range(kw):
                t = ky * kw + kx
                cols[:, :, t * c:(t + 1) * c] = x[i, ky:ky + ho, kx:kx + wo, :]
        out[i] = (cols.reshape(ho * wo, -1) @ wm).reshape(ho, wo, o)
    return out


def relu6(x: np.ndarray) -> np.ndarray:
    return np.minimum(np.maximum(x, x.dtype.type(0)), x.dtype.type(6))


def avg_pool_valid(x: np.ndarray, k: int, s: int) -> np.ndarray:
    """tf.nn.avg_pool VALID: window sum in raster order, divided by k*k."""
    n, h, w, c = x.shape
    ho, wo = (h - k) // s + 1, (w - k) // s + 1
    acc = np.zeros((n, ho, wo, c), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            acc += x[:, ky:ky + (ho - 1) * s + 1:s, kx:kx + (wo - 1) * s + 1:s, :]
    return acc / x.dtype.type(k * k)


def fused_batch_norm_infer(x, gamma, beta, mean, var, eps=BN_EPS):
    """FusedBatchNorm(is_training=False), CPU kernel:
    y = (x - mean) * (rsqrt(var + eps) * gamma) + beta."""
    t = x.dtype.type
    inv = (t(1) / np.sqrt(var.astype(x.dtype) + t(eps))) * gamma.astype(x.dtype)
    return (x - mean.astype(x.dtype)) * inv + beta.astype(x.dtype)


def batch_norm_2d(x, gamma, beta, mean, var, eps=BN_EPS):
    """tf.nn.batch_normalization (the unfused path tf.layers uses for rank-2
    inputs): inv = rsqrt(var+eps)*gamma;  y = x*inv + (beta - mean*inv)."""
    t = x.dtype.type
    inv = (t(1) / np.sqrt(var.astype(x.dtype) + t(eps))) * gamma.astype(x.dtype)
    return x * inv + (beta.astype(x.dtype) - mean.astype(x.dtype) * inv)


def resize_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """TF-1.13 ``compute_interpolation_weights`` with align_corners=False and no
    half-pixel centres, in float32 exactly as the kernel computes it:
    scale = in/float(out); src = i*scale; lo = int(src); hi = min(lo+1, in-1)."""
    scale = np.float32(in_size) / np.float32(out_size)
    src = (np.arange(out_size, dtype=np.float32) * scale).astype(np.float32)
    lo = src.astype(np.int64)
    hi = np.minimum(lo + 1, in_size - 1)
    lerp = (src - lo.astype(np.float32)).astype(np.float32)
    return lo, hi, lerp


def resize_bilinear_legacy(x: np.ndarray, out_side: int) -> np.ndarray:
    """tf.image.resize_bilinear(align_corners=False), TF 1.13 CPU kernel:
    top = tl + (tr-tl)*xl; bottom = bl + (br-bl)*xl; out = top + (bottom-top)*yl."""
    n, h, w, c = x.shape
    ylo, yhi, yl = resize_tables(h, out_side)
    xlo, xhi, xl = resize_tables(w, out_side)
    yl = yl.astype(x.dtype)[None, :, None, None]
    xl = xl.astype(x.dtype)[None, None, :, None]
    tl = x[:, ylo][:, :, xlo]
    tr = x[:, ylo][:, :, xhi]
    bl = x[:, yhi][:, :, xlo]
    br = x[:, yhi][:, :, xhi]
    top = tl + (tr - tl) * xl
    bottom = bl + (br - bl) * xl
    return top + (bottom - top) * yl


def softmax(x: np.ndarray) -> np.ndarray:
    shifted = x - x.max(axis=-1, keepdims=True)
    e = np.exp(shifted)
    return e / e.sum(axis=-1, keepdims=True)


# --------------------------------------------------------------------- graph
def _names(base: str):
    i = 0
    while True:
        yield base if i == 0 else "%s_%d" % (base, i)
        i += 1


class _Ctx:
    def __init__(self, weights, dtype, taps):
        self.w = weights
        self.dtype = dtype
        self.taps: Optional[Dict[str, np.ndarray]] = {} if taps else None
        self.conv_names = _names("conv2d")
        self.bn_names = _names("batch_normalization")
        self.dense_names = _names("dense")
        self.stage = 0
        self.dense_idx = 0

    def tap(self, name, val):
        if self.taps is not None:
            self.taps[name] = val

    def bn_params(self, name):
        return (self.w[name + "/gamma"], self.w[name + "/beta"],
                self.w[name + "/moving_mean"], self.w[name + "/moving_variance"])


def _conv_block(ctx: _Ctx, x, output_filters, pooling=True, pool_ksize=3, pool_stride=1,
                block_depth=1):
    """network.py:172-208 with the defaults the reference uses everywhere:
    kernel 3, stride 1, VALID, no bias, ReLU6, avg-pool, BN after the pool."""
    make_residual = block_depth > 1
    out = x
    residual_input = None
    for depth in range(block_depth):
        s = ctx.stage
        k = ctx.w[next(ctx.conv_names) + "/kernel"].astype(ctx.dtype)
        out = relu6(conv2d_valid(out, k))
        ctx.tap("s%d.conv" % s, out)
        if pooling:
            out = avg_pool_valid(out, pool_ksize, pool_stride)
            ctx.tap("s%d.pool" % s, out)
        out = fused_batch_norm_infer(out, *ctx.bn_params(next(ctx.bn_names)))
        ctx.tap("s%d.bn" % s, out)
        if depth == 0:
            residual_input = out
        ctx.stage += 1
    if make_residual:
        s = ctx.stage - 1
        out = out + resize_bilinear_legacy(residual_input, out.shape[1])
        ctx.tap("s%d.add" % s, out)
        out = fused_batch_norm_infer(out, *ctx.bn_params(next(ctx.bn_names)))
        ctx.tap("s%d.bn2" % s, out)
    return out


def _dense_block(ctx: _Ctx, x, batch_norm=True, biased=False):
    """network.py:210-223."""
    d = ctx.dense_idx
    name = next(ctx.dense_names)
    out = x @ ctx.w[name + "/kernel"].astype(ctx.dtype)
    if biased:
        out = out + ctx.w[name + "/bias"].astype(ctx.dtype)
    ctx.tap("d%d.mm" % d, out)
    out = relu6(out)
    ctx.tap("d%d.relu" % d, out)
    if batch_norm:
        out = batch_norm_2d(out, *ctx.bn_params(ctx.bn_name_for_dense()))
        ctx.tap("d%d.bn" % d, out)
    ctx.dense_idx += 1
    return out


def forward(weights: Dict[str, np.ndarray], x_rgb: np.ndarray, dtype=np.float32,
            taps: bool = False):
    """Full forward pass on pre-processed input ``x_rgb`` [N,S,S,3] (RGB, [-1,1]).

    dtype=float32 restates the reference's arithmetic type; dtype=float64 is the
    high-precision "truth" used to set tolerances.  Returns a dict with
    ``logits`` [N,C], ``probs`` [N,C] (cast to float32), ``ids`` [N] int64 and,
    when ``taps`` is set, ``taps`` name -> array for every graph node.
    """
    ctx = _Ctx(weights, dtype, taps)
    ctx.bn_name_for_dense = lambda: next(ctx.bn_names)
    x = np.asarray(x_rgb).astype(dtype)
    ctx.tap("input", x)
    out = _conv_block(ctx, x, 8)
    out = _conv_block(ctx, out, 32, pool_ksize=4, pool_stride=1, block_depth=3)
    out = _conv_block(ctx, out, 64, pool_ksize=4, pool_stride=2, block_depth=2)
    out = _conv_block(ctx, out, 128, pooling=False)
    out = _conv_block(ctx, out, 16, pool_ksize=4, pool_stride=2, block_depth=3)
    flat = out.reshape(out.shape[0], -1)
    ctx.tap("flat", flat)
    out = _dense_block(ctx, flat)
    out = _dense_block(ctx, out)
    out = _dense_block(ctx, out)
    logits = _dense_block(ctx, out, batch_norm=False, biased=True)
    probs = softmax(logits)
    ids = np.argmax(probs, axis=-1).astype(np.int64)
    ctx.tap("softmax", probs)
    res = {"logits": logits, "probs": probs.astype(np.float32), "ids": ids}
    if taps:
        res["taps"] = ctx.taps
    return res


def infer(weights, im_bgr_u8_batch: np.ndarray, dtype=np.float32, taps: bool = False):
    """network.py:128-135 ``RoomNet.infer`` in optimized mode: returns the dict of
    ``forward`` for a [N,S,S,3] BGR uint8 batch."""
    return forward(weights, preprocess_batch(im_bgr_u8_batch), dtype=dtype, taps=taps)


def node_names(block_spec=None) -> List[str]:
    """Tap names in execution order for the 224/600 graph."""
    names = ["input"]
    depths = [(1, True), (3, True), (2, True), (1, False), (3, True)]
    s = 0
    for depth, pooling in depths:
        for _ in range(depth):
            names.append("s%d.conv" % s)
            if pooling:
                names.append("s%d.pool" % s)
            names.append("s%d.bn" % s)
            s += 1
        if depth > 1:
            names += ["s%d.add" % (s - 1), "s%d.bn2" % (s - 1)]
    names.append("flat")
    for d in range(4):
        names += ["d%d.mm" % d, "d%d.relu" % d]
        if d < 3:
            names.append("d%d.bn" % d)
    names.append("softmax")
    return names


def synth_dense_kernel_600(flat_len: int = 3136) -> np.ndarray:
    """the shipped dense/kernel only fits im_side=224; the 600x600
    stress variant uses a seeded synthetic first dense kernel."""
    return np.random.default_rng(600).uniform(-0.04, 0.04, (flat_len, 32)).astype(np.float32)
