"""ORACLE (test infrastructure, not product code) -- ctypes driver that wires the
plain-C op restatements of oracle/tf_ops.c into the graph of the reference's
``network.py:225-237`` (conv_block :172-208, dense_block :210-223, heads :44-45).

PARITY UNPINNED (see oracle/tf_ops.c).  This is the *second* independent
restatement (oracle/roomnet_ref.py is the first); it is also what ``bench.py``
times as ``cpu_baseline`` (kind "port") because it is multi-threaded plain C.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtf_ops_oracle.so")
BN_EPS = 1e-3

_lib: Optional[ctypes.CDLL] = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "tf_ops.c")
    if force or not os.path.isfile(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libtf_ops_oracle.so"])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.rn_ref_max_threads.restype = ctypes.c_int
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n: int) -> None:
    lib().rn_ref_set_threads(ctypes.c_int(n))


def max_threads() -> int:
    return int(lib().rn_ref_max_threads())


def preprocess(im_bgr_u8: np.ndarray) -> np.ndarray:
    im = np.ascontiguousarray(im_bgr_u8, dtype=np.uint8)
    out = np.empty(im.shape, np.float32)
    lib().rn_ref_preprocess(_p(im), _p(out), ctypes.c_int64(im.size // 3))
    return out


class _Run:
    def __init__(self, weights: Dict[str, np.ndarray], taps: bool):
        self.w = {k: _f32(v) for k, v in weights.items()}
        self.L = lib()
        self.taps: Optional[Dict[str, np.ndarray]] = {} if taps else None
        self.n_conv = self.n_bn = self.n_dense = 0
        self.stage = 0
        self.didx = 0

    @staticmethod
    def _nm(base, i):
        return base if i == 0 else "%s_%d" % (base, i)

    def tap(self, name, val):
        if self.taps is not None:
            self.taps[name] = val.copy()

    def next_bn(self):
        nm = self._nm("batch_normalization", self.n_bn)
        self.n_bn += 1
        return (self.w[nm + "/gamma"], self.w[nm + "/beta"], self.w[nm + "/moving_mean"],
                self.w[nm + "/moving_variance"])

    # --- ops
    def conv(self, x):
        k = self.w[self._nm("conv2d", self.n_conv) + "/kernel"]
        self.n_conv += 1
        n, h, w, c = x.shape
        kh, kw, _, o = k.shape
        out = np.empty((n, h - kh + 1, w - kw + 1, o), np.float32)
        self.L.rn_ref_conv2d_valid(_p(x), _p(k), _p(out), n, h, w, c, o, kh, kw)
        return out

    def relu6(self, x):
        self.L.rn_ref_relu6(_p(x), ctypes.c_int64(x.size))
        return x

    def pool(self, x, k, s):
        n, h, w, c = x.shape
        out = np.empty((n, (h - k) // s + 1, (w - k) // s + 1, c), np.float32)
        self.L.rn_ref_avg_pool_valid(_p(x), _p(out), n, h, w, c, k, s)
        return out

    def bn4(self, x):
        g, b, m, v = self.next_bn()
        out = np.empty_like(x)
        self.L.rn_ref_fused_bn_infer(_p(x), _p(out), ctypes.c_int64(x.size // x.shape[-1]),
                                     x.shape[-1], _p(g), _p(b), _p(m), _p(v), ctypes.c_float(BN_EPS))
        return out

    def resize(self, x, side):
        n, h, w, c = x.shape
        out = np.empty((n, side, side, c), np.float32)
        self.L.rn_ref_resize_bilinear(_p(x), _p(out), n, h, w, c, side, side)
        return out

    def add(self, a, b):
        out = np.empty_like(a)
        self.L.rn_ref_add(_p(a), _p(b), _p(out), ctypes.c_int64(a.size))
        return out

    # --- blocks
    def conv_block(self, x, output_filters, pooling=True, pool_ksize=3, pool_stride=1, block_depth=1):
        out = x
        residual_input = None
        for depth in range(block_depth):
            s = self.stage
            out = self.relu6(self.conv(out))
            assert out.shape[-1] == output_filters
            self.tap("s%d.conv" % s, out)
            if pooling:
                out = self.pool(out, pool_ksize, pool_stride)
                self.tap("s%d.pool" % s, out)
            out = self.bn4(out)
            self.tap("s%d.bn" % s, out)
            if depth == 0:
                residual_input = out
            self.stage += 1
        if block_depth > 1:
            s = self.stage - 1
            out = self.add(out, self.resize(residual_input, out.shape[1]))
            self.tap("s%d.add" % s, out)
            out = self.bn4(out)
            self.tap("s%d.bn2" % s, out)
        return out

    def dense_block(self, x, batch_norm=True, biased=False):
        d = self.didx
        nm = self._nm("dense", self.n_dense)
        self.n_dense += 1
        k = self.w[nm + "/kernel"]
        m = x.shape[0]
        out = np.empty((m, k.shape[1]), np.float32)
        self.L.rn_ref_matmul(_p(x), _p(k), _p(out), m, k.shape[0], k.shape[1])
        if biased:
            self.L.rn_ref_bias_add(_p(out), _p(self.w[nm + "/bias"]), m, k.shape[1])
        self.tap("d%d.mm" % d, out)
        out = self.relu6(out)
        self.tap("d%d.relu" % d, out)
        if batch_norm:
            g, b, mu, v = self.next_bn()
            o2 = np.empty_like(out)
            self.L.rn_ref_bn_2d(_p(out), _p(o2), m, out.shape[1], _p(g), _p(b), _p(mu), _p(v),
                                ctypes.c_float(BN_EPS))
            out = o2
            self.tap("d%d.bn" % d, out)
        self.didx += 1
        return out


def forward(weights: Dict[str, np.ndarray], x_rgb: np.ndarray, taps: bool = False):
    r = _Run(weights, taps)
    x = _f32(x_rgb)
    r.tap("input", x)
    out = r.conv_block(x, 8)
    out = r.conv_block(out, 32, pool_ksize=4, pool_stride=1, block_depth=3)
    out = r.conv_block(out, 64, pool_ksize=4, pool_stride=2, block_depth=2)
    out = r.conv_block(out, 128, pooling=False)
    out = r.conv_block(out, 16, pool_ksize=4, pool_stride=2, block_depth=3)
    flat = np.ascontiguousarray(out.reshape(out.shape[0], -1))
    r.tap("flat", flat)
    out = r.dense_block(flat)
    out = r.dense_block(out)
    out = r.dense_block(out)
    logits = r.dense_block(out, batch_norm=False, biased=True)
    m, n = logits.shape
    probs = np.empty_like(logits)
    r.L.rn_ref_softmax(_p(logits), _p(probs), m, n)
    ids = np.empty((m,), np.int64)
    r.L.rn_ref_argmax(_p(probs), _p(ids), m, n)
    r.tap("softmax", probs)
    res = {"logits": logits, "probs": probs, "ids": ids}
    if taps:
        res["taps"] = r.taps
    return res


def infer(weights, im_bgr_u8_batch: np.ndarray, taps: bool = False):
    """network.py:128-135 (optimized mode) on a [N,S,S,3] BGR uint8 batch."""
    return forward(weights, preprocess(im_bgr_u8_batch), taps=taps)
