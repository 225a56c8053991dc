"""Third, independent CPU restatement of the reference graph with torch-CPU library ops (MKL-DNN conv / pool),
the fast one of the three: test infrastructure and bench.py's `cpu_baseline` leg only -- never a product path.

Follows reference network.py:172-244 (conv_block / dense_block / init_nn_graph) and :44-45 (heads) with the
TensorFlow-1.13.1 op semantics listed in SURVEY.md 8a: conv3x3 VALID stride 1 no bias + ReLU6 (:184-186), avg-pool
VALID (:189), FusedBatchNorm inference (x - mean) * (rsqrt(var + 1e-3) * gamma) + beta (:193, :202), legacy bilinear
resize + add (:199), unfused BN of the dense blocks x * inv + (beta - mean * inv) (:217), ReLU6 on the logits (:214).
PARITY UNPINNED against TensorFlow itself (TF 1.13.1 cannot be installed here; the reference holds no vectors): it is
pinned to the NumPy and plain-C restatements by tests/test_oracle.py."""
from __future__ import annotations

import numpy as np

from oracle import roomnet_ref as R


def _nm(base, i):
    return base if i == 0 else "%s_%d" % (base, i)


def forward_tensor(w, x, raw_logits=False):
    """The graph on a torch tensor: x float [n, 3, S, S] (RGB, already scaled to [-1, 1]) and w = {name: tensor} -> logits [n, 6]
    (`raw_logits`: the last dense block's output BEFORE its ReLU6 -- what tools/make_golden.py's image search climbs, since the
    clamp has no gradient where it is active).  Differentiable: autograd runs through it when x requires grad."""
    import torch
    import torch.nn.functional as F
    state = {"conv": 0, "bn": 0}

    def bn(t):
        n = _nm("batch_normalization", state["bn"])
        state["bn"] += 1
        inv = torch.rsqrt(w[n + "/moving_variance"] + 1e-3) * w[n + "/gamma"]
        if t.dim() == 4:
            shape = (1, -1, 1, 1)
            return (t - w[n + "/moving_mean"].view(shape)) * inv.view(shape) + w[n + "/beta"].view(shape)
        return t * inv.view(1, -1) + (w[n + "/beta"] - w[n + "/moving_mean"] * inv).view(1, -1)

    def legacy_resize(t, out):
        _, _, h, wd = t.shape
        ylo, yhi, yl = R.resize_tables(h, out)
        xlo, xhi, xl = R.resize_tables(wd, out)
        yl = torch.from_numpy(yl).to(t.dtype).view(1, 1, -1, 1)
        xl = torch.from_numpy(xl).to(t.dtype).view(1, 1, 1, -1)
        rows0, rows1 = t[:, :, torch.from_numpy(ylo)], t[:, :, torch.from_numpy(yhi)]
        xlo_t, xhi_t = torch.from_numpy(xlo), torch.from_numpy(xhi)
        top = rows0[..., xlo_t] + (rows0[..., xhi_t] - rows0[..., xlo_t]) * xl
        bot = rows1[..., xlo_t] + (rows1[..., xhi_t] - rows1[..., xlo_t]) * xl
        return top + (bot - top) * yl

    def block(t, pooling=True, k=3, s=1, depth=1):
        first = None
        for d in range(depth):
            kern = w[_nm("conv2d", state["conv"]) + "/kernel"].permute(3, 2, 0, 1).contiguous()
            state["conv"] += 1
            t = torch.clamp(F.conv2d(t, kern), 0.0, 6.0)
            if pooling:
                t = F.avg_pool2d(t, k, s)
            t = bn(t)
            if d == 0:
                first = t
        if depth > 1:
            t = bn(t + legacy_resize(first, t.shape[2]))
        return t

    t = block(x)                                  # network.py:226
    t = block(t, k=4, s=1, depth=3)               # :227
    t = block(t, k=4, s=2, depth=2)               # :228
    t = block(t, pooling=False)                   # :229
    t = block(t, k=4, s=2, depth=3)               # :230
    t = t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)     # NHWC flatten, :231-233
    for i in range(3):
        t = bn(torch.clamp(t @ w[_nm("dense", i) + "/kernel"], 0.0, 6.0))
    raw = t @ w["dense_3/kernel"] + w["dense_3/bias"]
    return raw if raw_logits else torch.clamp(raw, 0.0, 6.0)


def infer(weights, ims_u8, threads=None):
    """uint8 BGR [n, S, S, 3] -> dict(logits, probs, ids) (float32 / int64 numpy arrays)."""
    import torch
    if threads:
        torch.set_num_threads(int(threads))
    x = torch.from_numpy(R.preprocess_batch(np.asarray(ims_u8))).permute(0, 3, 1, 2).contiguous()
    w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    with torch.no_grad():
        logits = forward_tensor(w, x)
        probs = torch.softmax(logits, dim=-1)
    return {"logits": logits.numpy(), "probs": probs.numpy(), "ids": probs.argmax(-1).numpy().astype(np.int64)}
