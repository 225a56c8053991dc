#!/usr/bin/env python3
"""CPU model of the 16-bit path's rounding (round 6): the reference graph in torch float32 with the conv weights and the stage
outputs rounded where the HIP kernels round them, against the fp64 goldens of the 64-image parity set.  Attributes the bf16 logit
error to its sources and prices the two refinements of round 6 before they were built (NOTES.md, round 6):

  weights   "rne"      every weight rounded to nearest
            "carried"  the rounding residual carried from tap to tap of a (cin, cout) pair (rn_fused.hip: diffuse_taps)
  stores    "rne"      round to nearest even
            "rows3"    (bits + d(row mod 3)) >> 16 with d = 1/6, 3/6, 5/6 of an ulp (rn_stage.h: rn_dither_seed)
            "3x3"      a 3 x 3 ordered pattern over rows and columns (the model's best; not built)

  python tools/sim16.py            prints the table (a few minutes on 8 cores)
Development tool (it imports the oracle, like tools/make_golden.py); nothing in the product or the tests uses it."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import roomnet_ref as R                      # noqa: E402
from roomnet_amd.synth import parity_set                 # noqa: E402
from roomnet_amd.tf_bundle import BundleReader           # noqa: E402

W = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
FIELDS = np.load(os.path.join(ROOT, "tests", "golden", "class_fields.npz"))["fields_u8"]
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "parity_224.npz"))["logits_f64"]
ROWS3 = [[65536 * 1 // 6], [65536 * 3 // 6], [65536 * 5 // 6]]
PAT33 = [[(2 * ((i * 3 + j) * 4 % 9) + 1) * 65536 // 18 for j in range(3)] for i in range(3)]


def nm(base, i):
    return base if i == 0 else "%s_%d" % (base, i)


def rne(t, dt):
    return t if dt is None else t.to(dt).to(torch.float32)


def carried(kern, dt):
    """kern [cout, cin, 3, 3]: residual carried over the nine taps, centre first."""
    if dt is None:
        return kern
    k = kern.reshape(kern.shape[0], kern.shape[1], 9).clone()
    out, carry = torch.empty_like(k), torch.zeros_like(k[..., 0])
    for t in (4, 0, 8, 2, 6, 1, 7, 3, 5):
        v = k[..., t] + carry
        q = v.to(dt).to(torch.float32)
        out[..., t], carry = q, v - q
    return out.reshape(kern.shape)


def dithered(t, pat):
    """bf16 store of t [n, c, h, w] as (bits + pat[y mod ph][x mod pw]) >> 16."""
    n, c, h, w = t.shape
    p = torch.tensor(pat, dtype=torch.int64)
    d = p[torch.arange(h).view(-1, 1) % p.shape[0], torch.arange(w).view(1, -1) % p.shape[1]]
    bits = (t.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF) + d.view(1, 1, h, w)
    bits &= 0xFFFF0000
    return torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32).view(torch.float32)


def forward(x, wdt, adt, weights="rne", stores="rne", dither_stages=()):
    """x [n, 3, S, S] float32 -> logits; wdt / adt: torch dtype of the weights / stage outputs (None = float32)."""
    w = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in W.items()}
    st = {"conv": 0, "bn": 0}

    def bn(t):
        n = nm("batch_normalization", st["bn"])
        st["bn"] += 1
        inv = torch.rsqrt(w[n + "/moving_variance"] + 1e-3) * w[n + "/gamma"]
        sh = (1, -1, 1, 1)
        return (t - w[n + "/moving_mean"].view(sh)) * inv.view(sh) + w[n + "/beta"].view(sh)

    def resize(t, out):
        _, _, h, wd = t.shape
        ylo, yhi, yl = R.resize_tables(h, out)
        xlo, xhi, xl = R.resize_tables(wd, out)
        yl, xl = torch.from_numpy(yl).view(1, 1, -1, 1), torch.from_numpy(xl).view(1, 1, 1, -1)
        r0, r1 = t[:, :, torch.from_numpy(ylo)], t[:, :, torch.from_numpy(yhi)]
        a, b = torch.from_numpy(xlo), torch.from_numpy(xhi)
        top = r0[..., a] + (r0[..., b] - r0[..., a]) * xl
        bot = r1[..., a] + (r1[..., b] - r1[..., a]) * xl
        return top + (bot - top) * yl

    def store(t, ci):
        if adt is None:
            return t
        if stores != "rne" and ci in dither_stages and adt == torch.bfloat16:
            return dithered(t, ROWS3 if stores == "rows3" else PAT33)
        return rne(t, adt)

    def block(t, pooling=True, k=3, s=1, depth=1):
        first = None
        for d in range(depth):
            ci = st["conv"]
            st["conv"] += 1
            kern = w[nm("conv2d", ci) + "/kernel"].permute(3, 2, 0, 1).contiguous()
            if ci > 0:                                   # (stage 0's weights are exact in the kernels: fp16 hi + lo pairs)
                kern = carried(kern, wdt) if weights == "carried" else rne(kern, wdt)
            t = torch.clamp(F.conv2d(t, kern), 0.0, 6.0)
            if pooling:
                t = F.avg_pool2d(t, k, s)
            t = bn(t)
            if depth > 1 and d == depth - 1:
                t = bn(t + resize(first, t.shape[2]))
            t = store(t, ci)
            if d == 0:
                first = t
        return t

    t = block(x)
    t = block(t, k=4, s=1, depth=3)
    t = block(t, k=4, s=2, depth=2)
    t = block(t, pooling=False)
    t = block(t, k=4, s=2, depth=3)
    t = t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)
    for i in range(3):
        n = nm("batch_normalization", st["bn"])
        st["bn"] += 1
        inv = torch.rsqrt(w[n + "/moving_variance"] + 1e-3) * w[n + "/gamma"]
        t = torch.clamp(t @ w[nm("dense", i) + "/kernel"], 0.0, 6.0) * inv.view(1, -1) + (w[n + "/beta"] - w[n + "/moving_mean"] * inv).view(1, -1)
    return torch.clamp(t @ w["dense_3/kernel"] + w["dense_3/bias"], 0.0, 6.0)


def main():
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    x = torch.from_numpy(R.preprocess_batch(parity_set(224, FIELDS))).permute(0, 3, 1, 2).contiguous()
    B, H = torch.bfloat16, torch.float16

    def row(name, **kw):
        with torch.no_grad():
            e = np.abs(forward(x, **kw).numpy() - GOLD).max(1)
        print("%-64s max %.4f  mean %.4f  worst images %s" % (name, e.max(), e.mean(), np.argsort(e)[::-1][:3].tolist()), flush=True)

    row("float32 model", wdt=None, adt=None)
    row("bf16, plain rounding (rounds 1-5)", wdt=B, adt=B)
    row("bf16, weights only rounded", wdt=B, adt=None)
    row("bf16, stores only rounded", wdt=None, adt=B)
    row("bf16, carried weight rounding, weights only", wdt=B, adt=None, weights="carried")
    row("bf16, carried weights, plain stores", wdt=B, adt=B, weights="carried")
    for st_set in ((1, 3, 4, 5), (0, 1, 3, 4, 5), tuple(range(10))):
        row("bf16, carried weights, rows3 dither of stages %s" % (st_set,), wdt=B, adt=B, weights="carried", stores="rows3", dither_stages=st_set)
    row("bf16, carried weights, 3x3 dither of every stage", wdt=B, adt=B, weights="carried", stores="3x3", dither_stages=tuple(range(10)))
    row("bf16, plain weights, rows3 dither of stages (1, 3, 4, 5)", wdt=B, adt=B, stores="rows3", dither_stages=(1, 3, 4, 5))
    row("fp16, plain rounding", wdt=H, adt=H)
    row("fp16, carried weights", wdt=H, adt=H, weights="carried")


if __name__ == "__main__":
    main()
