#!/usr/bin/env python3
"""Find synthetic inputs that the reference checkpoint puts into EACH of its six classes (infer.py:22 CLASS_LABELS) with a
comfortable top-2 margin, for the parity fixtures (the seeded parity set only ever reached classes 0, 1, 2).

    python tools/search_class_images.py [--per-class 5] [--steps 400] [--grid 14] [--seed 7]
        -> tests/golden/class_fields.npz      (then: python tools/make_golden.py)

An image of the set is a low-frequency colour field: a coarse uint8 grid `field[3, g, g]`, bilinearly up-sampled to side x side
(roomnet_amd.synth.field_image -- the same up-sampling the seeded low-pass images of parity_batch use).  Only the grids are
stored (3 g^2 bytes per image); the images are regenerated from them at 224 and at 600.

Search (build container only; CPU; test infrastructure -- nothing of the product path runs here): gradient ascent on the grid
through the torch restatement of the graph (oracle/torch_ref.forward_tensor, pre-ReLU6 logits, straight-through rounding of the
uint8 quantisation), objective = logit of the wanted class minus the largest other logit, capped so that the winner stays below
the 6-clamp.  Every candidate is then re-evaluated from its QUANTISED grid by the fp64 NumPy restatement
(oracle/roomnet_ref.py); kept when the fp64 argmax is the wanted class and the fp64 top-2 margin exceeds --min-margin.
Seeded and deterministic for a given torch build; the committed grids are the record."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import roomnet_ref as R, torch_ref  # noqa: E402
from roomnet_amd import tf_bundle  # noqa: E402
from roomnet_amd.synth import field_image  # noqa: E402


def upsample_matrix(g, side):
    """[side, g] matrix of the separable lerp of synth._upsample_bilinear (so that torch and numpy agree)."""
    pos = np.linspace(0.0, g - 1.0, side)
    lo = np.minimum(pos.astype(np.int64), g - 2)
    t = pos - lo
    m = np.zeros((side, g))
    m[np.arange(side), lo] = 1 - t
    m[np.arange(side), lo + 1] += t
    return m


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-class", type=int, default=5)
    ap.add_argument("--tries", type=int, default=8, help="candidates per class and round")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--grid", type=int, default=14)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--min-margin", type=float, default=0.6)
    ap.add_argument("--side", type=int, default=224)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "class_fields.npz"))
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    weights = tf_bundle.BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
    w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    g, side, ncls = args.grid, args.side, 6
    U = torch.from_numpy(upsample_matrix(g, side)).float()
    kept = {c: [] for c in range(ncls)}
    rnd = 0
    while any(len(kept[c]) < args.per_class for c in range(ncls)) and rnd < 6:
        want = [c for c in range(ncls) if len(kept[c]) < args.per_class]
        target = torch.tensor([c for c in want for _ in range(args.tries)])
        n = len(target)
        theta = (torch.randn(n, 3, g, g) * (0.5 + 0.5 * rnd)).requires_grad_(True)
        opt = torch.optim.Adam([theta], lr=0.08)
        onehot = torch.nn.functional.one_hot(target, ncls).bool()
        for it in range(args.steps):
            opt.zero_grad()
            field = torch.sigmoid(theta) * 255.0
            fq = field + (torch.round(field) - field).detach()                      # the stored grid is uint8
            img = torch.einsum("sg,ncgh,th->ncst", U, fq, U)                       # BGR planes, [0, 255]
            iq = img + (torch.floor(img.clamp(0, 255)) - img).detach()              # .astype(uint8) of field_image
            x = ((iq / 255.0) * 2.0 - 1.0).flip(1)                                  # BGR -> RGB (network.py:129)
            z = torch_ref.forward_tensor(w, x, raw_logits=True)
            zt = z[onehot]
            zo = z.masked_fill(onehot, -1e9).max(1).values
            margin = zt.clamp(max=5.0) - zo.clamp(min=0.0)                          # the real logits are ReLU6'd
            loss = -(margin.clamp(max=3.0)).sum() + 0.05 * (zt - 4.0).clamp(min=0).pow(2).sum()
            loss.backward()
            opt.step()
            if it % 50 == 0 or it == args.steps - 1:
                print("round %d it %3d  margins by class: %s" % (rnd, it, " ".join(
                    "%d:%.2f" % (c, float(margin[target == c].max())) for c in want)), flush=True)
        grids = np.rint(torch.sigmoid(theta).detach().numpy() * 255.0).clip(0, 255).astype(np.uint8)
        ims = np.stack([field_image(f, side) for f in grids])
        for i0 in range(0, n, 8):
            r = R.infer(weights, ims[i0:i0 + 8], np.float64)
            srt = np.sort(r["logits"], axis=1)
            for j in range(len(r["ids"])):
                c = int(target[i0 + j])
                m = float(srt[j, -1] - srt[j, -2])
                ok = int(r["ids"][j]) == c and m > args.min_margin and len(kept[c]) < args.per_class
                print("  class %d candidate: fp64 id %d margin %.3f top %.3f %s" % (c, r["ids"][j], m, srt[j, -1], "KEPT" if ok else ""), flush=True)
                if ok:
                    kept[c].append((grids[i0 + j], m))
        rnd += 1
    fields = np.stack([f for c in range(ncls) for f, _ in kept[c]])
    wanted = np.array([c for c in range(ncls) for _ in kept[c]], np.int64)
    np.savez_compressed(args.out, note=np.array(
        "coarse uint8 BGR colour grids [k, 3, g, g]; image = roomnet_amd.synth.field_image(grid, side); found by "
        "tools/search_class_images.py (gradient ascent through oracle/torch_ref.py, accepted by the fp64 restatement); "
        "self-generated, TF parity unpinned"), fields_u8=fields, wanted_ids=wanted)
    print("kept per class:", {c: len(kept[c]) for c in range(ncls)}, "->", args.out)


if __name__ == "__main__":
    main()
