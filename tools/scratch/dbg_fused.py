"""GPU debug: where does the cross-stage fused kernel differ from the stage-launch path?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
weights = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
ims_all = parity_batch(224, seed=1)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pick = (np.arange(nb) * 7) % len(ims_all)
ims = ims_all[pick]
for dtype in ("bf16", "f16"):
    fused = _capi.Engine(build_graph(6, 224), weights, device=0, dtype=dtype, max_batch=nb)
    plain = _capi.Engine(build_graph(6, 224), weights, device=0, dtype=dtype, max_batch=nb, stage_launches=True)
    fused.forward_u8(ims); plain.forward_u8(ims)
    a, b = fused.tap("s3.bn2", nb), plain.tap("s3.bn2", nb)
    a2 = fused.tap("s3.bn2", nb)
    fused.forward_u8(ims)
    a3 = fused.tap("s3.bn2", nb)
    print(dtype, "fused run-to-run identical:", bool((a == a3).all()))
    bad = a != b
    print(dtype, "mismatches", int(bad.sum()), "of", bad.size, "max abs", float(np.abs(a - b).max()),
          "max rel", float((np.abs(a - b) / np.maximum(np.abs(b), 1e-6)).max()))
    print("  per image:", bad.reshape(nb, -1).sum(1).tolist())
    rows = bad.sum(axis=(0, 2, 3)); cols = bad.sum(axis=(0, 1, 3)); ch = bad.sum(axis=(0, 1, 2))
    print("  rows with mismatches:", np.nonzero(rows)[0][:40].tolist(), "...")
    print("  row hist (first 60):", rows[:60].tolist())
    print("  col hist by tile position (col % 29):", [int(cols[np.arange(len(cols)) % 29 == k].sum()) for k in range(29)])
    print("  col hist by tile:", [int(cols[29 * t:29 * t + 29].sum()) for t in range(8)])
    print("  channel hist:", ch.tolist())
    fused.close(); plain.close()

# which arm is closer to the oracle on the elements where they differ?
from oracle import c_oracle
sel = [2, 4]
ref = c_oracle.infer(weights, ims[sel], taps=True)
want = np.asarray(ref["taps"]["s3.bn2"])
for dtype in ("f16",):
    fused = _capi.Engine(build_graph(6, 224), weights, device=0, dtype=dtype, max_batch=nb)
    plain = _capi.Engine(build_graph(6, 224), weights, device=0, dtype=dtype, max_batch=nb, stage_launches=True)
    fused.forward_u8(ims); plain.forward_u8(ims)
    a, b = fused.tap("s3.bn2", nb)[sel], plain.tap("s3.bn2", nb)[sel]
    bad = a != b
    ea, eb = np.abs(a - want), np.abs(b - want)
    print("on %d differing elements: mean |fused-oracle| %.3e  mean |stagewise-oracle| %.3e; fused closer in %.1f%%" % (
        bad.sum(), ea[bad].mean(), eb[bad].mean(), 100.0 * (ea[bad] < eb[bad]).mean()))
    print("elsewhere: mean |err| %.3e" % ea[~bad].mean())
    # per-row error of each arm (all elements)
    ra = ea.mean(axis=(0, 2, 3)); rb = eb.mean(axis=(0, 2, 3))
    print("row mean err fused   :", " ".join("%.1e" % x for x in ra[24:48]))
    print("row mean err stagewise:", " ".join("%.1e" % x for x in rb[24:48]))
    fused.close(); plain.close()
