"""GPU box: many passes over the same 160 images on one handle -- every pass must give the bits of the first (both arms, both 16-bit dtypes)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from conftest import parity_set_of
w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
ims_all = parity_set_of(224)
ims = ims_all[(np.arange(160) * 7) % len(ims_all)]
g = build_graph(6, 224)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for dt in ("bf16", "f16"):
    for cf in (False, True):
        e = _capi.Engine(g, w, device=0, dtype=dt, max_batch=160, compute_frozen=cf)
        ids0, p0 = e.forward_u8(ims)
        t0 = {n: e.tap(n, 160) for n in ("s1.bn", "s3.bn2", "s4.bn", "s5.bn2")}
        nb = 0
        for rep in range(N):
            ids, p = e.forward_u8(ims)
            if not (np.array_equal(p, p0) and np.array_equal(ids, ids0)):
                nb += 1
                print("   ", dt, cf, "pass", rep, "probs differ by", float(np.abs(p - p0).max()))
            if rep % 25 == 24:
                for n, t in t0.items():
                    a = e.tap(n, 160)
                    if not np.array_equal(a, t):
                        nb += 1
                        d = np.abs(a - t)
                        print("   ", dt, cf, "pass", rep, n, "differs in", int((a != t).sum()), "elements, max", float(d.max()), "at", np.unravel_index(int(d.argmax()), d.shape))
        print(dt, "computing arm" if cf else "default", N, "passes:", "ok" if nb == 0 else "%d MISMATCHES" % nb, flush=True)
        bad += nb
        e.close()
sys.exit(1 if bad else 0)
