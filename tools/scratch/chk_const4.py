"""GPU box: the constant-channel fold of stages 4 / 5 (round 6) against the arm that computes every channel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from conftest import parity_set_of
from oracle import roomnet_ref as R

w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
rc = 0
for side in (224, 600):
    ims_all = parity_set_of(side)
    ww = dict(w)
    if side == 600:
        ww["dense/kernel"] = R.synth_dense_kernel_600()
    g = build_graph(6, side)
    for dt in ("bf16", "f16"):
        for nb in ((1, 8, 160) if side == 224 else (1, 5)):
            pick = (np.arange(nb) * 5) % len(ims_all)
            ims = ims_all[pick]
            fold = _capi.Engine(g, ww, device=0, dtype=dt, max_batch=nb)
            full = _capi.Engine(g, ww, device=0, dtype=dt, max_batch=nb, compute_frozen=True)
            ci = fold.const_info()
            ia, pa = fold.forward_u8(ims)
            ib, pb = full.forward_u8(ims)
            line = "%d %s nb %d const_info %s frozen %s" % (side, dt, nb, ci, fold.frozen_info())
            for name in ("s3.bn2", "s4.bn", "s5.bn2"):
                a, b = fold.tap(name, nb), full.tap(name, nb)
                ne = a != b
                per_c = ne.reshape(-1, a.shape[-1]).sum(0)
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                line += "\n    %-7s differing %d of %d (channels with differences: %d) max|d| %.3g absmax %.3g nan %d" % (
                    name, int(ne.sum()), a.size, int((per_c > 0).sum()), d.max(), np.abs(b).max(), int(np.isnan(a).sum()))
                if name != "s3.bn2":
                    const_c = [c for c in range(a.shape[-1]) if np.unique(a[..., c]).size == 1]
                    const_b = [c for c in range(a.shape[-1]) if np.unique(b[..., c]).size == 1]
                    exact = all((a[..., c] == b[..., c]).all() for c in const_b)
                    line += "\n            constant channels: fold arm %d, computing arm %d, equal on the computing arm's constants: %s" % (len(const_c), len(const_b), exact)
                    if not exact:
                        rc = 1
            dp = float(np.abs(pa - pb).max())
            line += "\n    max|dprob| %.3g ids equal %s" % (dp, bool((ia == ib).all()))
            if dp > 2e-3 or not (ia == ib).all():
                rc = 1
            print(line, flush=True)
            fold.close(); full.close()
print("RESULT", "FAIL" if rc else "ok")
sys.exit(rc)
