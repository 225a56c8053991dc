"""GPU box: float32 matrix-core path vs float32 per-node path at several input sides (stage outputs and probabilities)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader

w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
rng = np.random.default_rng(0)
for side in [int(x) for x in sys.argv[1:]] or [200, 224, 256, 300, 420, 512, 600]:
    g = build_graph(6, side)
    ww = dict(w)
    ww["dense/kernel"] = (np.random.default_rng(side).standard_normal((g.flat_len, 32)) * (0.5 / np.sqrt(g.flat_len))).astype(np.float32)
    nb = 3
    ims = rng.integers(0, 256, (nb, side, side, 3), dtype=np.uint8)
    mm = _capi.Engine(g, ww, device=0, dtype="f32", max_batch=nb)
    pn = _capi.Engine(g, ww, device=0, dtype="f32", max_batch=nb, taps=True)
    ids_m, pr_m = mm.forward_u8(ims)
    ids_p, pr_p = pn.forward_u8(ims)
    worst = 0.0
    for nm in ("s1.bn", "s3.bn2", "s4.bn", "s5.bn2", "s6.bn", "s9.bn2"):
        a, b = mm.tap(nm, nb), pn.tap(nm, nb)
        worst = max(worst, float(np.abs(a - b).max() / np.abs(b).max()))
    print(side, "max stage |d| / absmax %.2e  max |dprob| %.2e  ids equal %s" % (worst, np.abs(pr_m - pr_p).max(), (ids_m == ids_p).all()))
    mm.close(); pn.close()
