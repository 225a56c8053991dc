"""GPU box: 600 x 600 bf16 / f16 logit and stage errors against the fp64 goldens (the library given by ROOMNET_HIP_LIB or the product one)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from conftest import parity_set_of, GOLDEN
from oracle import roomnet_ref as R, c_oracle
w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
for side in (224, 600):
    g = np.load(os.path.join(GOLDEN, "parity_%d.npz" % side))
    ww = dict(w)
    if side == 600:
        ww["dense/kernel"] = R.synth_dense_kernel_600()
    ims = parity_set_of(side)
    if "image_indices" in g.files:
        ims = ims[g["image_indices"]]
    ref = c_oracle.infer(ww, ims[1:2], taps=True) if side == 600 else None
    for dt in ("bf16", "f16"):
        for nd in (False, True):
            e = _capi.Engine(build_graph(6, side), ww, device=0, dtype=dt, max_batch=len(ims), no_dither=nd)
            ids, probs = e.forward_u8(ims)
            lg = e.tap("d3.relu", len(ims))
            err = np.abs(lg - g["logits_f64"]).max(1)
            line = "%d %s %s: max |dlogit| %.4f mean-of-max %.4f worst images %s" % (side, dt, "no_dither" if nd else "default", err.max(), err.mean(), np.argsort(err)[::-1][:3].tolist())
            if ref is not None:
                e.forward_u8(ims[1:2])
                rel = {}
                for n in ("s5.bn2", "s6.bn", "s7.bn", "s8.bn", "s9.bn2"):
                    got, want = e.tap(n, 1), np.asarray(ref["taps"][n])
                    rel[n] = round(float(np.abs(got - want).max() / np.abs(want).max()), 4)
                line += " stage rel " + str(rel)
            print(line, flush=True)
            e.close()
