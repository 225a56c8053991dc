"""GPU debug: dump stage outputs of the loaded library (ROOMNET_HIP_LIB) for later comparison: tools/dbg_dump.py out.npz [names...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
weights = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
names = sys.argv[2:] or ["s4.bn", "s5.bn2", "s9.bn2"]
out = {}
for dtype in ("bf16", "f16"):
    for nb in (3, 40):
        ims = parity_batch(224, seed=1)[:nb]
        e = _capi.Engine(build_graph(6, 224), weights, device=0, dtype=dtype, max_batch=nb)
        ids, probs = e.forward_u8(ims)
        out["%s_%d_probs" % (dtype, nb)] = probs
        for nm in names:
            out["%s_%d_%s" % (dtype, nb, nm)] = e.tap(nm, nb)
        e.close()
np.savez(sys.argv[1], **out)
print("dumped", len(out), "arrays to", sys.argv[1])
