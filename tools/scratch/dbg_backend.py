"""GPU box: where does the one-launch back end (130 images) leave the banded launches (chunks of 8)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader

w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
base = parity_batch(224, seed=1)
nb = 130
ims = base[np.arange(nb) % len(base)]
g = build_graph(6, 224)
big = _capi.Engine(g, w, device=0, dtype="bf16", max_batch=nb)
small = _capi.Engine(g, w, device=0, dtype="bf16", max_batch=8)
big.forward_u8(ims)
print(big.launch_groups())
a8 = big.tap("s8.bn", nb)
ref = []
for i in range(0, nb, 8):
    small.forward_u8(ims[i:i + 8])
    ref.append(small.tap("s8.bn", len(ims[i:i + 8])).copy())
ref = np.concatenate(ref)
d = np.abs(a8 - ref)
print("s8.bn: differing elements %d of %d, max |d| %.3e (absmax %.3e)" % ((d > 0).sum(), d.size, d.max(), np.abs(ref).max()))
print("images with differences:", np.unique(np.argwhere(d > 0)[:, 0])[:40])
print("rows:", np.unique(np.argwhere(d > 0)[:, 1]), "cols:", np.unique(np.argwhere(d > 0)[:, 2]), "ch:", np.unique(np.argwhere(d > 0)[:, 3]))
# run the big batch again: reproducible?
big.forward_u8(ims)
b8 = big.tap("s8.bn", nb)
print("second pass equals first:", np.array_equal(a8, b8), " second pass vs ref differing:", int((b8 != ref).sum()))
