import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
weights = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
ims = parity_batch(224, seed=1)[[14]]
nb = 1
fused = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="bf16", max_batch=nb)
plain = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="bf16", max_batch=nb, stage_launches=True)
fused.forward_u8(ims); plain.forward_u8(ims)
a, b = fused.tap("s1.bn", nb)[0], plain.tap("s1.bn", nb)[0]
bad = a != b
print("mismatch", int(bad.sum()), "of", bad.size, "max abs", float(np.abs(a - b).max()))
rows = bad.sum(axis=(1, 2)); cols = bad.sum(axis=(0, 2)); ch = bad.sum(axis=(0, 1))
print("rows bad (first 40):", rows[:40].tolist())
print("cols bad by tile pos (col % 29):", [int(cols[np.arange(len(cols)) % 29 == k].sum()) for k in range(29)])
print("cols bad by tile:", [int(cols[29 * t:29 * t + 29].sum()) for t in range(8)])
print("chan:", ch.tolist())
print("sample a,b at row 5 col 3:", a[5, 3, :6], b[5, 3, :6])
print("sample a,b at row 5 col 28:", a[5, 28, :6], b[5, 28, :6])
