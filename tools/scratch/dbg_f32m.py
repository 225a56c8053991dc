"""GPU box: where does the float32 matrix-core path (no taps) leave the per-node path?  Per stage output: max |err| / absmax,
then for the first bad stage the error by channel, by column and by row."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader

w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
ims = parity_batch(224, seed=1)[[14, 30]]
g = build_graph(6, 224)
ref = _capi.Engine(g, w, device=0, dtype="f32", max_batch=2, taps=True)
mm = _capi.Engine(g, w, device=0, dtype="f32", max_batch=2)
ref.forward_u8(ims)
mm.forward_u8(ims)
names = ["s0.bn", "s1.bn", "s2.bn", "s3.bn2", "s4.bn", "s5.bn2", "s6.bn", "s7.bn", "s8.bn", "s9.bn2"]
shown = False
for nm in names:
    a, b = mm.tap(nm, 2), ref.tap(nm, 2)
    e = np.abs(a - b)
    rel = float(e.max() / max(np.abs(b).max(), 1e-9))
    print("%-8s shape %-18s max|err|/absmax %.3e   frac elements off by > 1e-4 absmax: %.4f" % (nm, a.shape, rel, float((e > 1e-4 * np.abs(b).max()).mean())))
    if rel > 1e-3 and not shown:
        shown = True
        am = float(np.abs(b).max())
        print("   by channel:", np.array2string(e.max(axis=(0, 1, 2)) / am, precision=2, max_line_width=250))
        print("   by column :", np.array2string(e.max(axis=(0, 1, 3)) / am, precision=1, max_line_width=250))
        print("   by row    :", np.array2string(e.max(axis=(0, 2, 3)) / am, precision=1, max_line_width=250))
        print("   image 0 vs 1:", e[0].max() / am, e[1].max() / am)
