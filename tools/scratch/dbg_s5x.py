"""GPU box: the 16x16x32 stage-5 kernel against the row-streaming one (pair32 engine = round-2 kernels) and the oracle (scratch)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader
TAP = sys.argv[1] if len(sys.argv) > 1 else 's5.bn2'
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
ims = parity_batch(224, seed=1)[[14, 30, 2, 22, 9]]
for dt in ('bf16', 'f16'):
    a = _capi.Engine(build_graph(6, 224), w, dtype=dt, max_batch=8)
    b = _capi.Engine(build_graph(6, 224), w, dtype=dt, max_batch=8, pair32=True)
    ia, pa = a.forward_u8(ims)
    ib, pb = b.forward_u8(ims)
    xa, xb = a.tap(TAP, len(ims)), b.tap(TAP, len(ims))
    d = np.abs(xa - xb)
    tol = 0.02 * np.abs(xb).max()
    bad = np.argwhere(d > tol)
    print(dt, 'ids', ia.tolist(), ib.tolist(), 'max|dprob| %.3g' % np.abs(pa - pb).max(), TAP, xa.shape,
          'max|d| %.4g of absmax %.4g, differing %.2f%%, finite %s, bad(>2%%) %d' % (d.max(), np.abs(xb).max(), 100.0 * (d > 0).mean(), np.isfinite(xa).all(), len(bad)))
    if len(bad):
        print('  first bad', bad[:10].tolist())
        print('  bad cols', np.unique(bad[:, 2])[:50].tolist())
        print('  bad rows', np.unique(bad[:, 1])[:50].tolist())
        print('  bad chans', np.unique(bad[:, 3])[:64].tolist())
    a.close(); b.close()
