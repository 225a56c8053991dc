"""GPU box: the 16x16x32 fused pair against the 32x32x16 one (scratch)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
ims = parity_batch(224, seed=1)[[14, 30, 2, 22, 9]]
for dt in ('bf16', 'f16'):
    a = _capi.Engine(build_graph(6, 224), w, dtype=dt, max_batch=8)
    b = _capi.Engine(build_graph(6, 224), w, dtype=dt, max_batch=8, pair32=True)
    ia, pa = a.forward_u8(ims)
    ib, pb = b.forward_u8(ims)
    xa, xb = a.tap('s3.bn2', len(ims)), b.tap('s3.bn2', len(ims))
    d = np.abs(xa - xb)
    bad = np.argwhere(d > 0.05 * np.abs(xb).max())
    print(dt, 'ids', ia.tolist(), ib.tolist(), 'max|dprob| %.3g' % np.abs(pa - pb).max(), 's3.bn2 shape', xa.shape,
          'max|d| %.4g of absmax %.4g, differing %.3f%%, finite %s' % (d.max(), np.abs(xb).max(), 100.0 * (d > 0).mean(), np.isfinite(xa).all()))
    if len(bad):
        print('  bad elements', len(bad), 'first', bad[:12].tolist())
        cols = np.unique(bad[:, 2]); rows = np.unique(bad[:, 1]); ch = np.unique(bad[:, 3])
        print('  bad cols', cols[:40].tolist(), 'rows', rows[:20].tolist(), len(rows), 'chans', ch.tolist())
    a.close(); b.close()
