import sys, numpy as np
sys.path.insert(0,'.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from roomnet_amd.synth import parity_set
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
f = np.load('tests/golden/class_fields.npz')['fields_u8']
ims = parity_set(224, f)[[1, 14, 22, 30, 52, 60]]
for dt in ('f16', 'bf16'):
    a = _capi.Engine(build_graph(6,224), w, dtype=dt, max_batch=8)
    b = _capi.Engine(build_graph(6,224), w, dtype=dt, max_batch=8, stage_launches=True)
    ia, pa = a.forward_u8(ims); ib, pb = b.forward_u8(ims)
    ta, tb = a.tap('s1.bn', 6), b.tap('s1.bn', 6)
    d = np.abs(ta - tb)
    print(dt, 'groups', a.launch_groups(), 's1.bn max diff', d.max(), 'absmax', np.abs(tb).max(), 'nonfinite', (~np.isfinite(ta)).sum())
    print('  per-image max', d.reshape(6,-1).max(1))
    bad = np.argwhere(d > 0.02*np.abs(tb).max())
    print('  n bad', len(bad), bad[:10].tolist())
    print('  rows with bad', sorted(set(bad[:,1].tolist()))[:20], 'cols', sorted(set(bad[:,2].tolist()))[:40])
    print('  dprob', np.abs(pa-pb).max(), ia, ib)
