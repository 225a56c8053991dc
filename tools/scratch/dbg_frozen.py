import sys, numpy as np
sys.path.insert(0,'.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from roomnet_amd.synth import parity_set
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
f = np.load('tests/golden/class_fields.npz')['fields_u8']
ims = parity_set(224, f)[[1, 14, 22, 30, 37, 44, 52, 60]]
for dt in ('bf16', 'f16'):
    a = _capi.Engine(build_graph(6,224), w, dtype=dt, max_batch=8)
    b = _capi.Engine(build_graph(6,224), w, dtype=dt, max_batch=8, compute_frozen=True)
    ia, pa = a.forward_u8(ims); ib, pb = b.forward_u8(ims)
    ta, tb = a.tap('s3.bn2', 8), b.tap('s3.bn2', 8)
    print(dt, 's3.bn2 identical:', np.array_equal(ta, tb), 'n diff', int((ta != tb).sum()), 'max', float(np.abs(ta-tb).max()), 'probs identical', np.array_equal(pa, pb), ia.tolist())
