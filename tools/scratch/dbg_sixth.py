"""GPU box: where do the fused pair and the stage launches differ (1/6-scaled ReLU6 formulation)?"""
import sys
import numpy as np
sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
ims = parity_batch(224, seed=1)
nb = 33
pick = (np.arange(nb) * 7) % len(ims)
ims = ims[pick]
for dtype in ('bf16', 'f16'):
    for kw in ({}, {'pair32': True}):
        fused = _capi.Engine(build_graph(6, 224), w, device=0, dtype=dtype, max_batch=nb, **kw)
        plain = _capi.Engine(build_graph(6, 224), w, device=0, dtype=dtype, max_batch=nb, stage_launches=True)
        fused.forward_u8(ims); plain.forward_u8(ims)
        a, b = fused.tap('s3.bn2', nb), plain.tap('s3.bn2', nb)
        bad = np.argwhere(a != b)
        print(dtype, kw, 'differing', len(bad), 'of', a.size)
        for ix in bad[:6]:
            t = tuple(ix)
            print('   ', t, repr(a[t]), repr(b[t]), 'neighbours', a[t[0], t[1], max(t[2]-1,0):t[2]+2, t[3]])
        fused.close(); plain.close()
