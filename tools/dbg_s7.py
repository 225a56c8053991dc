import numpy as np, sys
sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from roomnet_amd.synth import parity_batch
from oracle import c_oracle
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
ims = parity_batch(224, 1)[[14]]
ref = c_oracle.infer(w, ims, taps=True)
e = _capi.Engine(build_graph(6, 224), w, dtype='bf16', max_batch=1)
e.forward_u8(ims)
for name in ('s6.bn', 's7.bn'):
    got = e.tap(name, 1)[0]; want = np.asarray(ref['taps'][name])[0]
    err = np.abs(got - want); am = np.abs(want).max()
    print(name, got.shape, 'rel', err.max() / am)
    print(' by row   ', np.round(err.max(axis=(1, 2)) / am, 3))
    print(' by col   ', np.round(err.max(axis=(0, 2)) / am, 3))
    print(' by chan  ', np.round(err.max(axis=(0, 1)) / am, 3))
    print(' got/want sample', got[5, 5, :4], want[5, 5, :4])

# which kernel rows made it into the result?
from oracle import roomnet_ref as R
x = np.asarray(ref['taps']['s6.bn'])[0:1].astype(np.float64)
k = w['conv2d_7/kernel'].astype(np.float64)
bnp = [w['batch_normalization_9/' + n].astype(np.float64) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
got = e.tap('s7.bn', 1)[0].astype(np.float64)
import itertools
for rows in ([0], [1], [2], [0, 1], [0, 2], [1, 2], [0, 1, 2]):
    kk = np.zeros_like(k); kk[rows] = k[rows]
    conv = R.relu6(R.conv2d_valid(x, kk))
    out = R.fused_batch_norm_infer(R.avg_pool_valid(conv, 4, 2), *bnp)[0]
    print('kernel rows', rows, 'max |got - oracle_subset| / absmax = %.4f' % (np.abs(got - out).max() / np.abs(out).max()))
