"""Debug helper (GPU box): per-row / per-column / per-channel error map of stage outputs vs the C oracle.
usage: python tools/dbg_stage.py s2.bn [s3.bn2 ...] [--dtype bf16] [--n 2]"""
import sys, argparse
import numpy as np
sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from roomnet_amd.synth import parity_batch
from oracle import c_oracle
ap = argparse.ArgumentParser(); ap.add_argument('nodes', nargs='+'); ap.add_argument('--dtype', default='bf16'); ap.add_argument('--n', type=int, default=2)
a = ap.parse_args()
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
ims = parity_batch(224, 1)[:a.n]
ref = c_oracle.infer(w, ims, taps=True)
import os
e = _capi.Engine(build_graph(6, 224), w, dtype=a.dtype, max_batch=a.n)
e.forward_u8(ims)
np.set_printoptions(linewidth=250)
for name in a.nodes:
    got = e.tap(name, a.n); want = np.asarray(ref['taps'][name])
    bad = ~np.isfinite(got)
    err = np.where(bad, 1e9, np.abs(got - want)); am = np.abs(want).max()
    print(name, got.shape, 'non-finite', int(bad.sum()), 'rel', np.where(bad, 0, err).max() / am)
    print(' bad rows  ', np.nonzero((err / am > 0.05).any(axis=(0, 2, 3)))[0])
    print(' bad cols  ', np.nonzero((err / am > 0.05).any(axis=(0, 1, 3)))[0])
    print(' bad chans ', np.nonzero((err / am > 0.05).any(axis=(0, 1, 2)))[0])
    print(' bad images', np.nonzero((err / am > 0.05).any(axis=(1, 2, 3)))[0])
    if bad.any() or (err / am > 0.05).any():
        idx = np.argwhere(err / am > 0.05)[:6]
        for (n_, y_, x_, c_) in idx:
            print('  at n=%d y=%d x=%d c=%d: got %r want %r | got[y-1] %r got[y+1] %r | want[y+-1] %r %r' % (
                n_, y_, x_, c_, got[n_, y_, x_, c_], want[n_, y_, x_, c_], got[n_, y_ - 1, x_, c_], got[n_, min(y_ + 1, got.shape[1]-1), x_, c_],
                want[n_, y_ - 1, x_, c_], want[n_, min(y_ + 1, got.shape[1]-1), x_, c_]))
        y_ = idx[0][1]; n_ = idx[0][0]
        print('  row y=%d c=0 got :' % y_, np.round(got[n_, y_, 112:150, 0], 2))
        print('  row y=%d c=0 want:' % y_, np.round(want[n_, y_, 112:150, 0], 2))
        print('  row y=%d c=2 got :' % y_, np.round(got[n_, y_, 112:150, 2], 2))
        print('  row y=%d c=2 want:' % y_, np.round(want[n_, y_, 112:150, 2], 2))
