"""GPU box experiment: do consecutive forward passes overlap when they alternate between two handles on two streams?"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import perf_batch
from roomnet_amd.tf_bundle import BundleReader
B = 256
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
dev = torch.device('cuda:0')
ims = torch.from_numpy(perf_batch(B, 224, seed=0)).to(dev)
def make():
    e = _capi.Engine(build_graph(6, 224), w, device=0, dtype='bf16', max_batch=B)
    s = torch.cuda.Stream()
    e.set_stream(s.cuda_stream)
    p = torch.empty((B, 6), dtype=torch.float32, device=dev)
    i = torch.empty((B,), dtype=torch.int64, device=dev)
    return e, s, p, i
engs = [make() for _ in range(3)]
def run(nstream, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        e, s, p, i = engs[k % nstream]
        e.forward_u8_device(ims.data_ptr(), B, p.data_ptr(), i.data_ptr())
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)
run(1, 300)
for rep in range(2):
    for ns in (1, 2, 3):
        print('streams %d: %.0f img/s' % (ns, run(ns, 300)))
ref = engs[0][3].cpu().numpy()
assert (engs[1][3].cpu().numpy() == ref).all() and (engs[2][3].cpu().numpy() == ref).all()
