"""GPU box, diagnostic library built by tools/build_clock.sh: in-kernel clock per launch of the forward pass.
Runs the pass back to back for a few seconds on random images (the chip settles at the clock it holds under this load),
then sets RN_CLOCK_REPORT for ONE more pass, which the library reports on: per launch the median over workgroups of
delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
usage: ROOMNET_HIP_LIB=roomnet_amd/lib/libroomnet_hip_clock.so python tools/clock_run.py [seconds] [dtype]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import perf_batch
from roomnet_amd.tf_bundle import BundleReader

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
B = 256
e = _capi.Engine(build_graph(6, 224), BundleReader('roomnet_amd/final_model/roomnet').load_all(), dtype=dtype, max_batch=B)
ims = perf_batch(B, 224, seed=0)
d_in = e.device_malloc(ims.nbytes)
d_p = e.device_malloc(B * 6 * 4)
d_i = e.device_malloc(B * 8)
e.h2d(d_in, ims)
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(50):
        e.forward_u8_device(d_in, B, d_p, d_i)
    n += 50
os.environ['RN_CLOCK_REPORT'] = '1'
e.forward_u8_device(d_in, B, d_p, d_i)          # the reported pass: enqueued right behind the others, synchronised at its end
el = time.perf_counter() - t0
print('[clock] %s, batch %d: %d passes back to back in %.2f s (%.0f img/s incl. the stamps), then the reported pass' % (dtype, B, n, el, n * B / el))
e.close()
