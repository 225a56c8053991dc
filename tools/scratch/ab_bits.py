"""GPU box: are two builds of the library bit-identical?  tools/ab_bits.py LIB_A LIB_B [side]  (taps of every stage output,
probabilities and ids for 9 parity images; each library is loaded in its own subprocess)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from roomnet_amd.synth import parity_batch
side = int(sys.argv[2])
g = build_graph(6, side)
w = dict(BundleReader(%r).load_all())
if side != 224:
    w["dense/kernel"] = np.random.default_rng(600).uniform(-0.04, 0.04, (g.flat_len, 32)).astype(np.float32)
ims = parity_batch(side, seed=1)[[3, 9, 14, 22, 27, 30, 33, 36, 39]]
out = {}
for dt in ("bf16", "f16"):
    e = _capi.Engine(g, w, device=0, dtype=dt, max_batch=len(ims), stage_launches=True)
    ids, probs = e.forward_u8(ims)
    out[dt + "_ids"], out[dt + "_probs"] = ids, probs
    for s in g.stages:
        name = "s%%d.%%s" %% (s.index, "bn2" if s.residual else "bn")
        out[dt + "_" + name] = e.tap(name, len(ims))
    e.close()
np.savez(sys.argv[1], **out)
''' % (ROOT, os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet"))


def run(lib, path, side):
    env = dict(os.environ, ROOMNET_HIP_LIB=lib) if lib != "base" else {k: v for k, v in os.environ.items() if k != "ROOMNET_HIP_LIB"}
    subprocess.run([sys.executable, "-c", CHILD, path, str(side)], check=True, env=env, cwd=ROOT)
    return np.load(path)


if __name__ == "__main__":
    side = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    a = run(sys.argv[1], "/tmp/ab_a.npz", side)
    b = run(sys.argv[2], "/tmp/ab_b.npz", side)
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        if not same:
            bad += 1
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            print("DIFF %-12s %d of %d elements, max |d| %.3g" % (k, int((a[k] != b[k]).sum()), a[k].size, d.max()))
    print("bit-identical" if bad == 0 else "%d arrays differ" % bad)
    sys.exit(1 if bad else 0)
