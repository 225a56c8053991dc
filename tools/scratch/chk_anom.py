"""GPU box: where do the folded and the computing arm differ most in s5.bn2 (f16, nb 160)?  Same question to the round-5 library."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from conftest import parity_set_of
w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
g = build_graph(6, 224)
ims_all = parity_set_of(224)
nb = 160
ims = ims_all[(np.arange(nb) * 5) % len(ims_all)]
for lib in (None, os.path.join(ROOT, "tools", "ab", "libroomnet_hip_r5.so")):
    if lib and not os.path.isfile(lib):
        continue
    for dt in ("f16", "bf16"):
        fold = _capi.Engine(g, w, device=0, dtype=dt, max_batch=nb, lib_path=lib)
        full = _capi.Engine(g, w, device=0, dtype=dt, max_batch=nb, compute_frozen=True, lib_path=lib)
        for rep in range(3):
            fold.forward_u8(ims); full.forward_u8(ims)
            a, b = fold.tap("s5.bn2", nb), full.tap("s5.bn2", nb)
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
            idx = np.unravel_index(int(d.argmax()), d.shape)
            big = np.argwhere(d > 0.01)
            print("lib %s %s rep %d: max|d| %.4g at %s fold %.5f full %.5f; elements with |d| > 0.01: %d %s" % (
                "r5" if lib else "new", dt, rep, d.max(), idx, a[idx], b[idx], len(big), big[:6].tolist()), flush=True)
            a4, b4 = fold.tap("s4.bn", nb), full.tap("s4.bn", nb)
            d4 = np.abs(a4.astype(np.float64) - b4.astype(np.float64))
            print("     s4.bn max|d| %.4g" % d4.max())
        fold.close(); full.close()
f32 = _capi.Engine(g, w, device=0, dtype="f32", max_batch=nb)
f32.forward_u8(ims)
r = f32.tap("s5.bn2", nb)
print("f32 handle absmax", np.abs(r).max())
np.save(os.path.join(ROOT, "gpurun_out", "r6", "s5bn2_f32_sample.npy"), r[:2])
