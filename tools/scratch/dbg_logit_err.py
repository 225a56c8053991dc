"""GPU box: which stage makes the worst image's logit error (parity set of 40, fused 16-bit path vs float32 per-node path)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.synth import parity_batch
from roomnet_amd.tf_bundle import BundleReader

w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
ims = parity_batch(224, seed=1)
gold = np.load(os.path.join(ROOT, "tests", "golden", "parity_224.npz"))
g = build_graph(6, 224)
names = ["s0.bn", "s1.bn", "s2.bn", "s3.bn2", "s4.bn", "s5.bn2", "s6.bn", "s7.bn", "s8.bn", "s9.bn2", "d0.relu", "d1.relu", "d2.relu", "d3.relu"]
f32 = _capi.Engine(g, w, device=0, dtype="f32", max_batch=8, taps=True)
for dt in ("bf16", "f16"):
    for stagewise in (False, True):
        e = _capi.Engine(g, w, device=0, dtype=dt, max_batch=8, stage_launches=stagewise)
        errs = []
        for i in range(0, 40, 8):
            e.forward_u8(ims[i:i + 8])
            errs.append(np.abs(e.tap("d3.relu", 8) - gold["logits_f64"][i:i + 8]).max(1))
        errs = np.concatenate(errs)
        worst = np.argsort(errs)[::-1][:4]
        print(dt, "stage launches" if stagewise else "fused", "worst images", worst.tolist(), np.round(errs[worst], 4).tolist(), "mean %.4f" % errs.mean())
        wi = int(worst[0])
        c0 = wi // 8 * 8
        e.forward_u8(ims[c0:c0 + 8])
        f32.forward_u8(ims[c0:c0 + 8])
        k = wi - c0
        for nm in names:
            try:
                a = e.tap(nm, 8)[k].astype(np.float64)
            except Exception as ex:
                print("   %-8s (not tappable: %s)" % (nm, str(ex)[:40]))
                continue
            b = f32.tap(nm, 8)[k].astype(np.float64)
            d = np.abs(a - b)
            print("   %-8s absmax %8.4f  max|d| %.5f  rel %.5f  mean|d| %.6f  mean signed %.6f" % (nm, np.abs(b).max(), d.max(), d.max() / np.abs(b).max(), d.mean(), (a - b).mean()))
        print("   logits 16-bit", np.round(e.tap("d3.relu", 8)[k], 4).tolist())
        print("   logits f32   ", np.round(f32.tap("d3.relu", 8)[k], 4).tolist())
        print("   logits fp64  ", np.round(gold["logits_f64"][wi], 4).tolist())
        e.close()
