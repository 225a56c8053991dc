"""GPU box: sweep input sides through the fused 16-bit path against the float32 per-node path (random weights of the graph's shapes
are not needed: the reference checkpoint's conv stack is size-agnostic; only the first dense layer depends on the side)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader

w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
rng = np.random.default_rng(0)
sides = [int(x) for x in sys.argv[1:]] or [200, 208, 216, 224, 232, 240, 256, 288, 300, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640]
for side in sides:
    try:
        g = build_graph(6, side)
    except Exception as ex:
        print(side, "graph:", str(ex)[:80]); continue
    flat = g.flatten_len if hasattr(g, "flatten_len") else None
    # the checkpoint's first dense kernel is [64, 32]: other sides need their own; draw one with the checkpoint's statistics
    ww = dict(w)
    ww["dense/kernel"] = (np.random.default_rng(side).standard_normal((g.flat_len, 32)) * (0.5 / np.sqrt(g.flat_len))).astype(np.float32)
    try:
        nb = 8
        ims = rng.integers(0, 256, (nb, side, side, 3), dtype=np.uint8)
        for i in range(nb):                                   # smooth fields like photographs
            ims[i] = (ims[i].astype(np.float32) * 0.25 + 96 + 64 * np.sin(np.arange(side)[None, :, None] / (7.0 + i))).clip(0, 255).astype(np.uint8)
        res = {}
        for dt in ("f32", "bf16", "f16"):
            e = _capi.Engine(g, ww, device=0, dtype=dt, max_batch=nb)
            t0 = time.time()
            ids, probs = e.forward_u8(ims)
            res[dt] = (ids.copy(), probs.copy(), e.launch_groups() if dt != "f32" else None)
            e.close()
        d_bf = np.abs(res["bf16"][1] - res["f32"][1]).max()
        d_f16 = np.abs(res["f16"][1] - res["f32"][1]).max()
        print(side, "ok  max|dprob| bf16 %.4f f16 %.4f  ids equal %s %s  launches %s" % (d_bf, d_f16, (res["bf16"][0] == res["f32"][0]).all(), (res["f16"][0] == res["f32"][0]).all(), res["bf16"][2]))
    except Exception as ex:
        print(side, "FAILED:", str(ex)[:200])
