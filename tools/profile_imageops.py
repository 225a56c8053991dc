#!/usr/bin/env python3
"""GPU box: the batched crop + resize kernel (rn_crop_resize_batch_u8_device, network.py:149-152) on device-resident images:
256 x 1080p (1080 x 1920) and 256 x VGA (480 x 640) -> 224 x 224, `--repeat` launches each.  Prints one JSON line per case:
images/sec of the kernel alone (wall clock around the launches), its algorithmic bytes -- the source bytes the bilinear taps
touch at 64-byte granularity + 3 x 224 x 224 written -- and that rate against 8 TB/s.  Run under rocprofv3 for profiles/r6_imageops.*."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
import ctypes as C


def touched_bytes(h, w, S=224, gran=64):
    """Source bytes of the centred square the INTER_LINEAR taps of an S x S output touch, in `gran`-byte pieces of each row."""
    side = min(h, w)
    x0 = (w - side) // 2
    scale = side / S
    def taps(n):
        f = (np.arange(S) + 0.5) * scale - 0.5
        s0 = np.clip(np.floor(f).astype(int), 0, n - 1)
        return s0, np.minimum(s0 + 1, n - 1)
    xa, xb = taps(side)
    ya, yb = taps(side)
    rows = np.unique(np.concatenate([ya, yb]))
    b0 = ((x0 + xa) * 3) // gran
    b1 = ((x0 + xb) * 3 + 2) // gran
    pieces = np.unique(np.concatenate([b0, b1, (b0 + b1) // 2]))
    return int(len(rows) * len(pieces) * gran)


ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--repeat", type=int, default=50)
ap.add_argument("--case", default="both", choices=["1080p", "vga", "both"])
args = ap.parse_args()
w = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
e = _capi.Engine(build_graph(6, 224), w, device=0, dtype="bf16", max_batch=args.n)
rng = np.random.default_rng(0)
for name, (h, wd) in (("1080p", (1080, 1920)), ("vga", (480, 640))):
    if args.case not in (name, "both"):
        continue
    base = rng.integers(0, 256, (h, wd, 3), dtype=np.uint8)
    d_srcs = []
    for i in range(args.n):
        d = e.device_malloc(base.nbytes)
        e.h2d(d, np.roll(base, i * 7, axis=1))
        d_srcs.append(d)
    d_dst = e.device_malloc(args.n * 224 * 224 * 3)
    ptrs = (C.c_void_p * args.n)(*d_srcs)
    hs = (C.c_int * args.n)(*([h] * args.n))
    ws = (C.c_int * args.n)(*([wd] * args.n))
    call = lambda: _capi._check(e.lib, e.lib.rn_crop_resize_batch_u8_device(e.handle, ptrs, hs, ws, args.n, C.c_void_p(d_dst)), "batch resize")
    for _ in range(3):
        call()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        call()
    e.sync()
    dt = (time.perf_counter() - t0) / args.repeat
    # one launch per image, as before round 6 (rn_crop_resize_u8_device), for comparison
    t0 = time.perf_counter()
    for _ in range(3):
        for i in range(args.n):
            e.lib.rn_crop_resize_u8_device(e.handle, C.c_void_p(d_srcs[i]), h, wd, C.c_void_p(d_dst), i)
    e.sync()
    dt1 = (time.perf_counter() - t0) / 3
    alg = args.n * (touched_bytes(h, wd) + 224 * 224 * 3)
    print(json.dumps({"case": "%d x %s (%dx%d) -> 224x224, device-resident" % (args.n, name, wd, h), "launches": 1, "ms_per_batch": dt * 1e3,
                      "images_per_sec": args.n / dt, "algorithmic_bytes_per_batch": alg, "achieved_GBps": alg / dt / 1e9,
                      "frac_of_8TBps": alg / dt / 8e12, "source_bytes_uploaded_per_image": min(h, wd) ** 2 * 3,
                      "one_launch_per_image_ms_per_batch": dt1 * 1e3, "one_launch_per_image_images_per_sec": args.n / dt1}), flush=True)
    for d in d_srcs:
        e.device_free(d)
    e.device_free(d_dst)
e.close()
