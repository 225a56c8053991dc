#!/usr/bin/env python3
"""Single-process multi-GPU benchmark of the C ABI's rn_group_* entry (include/roomnet_hip.h): one process, N devices,
ctypes only -- no torch, no torch.distributed.  Prints ONE JSON line with the same fields as bench.py.

    python tools/group_bench.py --gpus N [--steps K] [--warmup W] [--batch 256] [--side 224] [--dtype bf16]

A step = every device classifies its resident shard (batch images of side x side, uint8 BGR, already in its HBM) and one
RCCL all-gather leaves all N x batch results on every device (SURVEY 8d "multi-GPU timing": shards resident -> gathered
[N, 6] on every rank); value = N x batch x K / wall time between two rn_group_sync().  Before the timed region the
group's host entry classifies the parity images and the result is compared with tests/golden/ (the same gate as
bench.py); after it every device's slot of the gathered buffer is checked to hold valid results.
On an 8-GPU node: python tools/group_bench.py --gpus 8     (nothing else is needed; not yet run on hardware, the
development pool has one MI355X per box)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

from bench import HBM_PEAK, MFMA_PEAK_16, check_parity  # noqa: E402  (constants + the golden gate; bench.py imports torch lazily)
from roomnet_amd import _capi  # noqa: E402
from roomnet_amd.graph import build_graph  # noqa: E402
from roomnet_amd.synth import perf_batch  # noqa: E402
from roomnet_amd.tf_bundle import BundleReader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-steps", type=int, default=240, help="untimed passes before the warm-up steps (engine clock out of idle; see bench.py)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 256 at 224, 64 at 600)")
    ap.add_argument("--side", type=int, default=224)
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16"])
    ap.add_argument("--dry-run", action="store_true",
                    help="no library, no GPU: print the plan of the run (devices, contiguous shards of the global batch, bytes of the "
                         "all-gather) as one JSON line and exit -- what a box without N GPUs can check of this tool")
    args = ap.parse_args()
    B = args.batch or (256 if args.side == 224 else 64)
    dtype = args.dtype or ("bf16" if args.side == 224 else "f16")
    N = args.gpus
    if args.dry_run:
        g = build_graph(6, args.side)
        print(json.dumps({"dry_run": True, "n_gpus": N, "devices": list(range(N)), "steps": args.steps, "warmup": args.warmup,
                          "dtype": dtype, "shards": [[d * B, (d + 1) * B] for d in range(N)],
                          "all_gather_bytes_per_device": B * (g.num_classes * 4 + 8), "gathered_bytes": N * B * (g.num_classes * 4 + 8),
                          "config": {"images_per_gpu": B, "global_batch": N * B, "im_side": args.side, "parallelism": "dp%d" % N,
                                     "host": "single process, C ABI rn_group_*, ctypes"}}))
        return
    weights = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
    if args.side != 224:
        rng = np.random.default_rng(600)      # the checkpoint's dense/kernel only fits 224: seeded synthetic one
        graph0 = build_graph(6, args.side)
        weights["dense/kernel"] = rng.uniform(-0.04, 0.04, (graph0.flat_len, 32)).astype(np.float32)
    graph = build_graph(6, args.side)
    grp = _capi.Group(graph, weights, devices=list(range(N)), dtype=dtype, max_batch_per_device=B)
    lib = grp.lib
    C.CDLL(None).fflush(None)      # librccl prints a banner through C stdio: out now, so the JSON line stays the last line of stdout
    parity = check_parity(grp.forward_u8, args.side, dtype, B * N)
    # resident shards: device d holds its own seeded batch
    shards = (C.c_void_p * N)()
    counts = (C.c_int * N)(*([B] * N))
    for d in range(N):
        h = lib.rn_group_handle(grp._g, d)
        p = C.c_void_p()
        _capi._check(lib, lib.rn_device_malloc(h, B * args.side * args.side * 3, C.byref(p)), "rn_device_malloc")
        ims = perf_batch(B, args.side, seed=d)
        _capi._check(lib, lib.rn_memcpy_h2d(h, p, ims.ctypes.data, ims.nbytes), "rn_memcpy_h2d")
        shards[d] = p.value

    def step():
        _capi._check(lib, lib.rn_group_forward_u8_device(grp._g, shards, counts), "rn_group_forward_u8_device")

    for _ in range(args.spinup_steps + args.warmup):
        step()
    _capi._check(lib, lib.rn_group_sync(grp._g), "rn_group_sync")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _capi._check(lib, lib.rn_group_sync(grp._g), "rn_group_sync")
    elapsed = time.perf_counter() - t0
    # every device's view of the gathered buffer: N slots of valid results, identical on all devices
    ncls = graph.num_classes
    slot = B * (ncls * 4 + 8)
    views = []
    for d in range(N):
        buf, sb = C.c_void_p(), C.c_size_t()
        _capi._check(lib, lib.rn_group_result_buffer(grp._g, d, C.byref(buf), C.byref(sb)), "rn_group_result_buffer")
        assert sb.value == slot
        host = np.empty(slot * N, np.uint8)
        _capi._check(lib, lib.rn_memcpy_d2h(lib.rn_group_handle(grp._g, d), host.ctypes.data, buf, host.nbytes), "rn_memcpy_d2h")
        views.append(host)
        for r in range(N):
            probs = host[r * slot:r * slot + B * ncls * 4].view(np.float32).reshape(B, ncls)
            ids = host[r * slot + B * ncls * 4:(r + 1) * slot].view(np.int64)
            assert np.allclose(probs.sum(1), 1.0, atol=1e-4) and (probs.argmax(1) == ids).all(), "device %d: slot %d is not a result" % (d, r)
    for d in range(1, N):
        assert (views[d] == views[0]).all(), "device %d holds a different gathered buffer than device 0" % d
    value = N * B * args.steps / elapsed
    bytes_per_img = graph.boundary_elements_per_image() * 2
    out = {"metric": "images/sec, %dx%d batch-%d RoomNet inference" % (args.side, args.side, B),
           "value": value, "unit": "images/sec", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "spinup_steps": args.spinup_steps,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": dtype, "data": "synthetic",
           "config": {"workload": "RoomNet forward (reference final_model weights), uint8 BGR %dx%dx3 resident in each GPU's HBM -> "
                                  "probs+ids of all GPUs on every GPU, batch %d per GPU, %s storage / fp32 accumulate, one process "
                                  "(rn_group_*: one host thread, one stream per device, one RCCL all-gather per step)"
                                  % (args.side, args.side, B, dtype),
                      "images_per_gpu": B, "global_batch": N * B, "im_side": args.side, "parallelism": "dp%d" % N,
                      "host": "single process, C ABI rn_group_*, ctypes"},
           "parity": parity,
           "path": {"algorithmic_bytes_per_image": int(bytes_per_img), "hbm_frac": value * bytes_per_img / (N * HBM_PEAK),
                    "mfma_frac": value * graph.flops_per_image() / (N * MFMA_PEAK_16)}}
    grp.close()
    C.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
