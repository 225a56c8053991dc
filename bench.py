#!/usr/bin/env python3
"""Headline benchmark: images/sec of the RoomNet forward pass (BASELINE.json metric) on
synthetic 224x224 uint8 batches, batch 256 per GPU, 16-bit storage / fp32 accumulate.

  python bench.py --gpus 1 --steps 200 --warmup 20      (the defaults: 0.4 s of GPU time + ~15 s of cpu_baseline)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch that is already resident in HBM:
uint8 BGR [B,224,224,3] -> stage kernels -> head -> probs [B,6] + ids [B] in HBM, plus
(N > 1) one RCCL all-gather of every rank's probs+ids (32 bytes per image).  One process per GPU; weak
scaling (each rank owns its own batch of B images: BASELINE config 4 is 8 x 256).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      the dominant kernel (longest stage launch, timed live with HIP events on
                the library's stream): algorithmic stage-boundary bytes / duration vs 8 TB/s
  cpu_baseline  the oracle's plain-C restatement (oracle/tf_ops.c) timed on this host's
                cores on a bounded sample -- a reported baseline, not the target
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_PEAK_16 = 2.5e15      # FLOP/s dense bf16/fp16


def stage_bytes_per_image(graph, elem_bytes):
    """Algorithmic (stage-boundary model) bytes each stage moves per image."""
    out = []
    for s in graph.stages:
        in_b = s.in_side * s.in_side * s.cin * (1 if s.index == 0 else elem_bytes)
        n = in_b + s.out_side * s.out_side * s.cout * elem_bytes
        if s.residual:
            n += s.skip_side * s.skip_side * s.cout * elem_bytes
        out.append(n)
    return out


def measured_traffic(stage, batch, side, dtype):
    """HBM bytes per launch of `stage` from the newest committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json,
    built by tools/hbm_traffic.py from `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command), or None
    when no profile matches this configuration.  bench.py cannot profile itself."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if (t.get("batch"), t.get("im_side"), t.get("dtype")) != (batch, side, dtype):
            continue
        for st in t.get("stages", []):
            if st.get("stage") == stage:
                return int(st["traffic_bytes"])
    return None


def cpu_baseline(weights, side, budget_s=15.0):
    """Time the plain-C oracle on a bounded sample (rank 0, N=1 only)."""
    from oracle import c_oracle
    from roomnet_amd.synth import perf_batch
    cores = c_oracle.max_threads()
    ims = perf_batch(8, side, seed=0)
    c_oracle.infer(weights, ims[:1])                      # warm-up (thread pool, page-in)
    t0 = time.perf_counter()
    done = 0
    reps = 0
    while True:
        c_oracle.infer(weights, ims)
        done += len(ims)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 40:
            break
    el = time.perf_counter() - t0
    return {"value": done / el, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d passes over 8 uniform-noise %dx%d images in batch-8 mode, plain-C fp32 restatement "
                      "of the reference graph (oracle/tf_ops.c, OpenMP, %d threads), %.1f s" %
                      (reps, side, side, cores, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--side", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-parity-check", action="store_true", help="skip the golden check (timing experiments with garbage results)")
    ap.add_argument("--stage-launches", action="store_true",
                    help="one launch per conv stage (RN_FLAG_STAGE_LAUNCHES): the unfused comparison arm")
    ap.add_argument("--pcie", action="store_true",
                    help="also time the host-buffer entry point (rn_forward_u8: H2D copy + forward + D2H copy); reported as "
                         "path.pcie_inclusive_images_per_sec, never as `value`")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # before anything initialises the HSA runtime: the host driver only supports dmabuf IPC
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda is not available)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from roomnet_amd import _capi
    from roomnet_amd.graph import build_graph
    from roomnet_amd.synth import perf_batch
    from roomnet_amd.tf_bundle import BundleReader

    weights = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
    graph = build_graph(6, args.side)
    if args.side != 224:
        # the shipped dense/kernel only fits 224: seeded synthetic first dense kernel
        weights = dict(weights)
        weights["dense/kernel"] = np.random.default_rng(600).uniform(
            -0.04, 0.04, (graph.flat_len, 32)).astype(np.float32)
    B = args.batch
    eng = _capi.Engine(graph, weights, device=local_rank, dtype=args.dtype, max_batch=B,
                       stage_launches=args.stage_launches)

    ims = torch.from_numpy(perf_batch(B, args.side, seed=rank)).to(dev)
    # probs [B,6] fp32 and ids [B] int64 live in ONE byte buffer per rank, so the result exchange is a single
    # all-gather (32 bytes per image) instead of two latency-bound collectives
    from roomnet_amd.parallel import result_buffers
    combo, probs, ids = result_buffers(B, graph.num_classes, dev)
    if world > 1:
        g_combo = torch.empty((world * combo.numel(),), dtype=torch.uint8, device=dev)
    # run the library on torch's current stream so the collective is ordered behind the kernels
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        eng.forward_u8_device(ims.data_ptr(), B, probs.data_ptr(), ids.data_ptr())
        if world > 1:
            dist.all_gather_into_tensor(g_combo, combo)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: outputs are a distribution and an argmax of it
    p = probs.cpu().numpy()
    i = ids.cpu().numpy()
    if not args.no_parity_check:
        assert np.allclose(p.sum(1), 1.0, atol=1e-4) and (p.argmax(1) == i).all()

    # ---- per-stage device time (HIP events on the launch stream), separate from the timed region
    eng.set_profiling(True)
    stage_ms = np.zeros(len(graph.stages))
    head_ms = 0.0
    total_ms = 0.0
    nprof = max(1, args.profile_steps)
    for _ in range(nprof):
        eng.forward_u8_device(ims.data_ptr(), B, probs.data_ptr(), ids.data_ptr())
        t = eng.timing()
        stage_ms += np.array(t["stage_ms"])
        head_ms += t["head_ms"]
        total_ms += t["total_ms"]
    eng.set_profiling(False)
    stage_ms /= nprof
    head_ms /= nprof
    total_ms /= nprof

    pcie_rate = None
    if args.pcie and world == 1:
        host_ims = perf_batch(B, args.side, seed=rank)
        eng.forward_u8(host_ims)
        t1 = time.perf_counter()
        for _ in range(max(3, args.steps // 4)):
            eng.forward_u8(host_ims)
        pcie_rate = B * max(3, args.steps // 4) / (time.perf_counter() - t1)

    if rank == 0:
        elem = 4 if args.dtype == "f32" else 2
        sbytes = stage_bytes_per_image(graph, elem)
        dom = int(np.argmax(stage_ms))
        dom_bytes = sbytes[dom] * B
        dom_s = stage_ms[dom] * 1e-3
        achieved = dom_bytes / dom_s
        value = world * B * args.steps / elapsed
        bytes_per_img = graph.boundary_elements_per_image() * elem
        out = {
            "metric": "images/sec, 224x224 batch-256 RoomNet inference" if args.side == 224 and B == 256
                      else "images/sec, %dx%d batch-%d RoomNet inference" % (args.side, args.side, B),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "RoomNet forward (reference final_model weights), uint8 BGR %dx%dx3 in HBM -> "
                                   "probs+ids in HBM, batch %d per GPU, %s storage / fp32 accumulate%s"
                                   % (args.side, args.side, B, args.dtype,
                                      ", RCCL all-gather of probs+ids" if world > 1 else ""),
                       "images_per_gpu": B, "global_batch": world * B, "im_side": args.side,
                       "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": measured_traffic(dom, B, args.side, args.dtype),
                         "kernel": ("stage_rw_kernel, stage %d (%d->%d ch%s)" % (
                                        dom, graph.stages[dom].cin, graph.stages[dom].cout,
                                        " + residual" if graph.stages[dom].residual else ""))
                                   if dom > 0 and args.dtype != "f32" else "stage %d" % dom,
                         "kernel_ms": float(stage_ms[dom]), "algorithmic_bytes_per_launch": int(dom_bytes)},
            "path": {"algorithmic_bytes_per_image": int(bytes_per_img),
                     "hbm_frac": value * bytes_per_img / (world * HBM_PEAK),
                     "mfma_frac": value * graph.flops_per_image() / (world * MFMA_PEAK_16),
                     "stage_ms": [float(x) for x in stage_ms], "head_ms": float(head_ms),
                     "forward_ms_events": float(total_ms),
                     "stage_hbm_frac": [float(sbytes[k] * B / (stage_ms[k] * 1e-3) / HBM_PEAK)
                                        for k in range(len(sbytes))]},
        }
        if pcie_rate is not None:
            out["path"]["pcie_inclusive_images_per_sec"] = pcie_rate
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, args.side)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
