#!/usr/bin/env python3
"""Headline benchmark: images/sec of the RoomNet forward pass (BASELINE.json metric) on
synthetic 224x224 uint8 batches, batch 256 per GPU, 16-bit storage / fp32 accumulate.

  python bench.py --gpus 1 --steps 200 --warmup 20      (the defaults: ~0.4 s of GPU time + ~20 s of cpu_baseline)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch that is already resident in HBM:
uint8 BGR [B,224,224,3] -> stage kernels -> head -> probs [B,6] + ids [B] in HBM, plus
(N > 1) one RCCL all-gather of every rank's probs+ids (32 bytes per image).  One process per GPU; weak
scaling (each rank owns its own batch of B images: BASELINE config 4 is 8 x 256).  The library and the
collective run on ONE explicit (non-default) stream, so the all-gather is stream-ordered behind the head kernel.

Before anything is timed the SAME handle classifies the 64 parity images (roomnet_amd/synth.parity_set: all six classes) and the
result is checked against tests/golden/parity_224.npz at the SURVEY 8c tolerance: the timed kernels are the tested ones.

Two measurements per run, both complete passes, both reported.  (1) the COLD / contract pass: W warm-up steps and K timed
steps on a chip that idled through the parity check, nothing else in front of it -> `cold_images_per_sec` (the strict reading
of the command line; with the driver's W = 5, K = 20 that is 33 ms of work, inside the tens of milliseconds the engine clock
needs to leave its idle state: it reads ~5 % low).  (2) `value`: W warm-up steps + K timed steps again, on the same handle,
directly behind (1) -- so `value` is preceded by W + K + W untimed-for-value steps, reported as `untimed_steps_before_value`
(+ `--spinup-steps`, default 0).  `--no-cold-pass` drops (1): then `value` IS the strict reading.

`--handles 2` (default 1): the steps alternate between two engine handles (own activation tensors, stream, result buffer) so
the last launch of step k overlaps the first of step k + 1 (+0.3 % since the back end is one launch).

`python3 bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself (`self_launch`)
before torch / HIP are touched; under torch.distributed.run it is one of the ranks.

The default single-GPU invocation of the headline configuration also measures, in the same process and parity-gated the same
way, BASELINE's other single-GPU shards: 64 x 600 x 600 fp16 (config 5's per-GPU shard), 256 x 224 x 224 float32 (config 2's
arithmetic at the headline batch) and the latency of a batch-1 call (network.py:148-156's shape) -> `other_configs`
(`--no-other-configs` skips them).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      `frac` = the whole path against the roof the target is stated in (images/sec x 27.309 MB
                / (n_gpu x 8 TB/s)); `launches[]` = every launch of a step timed live with HIP events on the launch stream, each
                with `credited_frac` (algorithmic stage-boundary bytes of the stages it computes / time: a fused launch is
                credited with tensors that stay in LDS), `physical_frac` (PMC HBM bytes / time) and `mfma_frac`;
                `bound_frac` = max(physical HBM bytes of a step / 8 TB/s, matrix flops the handle issues / dense peak) / step time:
                the distance to what THIS fusion structure must move or multiply -- it cannot pass 1
  cpu_baseline  the CPU restatements (oracle/) timed on this host's cores on a bounded sample: batch-1 loop
                (infer.py:79-82) and batch 8, all cores and 1 thread, median of 3 -- a reported baseline, not the target
"""
from __future__ import annotations

import argparse
import contextlib
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
MFMA_PEAK_F32 = 1.573e14   # FLOP/s matrix fp32

# SURVEY.md 8c tolerances: (max |dprob|, top-2 logit margin above which ids must match)
PARITY_TOL = {"f32": (1e-5, 1e-3), "bf16": (0.05, 0.2), "f16": (0.05, 0.2)}


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


def stage_flops_per_image(graph):
    return [2.0 * s.conv_side * s.conv_side * 9.0 * s.cin * s.cout for s in graph.stages]


def measured_traffic(stages, batch, side, dtype):
    """HBM bytes per launch of the launch that computes `stages`, from the newest committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic.json, built by tools/hbm_traffic.py from `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of
    this same command), or None when no profile matches this configuration.  bench.py cannot profile itself."""
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
            if st.get("stages", [st.get("stage")]) == list(stages):
                return int(st["traffic_bytes"])
    return None


def check_parity(forward, side, dtype, max_batch):
    """Classify the parity images with `forward(uint8 batch) -> (ids, probs)` and compare with the committed golden
    results (fp64 restatement, tests/golden/).  Raises AssertionError when the handle does not reproduce them."""
    from roomnet_amd.synth import parity_set
    name = "parity_%d.npz" % side
    path = os.path.join(ROOT, "tests", "golden", name)
    if not os.path.isfile(path):
        return {"checked": False, "reason": "no golden file for side %d" % side}
    g = np.load(path)
    # 40 seeded images + the 24 class-covering colour fields (all six classes of infer.py:22 are reached)
    ims = parity_set(side, np.load(os.path.join(ROOT, "tests", "golden", "class_fields.npz"))["fields_u8"])
    if "image_indices" in g.files:
        ims = ims[g["image_indices"]]
    ids, probs = [], []
    for i in range(0, len(ims), max_batch):
        a, b = forward(ims[i:i + max_batch])
        ids.append(a)
        probs.append(b)
    ids, probs = np.concatenate(ids), np.concatenate(probs)
    tol_p, tol_m = PARITY_TOL[dtype]
    err = float(np.abs(probs - g["probs_f64"]).max())
    safe = g["top2_margin"] > tol_m
    wrong = int((ids[safe] != g["ids"][safe]).sum())
    assert err <= tol_p, "parity: probabilities differ from the golden by %g (tolerance %g)" % (err, tol_p)
    assert wrong == 0, "parity: %d class ids differ from the golden where the top-2 margin exceeds %g" % (wrong, tol_m)
    return {"checked": True, "golden": "tests/golden/" + name, "images": int(len(ims)), "classes_reached": sorted(set(int(i) for i in ids)),
            "max_abs_dprob": err, "tol_dprob": tol_p,
            "ids_compared": int(safe.sum()), "ids_wrong": wrong, "ids_differ_all": int((ids != g["ids"]).sum())}


def _bounded_rate(run_one_batch, batch, budget_s=3.0, reps=3):
    """images/sec of `run_one_batch()` (which classifies `batch` images): warm-up call, then the median of `reps`
    repetitions of as many calls as fit the time budget (at least one call per repetition)."""
    t0 = time.perf_counter()
    run_one_batch()
    warm = time.perf_counter() - t0                          # includes thread-pool start-up: an upper bound per call
    if warm > budget_s:                                      # too slow to repeat: report the single (warm-up) call
        return batch / warm, 1
    calls = max(1, int(budget_s / reps / max(warm, 1e-3)))
    ts = []
    for _ in range(reps):
        t1 = time.perf_counter()
        for _ in range(calls):
            run_one_batch()
        ts.append((time.perf_counter() - t1) / calls)
    return batch / float(np.median(ts)), reps


def cpu_baseline(weights, side):
    """Time the CPU restatements on a bounded sample (rank 0, N=1 only): mode A = batch-1 loop (the reference's
    infer.py:79-82), mode B = one batch of 8; many threads and one thread; median of 3 repetitions each, every mode
    inside a ~3 s budget.  "Many threads" = the cores this process may run on, capped at 32: more MKL-DNN threads than
    that made the torch restatement 100x SLOWER on the 256-thread GPU host (0.07 images/sec at 256 threads, 12 at 1)."""
    from oracle import c_oracle, torch_ref
    from roomnet_amd.synth import perf_batch
    import torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    many = max(1, min(avail, 32))
    ims = perf_batch(8, side, seed=0)
    torch_threads_before = torch.get_num_threads()
    t_start = time.perf_counter()
    modes, used = {}, {}
    for name, batch, threads in (("torch_batch8_manythreads", 8, many), ("torch_batch1_manythreads", 1, many),
                                 ("torch_batch8_1thread", 8, 1), ("torch_batch1_1thread", 1, 1)):
        if time.perf_counter() - t_start > 25.0:             # a pathological host: keep the default run within minutes
            break
        modes[name], _ = _bounded_rate(lambda: torch_ref.infer(weights, ims[:batch], threads=threads), batch)
        used[name] = threads
    torch.set_num_threads(torch_threads_before)
    c_threads = c_oracle.max_threads()
    modes["c_batch8_allthreads"], _ = _bounded_rate(lambda: c_oracle.infer(weights, ims), 8)
    used["c_batch8_allthreads"] = c_threads
    best = max(modes, key=lambda k: modes[k])
    el = time.perf_counter() - t_start
    return {"value": modes[best], "unit": "images/sec", "cores": used[best], "kind": "port",
            "mode": best, "modes": {k: round(v, 3) for k, v in modes.items()}, "threads": used,
            "sample": "uniform-noise %dx%d images; mode A = batch-1 loop (infer.py:79-82), mode B = one batch of 8; %d threads "
                      "(of %d usable cores) and 1 thread; median of 3 repetitions within ~3 s per mode; torch-CPU restatement "
                      "(oracle/torch_ref.py, MKL-DNN conv) and plain-C restatement (oracle/tf_ops.c, OpenMP, %d threads) of "
                      "the reference graph, fp32; %.1f s in all. `value` = the fastest mode. c_batch8_allthreads is the "
                      "oracle's NAIVE direct loops (written to be read against the reference, not tuned: no blocking, no "
                      "im2col/GEMM) -- a checker's rate, not what these cores can do. CPU restatement of the "
                      "reference, not TensorFlow, and not the optimisation target"
                      % (side, side, many, avail, c_threads, el)}


def measure_other_config(_capi, torch, graph_of, weights224, dev, side, batch, dtype, warmup, steps):
    """One more single-GPU configuration in the same process: its own handle, the parity gate against its golden file, W + K
    untimed steps (the cold pass of the headline), W warm-up + K timed steps -> value, then the RN_FLAG_COMPUTE_FROZEN arm under
    the same conditions.  Inputs resident in HBM; wall clock between torch.cuda.synchronize() calls."""
    from roomnet_amd.synth import perf_batch
    graph = graph_of(side)
    w = weights224
    if side != 224:
        w = dict(weights224)
        w["dense/kernel"] = np.random.default_rng(600).uniform(-0.04, 0.04, (graph.flat_len, 32)).astype(np.float32)
    ims = torch.from_numpy(perf_batch(batch, side, seed=0)).to(dev)
    probs = torch.empty((batch, graph.num_classes), dtype=torch.float32, device=dev)
    ids = torch.empty((batch,), dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(dev)
    res = {"config": "%d x %dx%d %s" % (batch, side, side, dtype), "batch": batch, "im_side": side, "dtype": dtype,
           "steps": steps, "warmup": warmup}
    rates = {}
    for arm, cf in (("value", False), ("computing_frozen", True)):
        e = _capi.Engine(graph, w, device=dev.index, dtype=dtype, max_batch=batch, compute_frozen=cf)
        e.set_stream(stream.cuda_stream)
        try:
            if arm == "value":
                def fwd_host(b):
                    t = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
                    p = torch.empty((len(b), graph.num_classes), dtype=torch.float32, device=dev)
                    i = torch.empty((len(b),), dtype=torch.int64, device=dev)
                    torch.cuda.synchronize()
                    with torch.cuda.stream(stream):
                        e.forward_u8_device(t.data_ptr(), len(b), p.data_ptr(), i.data_ptr())
                    torch.cuda.synchronize()
                    return i.cpu().numpy(), p.cpu().numpy()
                res["parity"] = check_parity(fwd_host, side, dtype, batch)
                res["folding"] = dict(e.frozen_info(), constant_channels=e.const_info())
            with torch.cuda.stream(stream):
                for _ in range(2 * warmup + steps):
                    e.forward_u8_device(ims.data_ptr(), batch, probs.data_ptr(), ids.data_ptr())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    e.forward_u8_device(ims.data_ptr(), batch, probs.data_ptr(), ids.data_ptr())
                torch.cuda.synchronize()
                rates[arm] = batch * steps / (time.perf_counter() - t0)
        finally:
            e.close()
    elem = 4 if dtype == "f32" else 2
    res["value"] = rates["value"]
    res["unit"] = "images/sec"
    res["ms_per_step"] = batch / rates["value"] * 1e3
    res["images_per_sec_computing_them"] = rates["computing_frozen"]
    if dtype == "f32":
        res["roofline"] = {"bound": "mfma", "frac": rates["value"] * graph.flops_per_image() / MFMA_PEAK_F32,
                           "frac_computing_them": rates["computing_frozen"] * graph.flops_per_image() / MFMA_PEAK_F32,
                           "scope": "images/sec x algorithmic conv flops per image / matrix-fp32 peak (BASELINE.md section 3)"}
    else:
        bpi = graph.boundary_elements_per_image() * elem
        res["roofline"] = {"bound": "hbm", "frac": rates["value"] * bpi / HBM_PEAK,
                           "frac_computing_them": rates["computing_frozen"] * bpi / HBM_PEAK,
                           "scope": "images/sec x algorithmic stage-boundary bytes per image / 8 TB/s (BASELINE.md section 3)"}
    return res


def measure_latency(torch, eng, graph, dev, stream, side, calls=200):
    """Latency of ONE batch-1 call (network.py:148-156: one image per sess.run), image resident in HBM: wall clock from the call
    to the synchronised result, median / minimum over `calls` calls behind 20 untimed ones."""
    from roomnet_amd.synth import perf_batch
    im = torch.from_numpy(perf_batch(1, side, seed=3)).to(dev)
    p = torch.empty((1, graph.num_classes), dtype=torch.float32, device=dev)
    i = torch.empty((1,), dtype=torch.int64, device=dev)
    ts = []
    with torch.cuda.stream(stream):
        for k in range(calls + 20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.forward_u8_device(im.data_ptr(), 1, p.data_ptr(), i.data_ptr())
            stream.synchronize()
            if k >= 20:
                ts.append(time.perf_counter() - t0)
        # the same calls back to back (device time per call without the host round trip)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            eng.forward_u8_device(im.data_ptr(), 1, p.data_ptr(), i.data_ptr())
        torch.cuda.synchronize()
        b2b = (time.perf_counter() - t0) / calls
    return {"config": "batch 1, %dx%d" % (side, side), "calls": calls, "median_ms": float(np.median(ts)) * 1e3, "min_ms": float(np.min(ts)) * 1e3,
            "back_to_back_ms_per_call": b2b * 1e3,
            "what": "one image per call (network.py:148-156), resident in HBM; wall clock call -> synchronised result"}


class StubEngine:
    """CPU stand-in for the GPU engine (tests only, `--stub-engine`): lets the world_size-2 gloo test drive THIS
    file's N > 1 code path -- process group, packed all-gather, barriers, max-over-ranks timing, rank-0 JSON."""

    def __init__(self, graph, batch):
        self.graph = graph
        self.n_stages = len(graph.stages)
        self.batch = batch

    def forward_into(self, ims, probs, ids):
        import torch
        x = ims.reshape(ims.shape[0], -1).to(torch.float32)
        logits = torch.stack([x[:, k::6].mean(1) for k in range(6)], 1) / 255.0
        p = torch.softmax(logits, dim=-1)
        probs.copy_(p)
        ids.copy_(p.argmax(-1))

    def launch_groups(self):
        return [[i] for i in range(self.n_stages)]


def _rendezvous(world):
    """torch.distributed.run exports MASTER_ADDR / MASTER_PORT; a lone process (--force-collective without a launcher)
    rendezvouses with itself on a free local port."""
    if "MASTER_PORT" in os.environ:
        return {}
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return {"init_method": "tcp://127.0.0.1:%d" % port}


def self_launch(n, argv):
    """`python3 bench.py --gpus N` without a launcher (no WORLD_SIZE / RANK in the environment): start the N ranks as
    child processes of THIS file -- before torch is imported or anything touches HIP in this process, and never by exec --
    one per GPU, rendezvous on 127.0.0.1; rank 0 writes its JSON line straight to our stdout, the other ranks' stdout goes
    to stderr.  Returns the exit code (first failing rank's; the others are then ended by PID)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RN_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            try:
                code = p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:                     # a rank died: the others would wait in a collective forever
                    q.terminate()
    return rc


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
    ap.add_argument("--event-steps", type=int, default=30, help="iterations of the per-step hipEvent timing pass (median reported)")
    ap.add_argument("--pair32", action="store_true", help="fused stage pair on the round-2 32x32x16 kernel (RN_FLAG_PAIR_32X32: comparison arm)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the golden check (timing experiments with garbage results)")
    ap.add_argument("--handles", type=int, default=1, choices=(1, 2),
                    help="engine handles (each with its own activation tensors and stream) that take the steps in turn: with 2, the "
                         "small launches at the end of step k (stages 6, 7, tail: one latency-bound workgroup per image) overlap the "
                         "first launch of step k + 1.  Every step is still one complete pass over one resident batch")
    ap.add_argument("--spinup-steps", type=int, default=0,
                    help="untimed passes BEFORE the W warm-up steps (~0.3 s at batch 256, 224 x 224): the engine clock needs tens of ms "
                         "of load to leave its idle state, and W = 5 steps are 7 ms.  A fixed count, the same on every rank (a step "
                         "holds a collective when N > 1).  Reported as `spinup_steps`; 0 = none")
    ap.add_argument("--no-cold-pass", dest="cold_pass", action="store_false",
                    help="skip the cold / contract pass (W warm-up + K timed steps on one handle, no spin-up) that is timed BEFORE the "
                         "spin-up and reported as `cold_images_per_sec`")
    ap.add_argument("--stage-launches", action="store_true",
                    help="one launch per conv stage (RN_FLAG_STAGE_LAUNCHES): the unfused comparison arm")
    ap.add_argument("--compute-frozen", action="store_true",
                    help="RN_FLAG_COMPUTE_FROZEN: convolve the channels rn_create proves constant too (the comparison arm: same bits)")
    ap.add_argument("--no-unfolded-arm", action="store_true", help="skip the RN_FLAG_COMPUTE_FROZEN comparison pass")
    ap.add_argument("--no-dither", action="store_true",
                    help="RN_FLAG_NO_DITHER: plain rounding of weights and stores (rounds 1-5; the comparison arm of the refined rounding)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip `other_configs` (64 x 600 x 600 fp16, 256 x 224 x 224 float32, batch-1 latency) of the default run")
    ap.add_argument("--stub-engine", action="store_true", help=argparse.SUPPRESS)    # CPU tensors + gloo (tests)
    ap.add_argument("--force-collective", action="store_true",
                    help="run the N > 1 code path (RCCL process group, per-step all-gather, barriers, gathered-block check) "
                         "with however many ranks there are, also one: exercises it on a one-GPU box")
    ap.add_argument("--pcie", action="store_true",
                    help="also time the host-buffer entry point (rn_forward_u8: H2D copy + forward + D2H copy); reported as "
                         "path.pcie_inclusive_images_per_sec, never as `value`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or args.force_collective         # the distributed code path (normally: more than one rank)
    if multi:
        # before anything initialises the HSA runtime: the host driver only supports dmabuf IPC
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch
    import torch.distributed as dist
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch N > 1 with torch.distributed.run)" % (args.gpus, world),
              file=sys.stderr)
    stub = args.stub_engine
    if stub:
        dev = torch.device("cpu")
        if multi:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, **_rendezvous(world))
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if multi:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev, **_rendezvous(world))

    from roomnet_amd.graph import build_graph
    from roomnet_amd.synth import perf_batch
    graph = build_graph(6, args.side)
    B = args.batch
    weights = None
    if stub:
        eng = StubEngine(graph, B)
    else:
        from roomnet_amd import _capi
        from roomnet_amd.tf_bundle import BundleReader
        weights = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
        if args.side != 224:
            # the shipped dense/kernel only fits 224: seeded synthetic first dense kernel
            weights = dict(weights)
            weights["dense/kernel"] = np.random.default_rng(600).uniform(
                -0.04, 0.04, (graph.flat_len, 32)).astype(np.float32)
        eng = _capi.Engine(graph, weights, device=local_rank, dtype=args.dtype, max_batch=B,
                           stage_launches=args.stage_launches, pair32=args.pair32, compute_frozen=args.compute_frozen, no_dither=args.no_dither)
    engs = [eng]
    if not stub and args.handles == 2:
        engs.append(_capi.Engine(graph, weights, device=local_rank, dtype=args.dtype, max_batch=B,
                                 stage_launches=args.stage_launches, pair32=args.pair32, compute_frozen=args.compute_frozen, no_dither=args.no_dither))

    ims = torch.from_numpy(perf_batch(B, args.side, seed=rank)).to(dev)
    # probs [B,6] fp32 and ids [B] int64 live in ONE byte buffer per rank, so the result exchange is a single
    # all-gather (32 bytes per image) instead of two latency-bound collectives
    from roomnet_amd.parallel import result_buffers, unpack_results
    # TWO result buffers (and gathered buffers): the all-gather of step k is asynchronous -- RCCL runs it on its own
    # stream behind the head kernel of step k -- and overlaps the kernels of step k+1, which write the other buffer; a
    # buffer is reused only after the gather that read it two steps earlier has finished (stream-ordered wait, no host
    # sync).  Every step still does one forward pass and one all-gather; all of it is inside the timed bracket.
    bufs = [result_buffers(B, graph.num_classes, dev) for _ in range(2)]
    combo, probs, ids = bufs[0]
    g_bufs = [torch.empty((world * combo.numel(),), dtype=torch.uint8, device=dev) for _ in range(2)] if multi else None
    works = [None, None]
    n_steps_done = [0]

    # ONE explicit stream for the library's kernels and the collective: torch's default stream is the null stream,
    # which the library's own (non-blocking) stream is not ordered against.
    # (two handles: handle k & 1, stream k & 1 and result buffer k & 1 belong together, so every ordering argument below
    #  holds per parity exactly as it does for one handle)
    if stub:
        stream = None
        streams = [None, None]

        def sync():
            drain()

        def forward(k=0):
            eng.forward_into(ims, bufs[k][1], bufs[k][2])
    else:
        stream = torch.cuda.Stream(dev)
        eng.set_stream(stream.cuda_stream)
        streams = [stream, stream]
        if len(engs) == 2:
            streams[1] = torch.cuda.Stream(dev)
            engs[1].set_stream(streams[1].cuda_stream)

        def sync():
            drain()
            torch.cuda.synchronize()

        def forward(k=0):
            engs[k % len(engs)].forward_u8_device(ims.data_ptr(), B, bufs[k][1].data_ptr(), bufs[k][2].data_ptr())

    def drain():
        for k in range(2):
            if works[k] is not None:
                works[k].wait()
                works[k] = None

    def step(serial=False):
        """One complete pass over the resident batch (+ the all-gather when N > 1).  Returns the slot (handle, stream,
        result buffer) it ran on.  `serial`: always slot 0 -- one handle, strictly serial steps (the cold / contract pass)."""
        k = 0 if serial else n_steps_done[0] & 1
        n_steps_done[0] += 1
        with (torch.cuda.stream(streams[k]) if streams[k] is not None else contextlib.nullcontext()):
            if multi and works[k] is not None:
                works[k].wait()                   # the gather that read buffer k two steps ago
            forward(k)
            if multi:
                works[k] = dist.all_gather_into_tensor(g_bufs[k], bufs[k][0], async_op=True)
        return k

    def on_stream():
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    # ---- parity gate: the handle that is about to be timed reproduces the golden results
    parity = {"checked": False, "reason": "--no-parity-check" if args.no_parity_check else "stub engine"}
    if not stub and not args.no_parity_check:
        def fwd_host(batch):
            t = torch.from_numpy(np.ascontiguousarray(batch)).to(dev)
            p = torch.empty((len(batch), graph.num_classes), dtype=torch.float32, device=dev)
            i = torch.empty((len(batch),), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            with on_stream():
                eng.forward_u8_device(t.data_ptr(), len(batch), p.data_ptr(), i.data_ptr())
            torch.cuda.synchronize()
            return i.cpu().numpy(), p.cpu().numpy()
        parity = check_parity(fwd_host, args.side, args.dtype, B)
        if len(engs) == 2:            # the second handle is timed too: same gate
            def fwd_host2(batch):
                t = torch.from_numpy(np.ascontiguousarray(batch)).to(dev)
                p = torch.empty((len(batch), graph.num_classes), dtype=torch.float32, device=dev)
                i = torch.empty((len(batch),), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                with torch.cuda.stream(streams[1]):
                    engs[1].forward_u8_device(t.data_ptr(), len(batch), p.data_ptr(), i.data_ptr())
                torch.cuda.synchronize()
                return i.cpu().numpy(), p.cpu().numpy()
            check_parity(fwd_host2, args.side, args.dtype, B)

    sync()
    # ---- cold / contract pass: exactly what the command line says and nothing else -- W warm-up steps, K timed steps, ONE
    # handle, strictly serial, no spin-up, on a chip that idled through the parity check.  Reported as `cold_images_per_sec`
    # next to `value` (which follows the spin-up and alternates two handles; both are complete passes).
    cold_elapsed = None
    if args.cold_pass:
        with on_stream():
            for _ in range(args.warmup):
                step(serial=True)
            sync()
            if multi:
                dist.barrier()
            sync()
            tc = time.perf_counter()
            for _ in range(args.steps):
                step(serial=True)
            sync()
            if multi:
                dist.barrier()
            sync()
            cold_elapsed = time.perf_counter() - tc
        if multi:
            tcold = torch.tensor([cold_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tcold, op=dist.ReduceOp.MAX)
            cold_elapsed = float(tcold.item())
    spinup_steps = 0
    with on_stream():
        if args.spinup_steps > 0 and not stub:
            # bring the chip to its sustained clock / power state: not part of W, not timed
            for _ in range(args.spinup_steps):
                step()
                spinup_steps += 1
            sync()
        for _ in range(args.warmup):
            step()
        sync()
        if multi:
            dist.barrier()
            # librccl prints a version banner through C stdio when its first communicator comes up; on a pipe that
            # sits in libc's buffer until exit and would land BEHIND the JSON line: push it out now
            import ctypes
            ctypes.CDLL(None).fflush(None)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        if multi:
            dist.barrier()
        sync()
        elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the comparison arm of the frozen-channel folding, in the same run: a handle that convolves the provably constant
    # channels too (RN_FLAG_COMPUTE_FROZEN), W warm-up + K timed steps, this rank only -> `folding.images_per_sec_computing_them`
    fold_info, unfolded_rate, unfolded_untimed = None, None, 0
    if not stub:
        fold_info = eng.frozen_info()
        fold_info["constant_channels"] = eng.const_info()
        folded = fold_info["pair_channels_not_convolved"] > 0 or fold_info["residual_stage_folded"] >= 0
        if folded and not args.no_unfolded_arm and rank == 0:
            e2 = _capi.Engine(graph, weights, device=local_rank, dtype=args.dtype, max_batch=B,
                              stage_launches=args.stage_launches, pair32=args.pair32, compute_frozen=True, no_dither=args.no_dither)
            e2.set_stream(stream.cuda_stream)
            # the conditions of `value`: the same untimed steps in front of the timed ones (the chip idled while the handle was built)
            unfolded_untimed = (args.warmup + args.steps if cold_elapsed is not None else 0) + spinup_steps + args.warmup
            with on_stream():
                for _ in range(unfolded_untimed):
                    e2.forward_u8_device(ims.data_ptr(), B, bufs[0][1].data_ptr(), bufs[0][2].data_ptr())
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(args.steps):
                    e2.forward_u8_device(ims.data_ptr(), B, bufs[0][1].data_ptr(), bufs[0][2].data_ptr())
                torch.cuda.synchronize()
                unfolded_rate = B * args.steps / (time.perf_counter() - t2)
            e2.close()
            with on_stream():
                forward(0)                       # (the sanity check below reads this handle's results)
            torch.cuda.synchronize()

    # sanity: outputs are a distribution and an argmax of it; every rank's gathered block carries that rank's results
    if not args.no_parity_check:
        for k in range(2 if n_steps_done[0] > 1 else 1):
            combo_k, probs_k, ids_k = bufs[k]
            p = probs_k.cpu().numpy()
            i = ids_k.cpu().numpy()
            assert np.allclose(p.sum(1), 1.0, atol=1e-4) and (p.argmax(1) == i).all()
            if multi:
                rows = g_bufs[k].view(world, combo_k.numel())
                assert torch.equal(rows[rank], combo_k), "rank %d: its own block of the all-gather differs from its results" % rank
                for r in range(world):
                    gi, gp = unpack_results(rows[r], B, graph.num_classes)
                    gp = gp.cpu().numpy()
                    assert np.allclose(gp.sum(1), 1.0, atol=1e-4) and (gp.argmax(1) == gi.cpu().numpy()).all(), \
                        "rank %d: block %d of the all-gather is not a result" % (rank, r)

    if stub and multi:
        # stub engine: every rank knows every rank's input (perf_batch(seed = rank)), so every block of the gathered buffer is
        # checked against what that rank must have produced -- on every rank (tests/test_bench_multi.py, world 8)
        from roomnet_amd.parallel import shard_counts
        assert shard_counts(world * B, world) == [B] * world
        for k in range(2 if n_steps_done[0] > 1 else 1):
            rows = g_bufs[k].view(world, bufs[k][0].numel())
            for r in range(world):
                exp_combo, exp_probs, exp_ids = result_buffers(B, graph.num_classes, dev)
                eng.forward_into(torch.from_numpy(perf_batch(B, args.side, seed=r)).to(dev), exp_probs, exp_ids)
                assert torch.equal(rows[r], exp_combo), "rank %d: block %d of the all-gather is not rank %d's result" % (rank, r, r)

    n_st = len(graph.stages)
    stage_ms = np.zeros(n_st)
    head_ms = total_ms = 0.0
    event_ms = []
    pcie_rate = pcie_pipe_rate = pcie_pinned_rate = None
    if not stub:
        # ---- per-stage device time (HIP events on the launch stream), separate from the timed region
        eng.set_profiling(True)
        nprof = max(1, args.profile_steps)
        with on_stream():
            for _ in range(nprof):
                forward()
                t = eng.timing()
                stage_ms += np.array(t["stage_ms"])
                head_ms += t["head_ms"]
                total_ms += t["total_ms"]
        eng.set_profiling(False)
        stage_ms /= nprof
        head_ms /= nprof
        total_ms /= nprof
        # ---- per-step device time: one event pair per step on the launch stream, median
        with on_stream():
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                   for _ in range(max(3, args.event_steps))]
            # each pair brackets ONE step on the stream that step runs on (two handles: the slots alternate and nothing orders
            # streams[1] against streams[0], so a pair on the wrong stream would bracket none of the step's kernels); a step
            # waits in its stream for the previous step of the SAME slot only, so with two handles a pair still sees its own
            # launches plus whatever of the other slot's step overlaps them
            for a_ev, b_ev in evs:
                k = n_steps_done[0] & 1 if len(engs) == 2 else 0
                a_ev.record(streams[k])
                step(serial=len(engs) == 1)
                b_ev.record(streams[k])
            sync()
        event_ms = [a_ev.elapsed_time(b_ev) for a_ev, b_ev in evs]
        if args.pcie and world == 1:
            host_ims = perf_batch(B, args.side, seed=rank)
            eng.forward_u8(host_ims)
            t1 = time.perf_counter()
            for _ in range(max(3, args.steps // 4)):
                eng.forward_u8(host_ims)
            pcie_rate = B * max(3, args.steps // 4) / (time.perf_counter() - t1)
            # two-slot pipeline (rn_submit_u8 / rn_collect): the upload of batch k+1 overlaps the kernels of batch k
            reps = max(4, args.steps // 4)
            eng.submit_u8(host_ims, 0)
            t1 = time.perf_counter()
            for k in range(reps):
                eng.submit_u8(host_ims, (k + 1) & 1)
                eng.collect(k & 1)
            pcie_pipe_rate = B * reps / (time.perf_counter() - t1)
            eng.collect(reps & 1)
            # the same pipeline out of page-locked buffers (rn_host_alloc, one per slot): the upload is a true asynchronous DMA
            pins = [_capi.PinnedArray(host_ims.shape, np.uint8) for _ in range(2)]
            for pa in pins:
                pa.array[...] = host_ims
            eng.submit_u8(pins[0].array, 0)
            t1 = time.perf_counter()
            for k in range(reps):
                eng.submit_u8(pins[(k + 1) & 1].array, (k + 1) & 1)
                eng.collect(k & 1)
            pcie_pinned_rate = B * reps / (time.perf_counter() - t1)
            eng.collect(reps & 1)
            for pa in pins:
                pa.close()

    if rank == 0:
        elem = 4 if args.dtype == "f32" else 2
        sbytes = stage_bytes_per_image(graph, elem)
        sflops = stage_flops_per_image(graph)
        groups = eng.launch_groups()                       # conv stages per launch, e.g. [[0], [1], [2, 3], [4], ...]
        group_ms = [float(stage_ms[g[-1]]) for g in groups]
        value = world * B * args.steps / elapsed
        bytes_per_img = graph.boundary_elements_per_image() * elem
        f32 = args.dtype == "f32"
        out = {
            "metric": "images/sec, 224x224 batch-256 RoomNet inference" if args.side == 224 and B == 256
                      else "images/sec, %dx%d batch-%d RoomNet inference" % (args.side, args.side, B),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "RoomNet forward (reference final_model weights), uint8 BGR %dx%dx3 in HBM -> "
                                   "probs+ids in HBM, batch %d per GPU, %s%s"
                                   % (args.side, args.side, B,
                                      "float32 storage and arithmetic: conv stages 1-6 one v_mfma_f32_32x32x2_f32 launch each, the rest one launch per graph node" if f32
                                      else "%s storage / fp32 accumulate" % args.dtype,
                                      ", RCCL all-gather of probs+ids" if multi else ""),
                       "images_per_gpu": B, "global_batch": world * B, "im_side": args.side,
                       "parallelism": "dp%d" % world},
            "parity": parity, "spinup_steps": spinup_steps, "handles": len(engs),
            "folding": None if fold_info is None else dict(
                fold_info, what="rn_create proves channels constant for EVERY input (BN gammas the reference's L2 regulariser drove to "
                                "1e-20 .. 1e-30: the fma that applies them returns its addend bit for bit) and does not convolve them; "
                                "`value` is this default handle; images_per_sec_computing_them = the same run on a handle that computes "
                                "every channel (RN_FLAG_COMPUTE_FROZEN), one rank, K timed steps behind the same number of untimed "
                                "steps `value` has in front of it (untimed_steps_before_computing_them); constant_channels = "
                                "rn_const_info: output channels whose 16-bit store is provably one number, written once at rn_create",
                images_per_sec_computing_them=unfolded_rate, untimed_steps_before_computing_them=unfolded_untimed),
            "untimed_steps_before_value": (args.warmup + args.steps if cold_elapsed is not None else 0) + spinup_steps + args.warmup,
        }
        if cold_elapsed is not None:
            out["cold_images_per_sec"] = world * B * args.steps / cold_elapsed
            out["cold"] = {"ms_per_step": cold_elapsed / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup, "handles": 1,
                           "spinup_steps": 0, "what": "the same W + K steps timed BEFORE the spin-up on one handle, strictly serial: "
                                                     "the figure of a strict reading of the command line"}
        if not stub:
            # ---- roofline: the WHOLE PATH against the roof the target is stated in (BASELINE.md section 3:
            # img/s x algorithmic bytes per image / (n_gpu x 8 TB/s); float32: img/s x flops per image / matrix-fp32 peak).
            # Per launch (`launches[]`, HIP events on the launch stream): `credited_frac` = algorithmic stage-boundary bytes of
            # every stage the launch computes / its time / 8 TB/s -- a fused launch is credited with tensors that never leave
            # LDS, so this figure is NOT a bandwidth and may pass 1; `physical_frac` = what the PMC counters say the launch
            # moved (profiles/*_hbm_traffic.json of this configuration) / its time / 8 TB/s; `mfma_frac` = algorithmic conv
            # flops (the reference's: every channel) / time / dense matrix peak of the dtype, `mfma_frac_executed` = without the
            # flops of channels the handle folds.
            dom = int(np.argmax(group_ms))
            mfma_peak = MFMA_PEAK_F32 if f32 else MFMA_PEAK_16
            # share of a stage's algorithmic flops the folded handle really issues (`mfma_frac_executed`): the first 32 -> 32
            # stage computes its live couts only, the next one contracts the live input channels (16-bit: five
            # two-tap K = 32 chunks for nine taps of 32 channels), the folded residual stage its live cout quarters
            share = [1.0] * n_st
            if fold_info is not None:
                p32 = [k for k in range(n_st - 1) if graph.stages[k].cin == 32 and graph.stages[k].cout == 32 and not graph.stages[k].residual]
                dead = fold_info["pair_channels_not_convolved"]
                if dead > 0 and p32:
                    # 16-bit: the producer computes one 16-cout half whatever the ring holds; the consumer contracts five two-tap
                    # chunks (16-channel ring) or three four-tap chunks (8-channel ring, round 6) instead of nine K = 32 taps
                    share[p32[0]] = (1.0 - dead / 32.0) if f32 else 0.5
                    share[p32[0] + 1] = (1.0 - dead / 32.0) if f32 else ((3 if dead >= 24 else 5) * 32) / (9 * 32.0)
                if fold_info["residual_stage_folded"] >= 0:
                    share[fold_info["residual_stage_folded"]] = fold_info["residual_stage_live_quarters"] / 4.0
                cc = fold_info.get("constant_channels") or {}
                if cc.get("stage", -1) >= 0:
                    # 16 constant couts not convolved; the stage behind contracts five K = 32 fragments per kernel row instead of six
                    share[cc["stage"]] = 1.0 - cc["channels_not_convolved"] / float(graph.stages[cc["stage"]].cout)
                    share[cc["stage"] + 1] *= 5.0 / 6.0
                    if cc["stage"] + 2 < n_st and graph.stages[cc["stage"] + 2].cin == 64:      # (its output's last 16 channels are constants too)
                        share[cc["stage"] + 2] *= 5.0 / 6.0
            launches = []
            for j, g in enumerate(groups):
                sec = max(group_ms[j], 1e-9) * 1e-3
                gb = sum(sbytes[k] for k in g) * B
                gf = sum(sflops[k] for k in g) * B
                tr = measured_traffic(g, B, args.side, args.dtype)
                s0 = graph.stages[g[0]]
                if f32:
                    kname = "stage_f32m_kernel / per-node kernels, stage %s" % "+".join(map(str, g))
                elif len(g) > 1 and g[0] == 0:
                    kname = "stage_rw_kernel (S0F), stages 0+1 fused (3->8->32 ch)"
                elif len(g) > 1 and g[-1] == n_st - 1:
                    kname = "%s, stages %s + dense head fused" % ("backend_kernel" if g[0] == 6 else "tail_kernel", "+".join(map(str, g)))
                elif len(g) > 1:
                    kname = "%s, stages %s fused (%d->%d ch x%d + residual)" % (
                        "stage23pc_kernel" if args.pair32 else "stage23x_kernel", "+".join(map(str, g)), s0.cin, s0.cout, len(g))
                else:
                    kname = "stage %d kernel (%d->%d ch%s)" % (g[0], s0.cin, s0.cout, " + residual" if s0.residual else "")
                launches.append({"kernel": kname, "stages": g, "ms": group_ms[j], "algorithmic_bytes": int(gb),
                                 "credited_frac": gb / sec / HBM_PEAK, "traffic": tr,
                                 "physical_frac": None if tr is None else tr / sec / HBM_PEAK,
                                 "mfma_frac": gf / sec / mfma_peak,
                                 "mfma_frac_executed": sum(sflops[k] * share[k] for k in g) * B / sec / mfma_peak})
            traffics = [l["traffic"] for l in launches]
            path_traffic = None if any(t is None for t in traffics) else int(sum(traffics))
            bpi = graph.boundary_elements_per_image() * elem
            if f32:
                ach = value * graph.flops_per_image() / world
                out["roofline"] = {"bound": "mfma", "achieved": ach / 1e12, "peak": MFMA_PEAK_F32 / 1e12, "unit": "TFLOP/s",
                                   "frac": ach / MFMA_PEAK_F32, "traffic": path_traffic,
                                   "scope": "whole path: images/sec x 4.4864 GFLOP per image (SURVEY 8d: the reference's flops, folded "
                                            "channels included) / matrix-fp32 peak per GPU; `executed_frac` counts the flops the handle issues",
                                   "executed_frac": value * sum(sflops[k] * share[k] for k in range(n_st)) / world / MFMA_PEAK_F32,
                                   "dominant": dom, "launches": launches}
            else:
                ach = value * bpi / world
                out["roofline"] = {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK, "traffic": path_traffic,
                                   "scope": "whole path (BASELINE.md section 3): images/sec x algorithmic stage-boundary bytes per image "
                                            "/ (n_gpu x 8 TB/s); `traffic` = physical HBM bytes of one step (PMC, all launches)",
                                   "algorithmic_bytes_per_step": int(bpi * B),
                                   "physical_frac": None if path_traffic is None else path_traffic / (elapsed / args.steps) / HBM_PEAK,
                                   "dominant": dom, "launches": launches}
                # the bound of THIS fusion structure: a step cannot take less than its physical HBM bytes at 8 TB/s, nor less than
                # the matrix flops the handle issues at the dense peak (never above 1; traffic from the committed PMC profile)
                issued = sum(sflops[k] * share[k] for k in range(n_st)) * B
                t_step = elapsed / args.steps
                t_hbm = None if path_traffic is None else path_traffic / HBM_PEAK
                t_mfma = issued / MFMA_PEAK_16
                out["roofline"]["bound_frac"] = max(t_mfma, t_hbm or 0.0) / t_step
                out["roofline"]["bound_terms"] = {"hbm_ms": None if t_hbm is None else t_hbm * 1e3, "mfma_ms": t_mfma * 1e3,
                                                  "issued_flops_per_step": issued, "step_ms": t_step * 1e3,
                                                  "what": "max(physical HBM bytes of a step / 8 TB/s, matrix flops issued / 2.5 PFLOP/s) / ms_per_step"}
            med = float(np.median(event_ms))
            out["path"] = {"algorithmic_bytes_per_image": int(bytes_per_img),
                           "hbm_frac": value * bytes_per_img / (world * HBM_PEAK),
                           "mfma_frac": value * graph.flops_per_image() / (world * (MFMA_PEAK_F32 if f32 else MFMA_PEAK_16)),
                           "launch_groups": groups, "launch_ms": group_ms,
                           "stage_ms": [float(x) for x in stage_ms], "head_ms": float(head_ms),
                           "forward_ms_events": float(total_ms),
                           "median_ms_event": med, "min_ms_event": float(np.min(event_ms)),
                           "images_per_sec_event_median": world * B / (med * 1e-3),
                           "launch_hbm_frac": [float(sum(sbytes[k] for k in g) * B / (max(group_ms[j], 1e-9) * 1e-3) / HBM_PEAK)
                                               for j, g in enumerate(groups)]}
            if pcie_rate is not None:
                out["path"]["pcie_inclusive_images_per_sec"] = pcie_rate
                out["path"]["pcie_pipelined_images_per_sec"] = pcie_pipe_rate
                out["path"]["pcie_pipelined_pinned_images_per_sec"] = pcie_pinned_rate
            headline = world == 1 and not multi and args.side == 224 and B == 256 and args.dtype == "bf16" and not (
                args.stage_launches or args.pair32 or args.compute_frozen or args.no_dither)
            if headline and not args.no_other_configs:
                # BASELINE's other single-GPU shards, same process, parity-gated (profiles/r6_*: the same numbers from own runs)
                # (a failure here -- a parity gate that does not hold, an allocation -- is reported in its entry; it must not take the
                #  headline line of the same run down with it)
                others = []
                try:
                    others.append(measure_latency(torch, eng, graph, dev, stream, args.side))
                except Exception as e:                                   # noqa: BLE001
                    others.append({"config": "batch 1, %dx%d" % (args.side, args.side), "error": "%s: %s" % (type(e).__name__, e)})
                w224 = BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
                for side_o, batch_o, dt_o in ((600, 64, "f16"), (224, 256, "f32")):
                    try:
                        others.append(measure_other_config(_capi, torch, lambda sd: build_graph(6, sd), w224, dev, side_o, batch_o, dt_o,
                                                           args.warmup, min(args.steps, 100)))
                    except Exception as e:                               # noqa: BLE001
                        others.append({"config": "%d x %dx%d %s" % (batch_o, side_o, side_o, dt_o), "error": "%s: %s" % (type(e).__name__, e)})
                out["other_configs"] = others
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(weights, args.side)
        print(json.dumps(out), flush=True)
    if not stub:
        for e in engs:
            e.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
