"""bench.py's own N > 1 code path on CPU: two ranks under torch.distributed.run with the gloo backend and the stub
engine (`--stub-engine`, CPU tensors) -- process-group set-up, the packed all-gather on every step, the barriers
around the timed region, the max-over-ranks time and the single JSON line of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_bench_two_ranks_gloo_stub_engine():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "4", "--stub-engine", "--no-cpu-baseline"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE JSON line, rank 1 nothing
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["vs_baseline"] is None


@pytest.mark.timeout(600)
def test_bench_eight_ranks_gloo_stub_engine_every_gathered_block():
    """World size 8 -- the node BASELINE config 4 names -- as the driver launches it (torch.distributed.run, one rank per GPU),
    on the stub engine: eight process groups members, the packed all-gather on every step, and on EVERY rank every one of the
    eight gathered blocks compared with what that block's rank must have produced from its own input (bench.py asserts it;
    a rank whose check fails exits non-zero and the launcher reports it)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
           "--batch", "5", "--stub-engine", "--no-cpu-baseline"]
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 40 and d["config"]["parallelism"] == "dp8" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 40 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


@pytest.mark.timeout(300)
def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python3 bench.py --gpus 2` as the driver types it -- no torch.distributed.run, no WORLD_SIZE: bench.py starts the
    two ranks itself (self_launch) and relays rank 0's single JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-engine", "--steps", "3", "--warmup", "1",
                        "--batch", "4", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert "all-gather" in d["config"]["workload"]


@pytest.mark.timeout(120)
def test_self_launch_reports_a_failing_rank():
    """A rank that dies takes the job down with its exit code instead of leaving the others in a collective."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-engine", "--steps", "1", "--warmup", "0",
                        "--batch", "0", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(120)
def test_group_bench_dry_run_prints_the_plan():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "group_bench.py"), "--gpus", "8", "--dry-run"], cwd=ROOT,
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["config"]["global_batch"] == 8 * 256
    assert d["shards"] == [[g * 256, (g + 1) * 256] for g in range(8)]


def test_bench_single_rank_stub_engine():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "3",
                        "--stub-engine", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 3 and d["value"] > 0


def test_bench_forced_collective_single_rank_gloo():
    """--force-collective: the distributed code path with one rank and no launcher (self-rendezvous)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "3",
                        "--stub-engine", "--no-cpu-baseline", "--force-collective"], cwd=ROOT, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "all-gather" in d["config"]["workload"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_rccl_path_on_one_gpu():
    """The N > 1 path on real hardware with the one GPU this pool's boxes have: RCCL process group of one rank, the
    all-gather enqueued behind the head kernel on the library's stream every step, barriers, the gathered-block check,
    and the parity gate in front of it all."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline",
                        "--force-collective", "--profile-steps", "1", "--event-steps", "3"], cwd=ROOT, capture_output=True,
                       text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["parity"]["checked"] and "all-gather" in d["config"]["workload"]
    assert d["value"] > 50000          # the collective must not serialise the pass (HBM-resident 150 k img/s without it)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_forced_collective_line_agrees_with_the_plain_line(record=None):
    """What the distributed code path costs on one GPU: the same command with and without it (one-rank RCCL group, the per-step
    all-gather enqueued behind the head kernel, barriers) on the same box, alternated A B A B, best of each arm.  The driver's
    N = 1 point of the scaling curve is the PLAIN line (`--gpus 1` builds no process group: it is BENCH's command); from N = 2 on
    every step carries the collective, measured here at 3-5 % of a 1.1 ms step (RCCL's enqueue + its copy kernel; the gather is
    asynchronous and double-buffered, so it is launch overhead on the stream, not a wait)."""
    def line(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline",
                            "--no-other-configs", "--no-unfolded-arm", "--profile-steps", "1", "--event-steps", "3"] + extra,
                           cwd=ROOT, capture_output=True, text=True, timeout=400)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    plain, coll = [], []
    for _ in range(2):
        plain.append(line([])["value"])
        coll.append(line(["--force-collective"])["value"])
    ratio = max(coll) / max(plain)
    print("plain %s forced-collective %s ratio %.4f" % (plain, coll, ratio))
    assert 0.92 <= ratio <= 1.02, (plain, coll)
