"""The C ABI's multi-GPU entry (rn_group_*, include/roomnet_hip.h): one handle per device, contiguous shards, one RCCL
all-gather of the packed results.  The GPU boxes of this pool have ONE MI355X, so what runs here is the one-device
group (RCCL communicator of size 1, the same code path); more devices are unmeasured on hardware."""
import numpy as np
import pytest

from roomnet_amd import _capi
from roomnet_amd.graph import build_graph

pytestmark = pytest.mark.gpu


def test_one_device_group_matches_the_plain_handle(weights, parity_images):
    g = build_graph(6, 224)
    grp = _capi.Group(g, weights, devices=[0], dtype="bf16", max_batch_per_device=8)
    eng = _capi.Engine(g, weights, device=0, dtype="bf16", max_batch=8)
    try:
        for idx in ([3], list(range(5, 13)), list(range(0, 19))):          # 1, 8 (= capacity) and 19 (chunked) images
            ims = parity_images[idx]
            ids_g, probs_g = grp.forward_u8(ims)
            ids_e, probs_e = eng.forward_u8(ims)
            np.testing.assert_array_equal(probs_g, probs_e)
            np.testing.assert_array_equal(ids_g, ids_e)
    finally:
        grp.close()
        eng.close()


def test_group_host_entry_reuses_its_upload_thread_and_takes_pinned_buffers(weights, parity_images):
    """rn_group_forward_u8 hands the shards to the group's persistent per-device upload threads: many calls on one group (the
    thread is parked and woken, not re-created), pageable and page-locked sources, a second group alive beside the first."""
    g = build_graph(6, 224)
    grp = _capi.Group(g, weights, devices=[0], dtype="f16", max_batch_per_device=4)
    grp2 = _capi.Group(g, weights, devices=[0], dtype="f16", max_batch_per_device=4)
    eng = _capi.Engine(g, weights, device=0, dtype="f16", max_batch=4)
    pin = _capi.PinnedArray((4, 224, 224, 3), np.uint8)
    try:
        for rep in range(12):
            idx = [(rep * 5 + j) % len(parity_images) for j in range(1 + rep % 4)]
            ims = parity_images[idx]
            ids_e, probs_e = eng.forward_u8(ims)
            src = ims
            if rep % 2:
                pin.array[:len(idx)] = ims
                src = pin.array[:len(idx)]
            ids_g, probs_g = (grp if rep % 3 else grp2).forward_u8(src)
            np.testing.assert_array_equal(probs_g, probs_e)
            np.testing.assert_array_equal(ids_g, ids_e)
    finally:
        pin.close()
        grp.close()
        grp2.close()
        eng.close()


def test_group_argument_checks(weights):
    g = build_graph(6, 224)
    with pytest.raises(ValueError):
        _capi.Group(g, weights, devices=[0, 0], dtype="bf16", max_batch_per_device=2)      # a device listed twice
    with pytest.raises(ValueError):
        _capi.Group(g, weights, devices=[0], dtype="bf16", max_batch_per_device=0)
    grp = _capi.Group(g, weights, devices=[0], dtype="f16", max_batch_per_device=2)
    try:
        assert grp.lib.rn_group_size(grp._g) == 1
        buf = np.zeros((3, 224, 224, 3), np.uint8)
        probs, ids = np.zeros((3, 6), np.float32), np.zeros((3,), np.int64)
        rc = grp.lib.rn_group_forward_u8(grp._g, buf.ctypes.data, 3, probs.ctypes.data, ids.ctypes.data)    # > capacity
        assert rc < 0 and b"out of range" in grp.lib.rn_last_error()
    finally:
        grp.close()


def test_group_bench_tool_prints_the_bench_line_for_one_device():
    """tools/group_bench.py (ctypes only, no torch): the same JSON line as bench.py from the single-process group; on
    this pool's boxes that is the one-device group -- an 8-GPU node runs the same file with --gpus 8."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "group_bench.py"), "--gpus", "1", "--steps", "3",
                          "--warmup", "1", "--batch", "16"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["unit"] == "images/sec" and line["value"] > 0
    assert line["parity"]["checked"] and line["parity"]["ids_wrong"] == 0
    assert "torch" not in out.stderr.lower() or "import" not in out.stderr.lower()
