"""GPU: handles come and go without leaving device memory behind (a serving process creates one handle per model version and
batch geometry over its lifetime; the reference's tf.Session owns its memory the same way, network.py:89)."""
import numpy as np
import pytest

from roomnet_amd import _capi
from roomnet_amd.graph import build_graph

pytestmark = pytest.mark.gpu


def _free_bytes():
    # hipMemGetInfo of the HIP runtime libroomnet_hip.so itself is linked against (torch brings its own copy of the runtime: a
    # second one in this process finds no device)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


@pytest.mark.parametrize("dtype,kw", [("bf16", {}), ("f16", {"compute_frozen": True}), ("f32", {}), ("f32", {"taps": True})])
def test_create_forward_destroy_returns_the_device_memory(weights, parity_images, dtype, kw):
    g = build_graph(6, 224)
    ims = parity_images[:8]

    def cycle():
        e = _capi.Engine(g, weights, device=0, dtype=dtype, max_batch=8, **kw)
        try:
            ids, probs = e.forward_u8(ims)
            assert probs.shape == (8, 6)
            e.submit_u8(ims, 0)              # the two-slot host pipeline allocates its staging buffers lazily
            ids2, _ = e.collect(0)
            np.testing.assert_array_equal(ids2, ids)
        finally:
            e.close()
        return ids

    first = cycle()                          # (the first handle of a process also pays for code objects and the runtime's pools)
    cycle()
    base = _free_bytes()
    for _ in range(6):
        np.testing.assert_array_equal(cycle(), first)
    lost = base - _free_bytes()
    assert lost <= 8 << 20, "six create / forward / destroy cycles kept %.1f MB of device memory" % (lost / 1e6)


def test_distinct_handles_run_concurrently_from_two_host_threads(weights, parity_images):
    """include/roomnet_hip.h: "distinct handles are independent".  Three host threads, each with a handle of its own (bf16, f16 and
    float32: three kernel families on three streams), classify different batches at the same time; every result equals the one
    the same handle gives alone."""
    import threading
    g = build_graph(6, 224)
    specs = [("bf16", parity_images[0:16]), ("f16", parity_images[16:32]), ("f32", parity_images[32:48])]
    engines = [_capi.Engine(g, weights, device=0, dtype=d, max_batch=16) for d, _ in specs]
    try:
        alone = [e.forward_u8(ims) for e, (_, ims) in zip(engines, specs)]
        errors, start = [], threading.Barrier(len(specs))

        def work(k):
            try:
                e, ims = engines[k], specs[k][1]
                start.wait()
                for _ in range(40):
                    ids, probs = e.forward_u8(ims)
                    np.testing.assert_array_equal(ids, alone[k][0])
                    np.testing.assert_array_equal(probs, alone[k][1])
            except BaseException as exc:       # noqa: BLE001 (reported by the main thread)
                errors.append((k, repr(exc)))

        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(specs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        assert not any(t.is_alive() for t in threads)
    finally:
        for e in engines:
            e.close()
