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
