"""GPU parity of the float32 HIP path (RN_DTYPE_F32, unfused per-node kernels) with
the oracle, through the C ABI.  BASELINE config 2: batch-1 / small-batch fp32 forward,
per-layer correctness.  Tolerances are BASELINE.md section 5: logits abs <= 1e-4,
probs abs <= 1e-5, ids identical where the top-2 margin > 1e-3; per-node max error
<= 1e-4 of the node's abs-max."""
import numpy as np
import pytest

from conftest import sample_positions, with_gammas
from oracle import c_oracle, roomnet_ref as R
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph

pytestmark = pytest.mark.gpu

TOL_LOGITS, TOL_PROBS, MARGIN = 1e-4, 1e-5, 1e-3


@pytest.fixture(scope="module")
def engine(weights):
    e = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="f32", max_batch=8, taps=True)
    yield e
    e.close()


def test_library_sees_a_gpu():
    assert _capi.device_count() >= 1


def test_logits_probs_ids_vs_golden(engine, parity_images, golden_parity, record):
    ids, probs, logits = [], [], []
    for i in range(0, len(parity_images), 8):               # 64 images = eight chunks of max_batch; logits tapped after each
        a, b = engine.forward_u8(parity_images[i:i + 8])
        ids.append(a)
        probs.append(b)
        logits.append(engine.tap("d3.relu", len(a)).copy())
    ids, probs, logits = np.concatenate(ids), np.concatenate(probs), np.concatenate(logits)
    record("parity_set_64_images_224", "f32_per_node_path", {
        "max_abs_dlogit_vs_fp64": float(np.abs(logits - golden_parity["logits_f64"]).max()), "tolerance_logits": TOL_LOGITS,
        "max_abs_dprob_vs_fp64": float(np.abs(probs - golden_parity["probs_f64"]).max()), "tolerance_probs": TOL_PROBS})
    np.testing.assert_allclose(logits, golden_parity["logits_f64"], atol=TOL_LOGITS, rtol=0)
    np.testing.assert_allclose(probs, golden_parity["probs_f64"], atol=TOL_PROBS, rtol=0)
    safe = golden_parity["top2_margin"] > MARGIN
    np.testing.assert_array_equal(ids[safe], golden_parity["ids"][safe])
    assert ids.dtype == np.int64 and probs.dtype == np.float32
    np.testing.assert_allclose(probs.sum(1), 1.0, atol=1e-5)


def test_every_node_vs_oracle_full_tensors(engine, weights, parity_images):
    idx = [14, 30]
    ims = parity_images[idx]
    ref = c_oracle.infer(weights, ims, taps=True)
    ids, probs = engine.forward_u8(ims)
    names = R.node_names()
    assert set(names) == set(engine.nodes())
    for name in names:
        got = engine.tap(name, len(idx))
        want = np.asarray(ref["taps"][name])
        assert got.shape == want.shape, name
        tol = 1e-4 * max(float(np.abs(want).max()), 1e-3)
        np.testing.assert_allclose(got, want, atol=tol, rtol=0, err_msg=name)
    np.testing.assert_array_equal(ids, ref["ids"])


def test_taps_vs_committed_golden(engine, parity_images, golden_taps):
    i = int(golden_taps["image_index"])
    engine.forward_u8(parity_images[i:i + 1])
    for name in R.node_names():
        v = engine.tap(name, 1)[0].ravel().astype(np.float64)
        absmax = float(golden_taps[name + "|absmax"])
        tol = 1e-4 * max(absmax, 1e-3)
        np.testing.assert_allclose(v[sample_positions(v.size)], golden_taps[name + "|samples"], atol=tol, rtol=0,
                                   err_msg=name)
        assert abs(v.mean() - float(golden_taps[name + "|mean"])) <= tol, name


def test_batch_one_equals_batched(engine, parity_images):
    ims = parity_images[[3, 17, 26]]
    ids_b, probs_b = engine.forward_u8(ims)
    for i in range(3):
        ids_1, probs_1 = engine.forward_u8(ims[i:i + 1])
        np.testing.assert_array_equal(probs_1[0], probs_b[i])      # bit-identical: no cross-image coupling
        assert ids_1[0] == ids_b[i]


def test_f32_entry_point_equals_u8_entry_point(engine, parity_images):
    ims = parity_images[[8, 21]]
    ids_u, probs_u = engine.forward_u8(ims)
    ids_f, probs_f = engine.forward_f32(R.preprocess_batch(ims))
    np.testing.assert_array_equal(probs_u, probs_f)
    np.testing.assert_array_equal(ids_u, ids_f)
    np.testing.assert_array_equal(engine.tap("input", 2), R.preprocess_batch(ims))


def test_device_resident_entry_point(engine, parity_images):
    ims = np.ascontiguousarray(parity_images[[0, 12, 38]])
    d_in = engine.device_malloc(ims.nbytes)
    d_probs = engine.device_malloc(3 * 6 * 4)
    d_ids = engine.device_malloc(3 * 8)
    try:
        engine.h2d(d_in, ims)
        engine.forward_u8_device(d_in, 3, d_probs, d_ids)
        engine.sync()
        probs = np.empty((3, 6), np.float32)
        ids = np.empty(3, np.int64)
        engine.d2h(probs, d_probs)
        engine.d2h(ids, d_ids)
    finally:
        for p in (d_in, d_probs, d_ids):
            engine.device_free(p)
    ids_h, probs_h = engine.forward_u8(ims)
    np.testing.assert_array_equal(probs, probs_h)
    np.testing.assert_array_equal(ids, ids_h)


def test_argmax_ties_resolve_to_lowest_index(engine, weights):
    # a solid black image lands on exact 0.0 ties among the non-max logits; make sure the
    # winner follows tf.argmax (lowest index) whenever the top logits tie exactly
    ims = np.zeros((1, 224, 224, 3), np.uint8)
    ids, probs = engine.forward_u8(ims)
    assert ids[0] == int(np.flatnonzero(probs[0] == probs[0].max())[0])


def test_error_conventions(engine, weights):
    with pytest.raises(ValueError):
        engine.forward_u8(np.zeros((1, 100, 100, 3), np.uint8))
    lib = engine.lib
    probs = np.empty((9, 6), np.float32)
    ids = np.empty(9, np.int64)
    ims = np.zeros((9, 224, 224, 3), np.uint8)
    rc = lib.rn_forward_u8(engine.handle, ims.ctypes.data, 9, probs.ctypes.data, ids.ctypes.data)
    assert rc == -5 and b"max_batch" in lib.rn_last_error()
    rc = lib.rn_forward_u8(engine.handle, None, 1, probs.ctypes.data, ids.ctypes.data)
    assert rc == -1
    with pytest.raises(ValueError):
        _capi.Engine(build_graph(6, 224), weights, device=99)
    with pytest.raises(ValueError):   # the 224 checkpoint does not fit a 600 graph (dense/kernel 64 vs 3136)
        _capi.Engine(build_graph(6, 600), weights, device=0)
    with pytest.raises(KeyError):
        bad = dict(weights)
        del bad["conv2d_3/kernel"]
        _capi.Engine(build_graph(6, 224), bad, device=0)


def test_timing_is_reported(engine, parity_images):
    engine.set_profiling(True)
    engine.forward_u8(parity_images[:4])
    t = engine.timing()
    engine.set_profiling(False)
    assert len(t["stage_ms"]) == 10 and all(x > 0 for x in t["stage_ms"]) and t["total_ms"] > 0
    assert engine.dominant_stage() == 4


def test_scratch_sharing_handle_matches_tap_handle(weights, parity_images, engine):
    e2 = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="f32", max_batch=4, taps=False)
    try:
        ims = parity_images[[5, 15, 25, 35]]
        ids2, probs2 = e2.forward_u8(ims)
        ids1, probs1 = engine.forward_u8(ims)
        # (the handle without taps shares its scratch buffers AND runs its conv stages on the matrix cores: same values up to the
        #  order of the convolution's K sum, test_matrix_core_f32_equals_per_node_path_closely)
        np.testing.assert_array_equal(ids1, ids2)
        np.testing.assert_allclose(probs1, probs2, atol=2e-6, rtol=0)
        a, b = e2.tap("s3.bn2", 4), engine.tap("s3.bn2", 4)
        assert float(np.abs(a - b).max()) <= 2e-5 * float(np.abs(b).max())
        with pytest.raises(_capi.RoomNetLibraryError):
            e2.tap("s3.conv", 4)
        # the first BN output of a residual stage that runs as one matrix-core launch is never written: an error, not
        # uninitialised memory (stage 9 runs per node on this handle: its s9.bn exists)
        for name in ("s3.bn", "s5.bn"):
            with pytest.raises(_capi.RoomNetLibraryError, match="never written"):
                e2.tap(name, 4)
        np.testing.assert_allclose(e2.tap("s9.bn", 4), engine.tap("s9.bn", 4), atol=2e-5 * float(np.abs(engine.tap("s9.bn", 4)).max()), rtol=0)
    finally:
        e2.close()


# ---- float32 conv stages on the matrix cores (rn_stage_f32m.hip): what an RN_DTYPE_F32 handle runs WITHOUT RN_FLAG_TAPS
# (the drop-in's default dtype; reference network.py:28).  Same tolerances as the per-node path above.
STAGE_OUT = ["s0.bn", "s1.bn", "s2.bn", "s3.bn2", "s4.bn", "s5.bn2", "s6.bn", "s7.bn", "s8.bn", "s9.bn2"]


@pytest.fixture(scope="module")
def engine_mm(weights):
    e = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="f32", max_batch=8)
    yield e
    e.close()


def test_matrix_core_f32_logits_probs_ids_vs_golden(engine_mm, parity_images, golden_parity, record):
    ids, probs, logits = [], [], []
    for i in range(0, len(parity_images), 8):
        a, b = engine_mm.forward_u8(parity_images[i:i + 8])
        ids.append(a)
        probs.append(b)
        logits.append(engine_mm.tap("d3.relu", len(a)).copy())
    ids, probs, logits = np.concatenate(ids), np.concatenate(probs), np.concatenate(logits)
    record("parity_set_64_images_224", "f32_matrix_core_path", {
        "max_abs_dlogit_vs_fp64": float(np.abs(logits - golden_parity["logits_f64"]).max()), "tolerance_logits": TOL_LOGITS,
        "max_abs_dprob_vs_fp64": float(np.abs(probs - golden_parity["probs_f64"]).max()), "tolerance_probs": TOL_PROBS,
        "frozen_info": engine_mm.frozen_info()})
    np.testing.assert_allclose(logits, golden_parity["logits_f64"], atol=TOL_LOGITS, rtol=0)
    np.testing.assert_allclose(probs, golden_parity["probs_f64"], atol=TOL_PROBS, rtol=0)
    safe = golden_parity["top2_margin"] > MARGIN
    np.testing.assert_array_equal(ids[safe], golden_parity["ids"][safe])


def test_matrix_core_f32_stage_outputs_vs_oracle(engine_mm, weights, parity_images):
    """Every stage output (the tensors the fused stage launches write) within 1e-4 of the node's abs-max of the oracle, on images
    that reach the 0- and the 6-clamp, at a batch that runs several bands per image (8 images on 256 CUs)."""
    idx = [1, 14, 22, 30, 35]
    ims = parity_images[idx]
    ref = c_oracle.infer(weights, ims, taps=True)
    ids, probs = engine_mm.forward_u8(ims)
    for name in STAGE_OUT:
        got = engine_mm.tap(name, len(idx))
        want = np.asarray(ref["taps"][name])
        assert got.shape == want.shape, name
        tol = 1e-4 * max(float(np.abs(want).max()), 1e-3)
        np.testing.assert_allclose(got, want, atol=tol, rtol=0, err_msg=name)
    np.testing.assert_array_equal(ids, ref["ids"])


def test_matrix_core_f32_equals_per_node_path_closely(engine, engine_mm, parity_images):
    """The two float32 paths differ only in the order of the convolution's K sum and of the pooling window sum."""
    ims = parity_images[[5, 19]]
    ids_a, probs_a = engine.forward_u8(ims)
    ids_b, probs_b = engine_mm.forward_u8(ims)
    np.testing.assert_array_equal(ids_a, ids_b)
    np.testing.assert_allclose(probs_a, probs_b, atol=2e-6, rtol=0)
    for name in STAGE_OUT:
        a, b = engine.tap(name, 2), engine_mm.tap(name, 2)
        assert float(np.abs(a - b).max()) <= 2e-5 * max(float(np.abs(a).max()), 1e-3), name
    # stage 0 runs as one fused launch there (same operations in the same order as its four per-node launches): same bits
    np.testing.assert_array_equal(engine.tap("s0.bn", 2), engine_mm.tap("s0.bn", 2))


def test_matrix_core_f32_band_decomposition_does_not_change_bits(weights, parity_images):
    """1 image (many bands per image) and 8 images (fewer bands) give the same bits: bands only re-partition rows."""
    e1 = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="f32", max_batch=1)
    e8 = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="f32", max_batch=8)
    try:
        ims = parity_images[[2, 9, 16, 23, 27, 31, 36, 39]]
        ids8, probs8 = e8.forward_u8(ims)
        for i in range(len(ims)):
            ids1, probs1 = e1.forward_u8(ims[i:i + 1])
            np.testing.assert_array_equal(probs1[0], probs8[i])
            assert ids1[0] == ids8[i]
    finally:
        e1.close()
        e8.close()


@pytest.mark.parametrize("side", [300, 600])
def test_matrix_core_f32_at_other_input_sides(weights, side):
    """The matrix-core float32 stages in column blocks / with other band counts (600 x 600 is the reference's default im_side,
    network.py:21): same numbers as the per-node path up to the order of the K sum and of the pooling window sum."""
    from roomnet_amd.synth import parity_batch
    g = build_graph(6, side)
    w = dict(weights)
    w["dense/kernel"] = R.synth_dense_kernel_600(g.flat_len)
    ims = parity_batch(side, seed=1)[[14, 22, 37]]
    mm = _capi.Engine(g, w, device=0, dtype="f32", max_batch=3)
    pn = _capi.Engine(g, w, device=0, dtype="f32", max_batch=3, taps=True)
    try:
        ids_a, probs_a = mm.forward_u8(ims)
        ids_b, probs_b = pn.forward_u8(ims)
        np.testing.assert_array_equal(ids_a, ids_b)
        np.testing.assert_allclose(probs_a, probs_b, atol=5e-6, rtol=0)
        for name in STAGE_OUT:
            a, b = mm.tap(name, 3), pn.tap(name, 3)
            assert float(np.abs(a - b).max()) <= 2e-5 * max(float(np.abs(b).max()), 1e-3), name
    finally:
        mm.close()
        pn.close()


def test_matrix_core_f32_folds_stage_5_frozen_channels(weights, parity_images):
    """Round 5: 44 of the 64 first-BN channels of stage 5 are frozen on the shipped checkpoint (y1 = ((x/16 - mean) * inv + beta) IS
    beta for every input: the reference's float32 computes the same expression); the default float32 handle relabels the
    stage's channels so that its second 32-cout tile is all frozen and does not convolve it (rn_create, rn_f32m_launch).  Likewise
    18 of stage 2's 32 output channels: relabelled to the end, stage 3 contracts the first 16 input channels only and starts its
    accumulators from the other 16's contribution (one constant per cout of a VALID convolution), and stage 2 convolves its 16
    live couts only (16 x 16 x 4 tiles) and writes the constants of the others.  Against the handle that computes
    everything (RN_FLAG_COMPUTE_FROZEN): the frozen channels hold the same bits, the live ones differ by the order of the K sum."""
    g = build_graph(6, 224)
    fold = _capi.Engine(g, weights, device=0, dtype="f32", max_batch=8)
    full = _capi.Engine(g, weights, device=0, dtype="f32", max_batch=8, compute_frozen=True)
    try:
        info = fold.frozen_info()
        assert info["residual_stage_folded"] == 5 and info["residual_stage_live_quarters"] == 2
        assert info["pair_channels_not_convolved"] == 16 and info["pair_channels_proven_frozen"] == 18
        assert full.frozen_info()["residual_stage_folded"] == -1 and full.frozen_info()["pair_channels_not_convolved"] == 0
        ims = parity_images[[3, 14, 22, 37, 44, 52, 56, 60]]
        ids_a, probs_a = fold.forward_u8(ims)
        ids_b, probs_b = full.forward_u8(ims)
        np.testing.assert_array_equal(fold.tap("s1.bn", 8), full.tap("s1.bn", 8))
        # s2.bn: relabelled in HBM, handed out in the reference's order; its frozen channels are the same constants on both handles
        # (the folding handle writes them without convolving), the live ones come from another matrix tile shape (K-sum order)
        s2a, s2b = fold.tap("s2.bn", 8), full.tap("s2.bn", 8)
        frozen = [c for c in range(32) if (s2b[..., c] == s2b[0, 0, 0, c]).all()]
        assert len(frozen) >= 18, frozen
        np.testing.assert_array_equal(s2a[..., frozen], s2b[..., frozen])
        assert float(np.abs(s2a - s2b).max()) <= 2e-5 * float(np.abs(s2b).max())
        for name in ("s3.bn2", "s4.bn", "s5.bn2", "s6.bn", "s9.bn2"):
            a, b = fold.tap(name, 8), full.tap(name, 8)
            assert float(np.abs(a - b).max()) <= 2e-5 * float(np.abs(b).max()), name
        np.testing.assert_allclose(probs_a, probs_b, rtol=0, atol=5e-6)          # (fp32 K-sum order; the oracle tolerance is 1e-5)
        np.testing.assert_array_equal(ids_a, ids_b)
    finally:
        fold.close()
        full.close()


@pytest.mark.parametrize("case", ["none", "nine_and_31", "other_sets", "everything"])
def test_matrix_core_f32_fold_on_other_checkpoints(weights, parity_images, case):
    """The float32 folds are a property of the CHECKPOINT too: frozen channels elsewhere, too few of them for a kernel variant
    (9 in stage 2: the relabelling happens, every channel is still contracted), or all of them (a stage needs live input
    channels) -- the handle folds what it can and agrees with the oracle on every stage output."""
    g = build_graph(6, 224)
    rng = np.random.default_rng(6)
    if case == "none":
        w, want = with_gammas(weights, [], [], 1), {"pair_channels_not_convolved": 0, "residual_stage_folded": -1}
    elif case == "nine_and_31":
        w, want = with_gammas(weights, rng.choice(32, 9, replace=False), rng.choice(64, 31, replace=False), 2), {"pair_channels_not_convolved": 0, "residual_stage_folded": -1}
    elif case == "other_sets":
        w, want = with_gammas(weights, rng.choice(32, 17, replace=False), rng.choice(64, 40, replace=False), 3), {
            "pair_channels_not_convolved": 16, "pair_channels_proven_frozen": 17, "residual_stage_folded": 5, "residual_stage_live_quarters": 2}
    else:
        w, want = with_gammas(weights, range(32), range(64), 4), {"pair_channels_not_convolved": 0, "residual_stage_folded": 5}
    ims = parity_images[[14, 30, 2, 52]]
    ref = c_oracle.infer(w, ims, taps=True)
    e = _capi.Engine(g, w, device=0, dtype="f32", max_batch=4)
    try:
        info = e.frozen_info()
        for k, v in want.items():
            assert info[k] == v, (case, info)
        ids, probs = e.forward_u8(ims)
        for name in ("s1.bn", "s2.bn", "s3.bn2", "s4.bn", "s5.bn2", "s6.bn", "s8.bn", "s9.bn2"):
            got, ref_t = e.tap(name, 4), np.asarray(ref["taps"][name])
            assert float(np.abs(got - ref_t).max()) <= 1e-4 * max(float(np.abs(ref_t).max()), 1e-3), (case, name)
        np.testing.assert_allclose(probs, ref["probs"], atol=2e-5, rtol=0)
    finally:
        e.close()
