"""world_size-2 'gloo' tests (CPU) of the data-parallel sharding + all-gather used for N > 1."""
import os
import tempfile

import numpy as np
import pytest

from roomnet_amd.parallel import shard_bounds, shard_counts

torch = pytest.importorskip("torch")


def test_shard_bounds_cover_the_batch_contiguously():
    for n in (0, 1, 7, 8, 255, 2048):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == shard_counts(n, w)
    assert shard_bounds(2048, 8, 3) == (768, 1024)
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _fake_forward(batch):
    """Deterministic stand-in for the GPU engine: depends only on each image's content."""
    b = np.asarray(batch, np.float64).reshape(len(batch), -1)
    logits = np.stack([b[:, k::6].mean(1) for k in range(6)], 1) / 255.0
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    return probs.argmax(1).astype(np.int64), probs


def _worker(rank, world, init_file, n, out_dir):
    import torch.distributed as dist
    from roomnet_amd.parallel import DataParallelRoomNet
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        ims = rng.integers(0, 256, (n, 16, 16, 3), dtype=np.uint8)

        class Model:
            num_classes = 6

            def infer(self, b):
                return _fake_forward(b)

        dp = DataParallelRoomNet(Model())
        ids, probs = dp.infer(ims)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ids=ids, probs=probs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 7, 1])
def test_two_rank_gloo_all_gather_matches_single_process(n):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rendezvous")
        mp.spawn(_worker, args=(2, init_file, n, d), nprocs=2, join=True)
        rng = np.random.default_rng(123)
        ims = rng.integers(0, 256, (n, 16, 16, 3), dtype=np.uint8)
        ref_ids, ref_probs = _fake_forward(ims)
        for r in range(2):
            got = np.load(os.path.join(d, "rank%d.npz" % r))
            np.testing.assert_array_equal(got["ids"], ref_ids)        # every rank holds the whole result
            np.testing.assert_array_equal(got["probs"], ref_probs)


def test_group_plan_matches_shard_counts_for_2_to_8_devices():
    """rn_group_plan -- the shard / slot plan rn_group_forward_u8 applies, exported as a pure function -- against the Python side's
    shard_bounds for ndev 2..8: n < ndev (empty shards), ragged n, full batches; offsets contiguous; the slot size; the range error."""
    from roomnet_amd import _capi
    from roomnet_amd.parallel import shard_bounds, shard_counts
    for ndev in range(1, 9):
        for n in (0, 1, 2, ndev - 1, ndev, ndev + 1, 7, 63, 255, 256, 257, 2047, 2048, ndev * 256):
            if n < 0 or n > ndev * 256:
                continue
            counts, offsets, slot = _capi.group_plan(n, ndev, 256, 6)
            assert counts == shard_counts(n, ndev), (n, ndev)
            assert offsets == [shard_bounds(n, ndev, r)[0] for r in range(ndev)], (n, ndev)
            assert sum(counts) == n and max(counts) - min(counts) <= 1 and all(c <= 256 for c in counts)
            assert slot == 256 * (6 * 4 + 8)
    import pytest
    with pytest.raises(ValueError, match="out of range"):        # RN_E_RANGE
        _capi.group_plan(2049, 8, 256, 6)
    with pytest.raises(ValueError, match="bad argument"):        # RN_E_INVALID
        _capi.group_plan(8, 0, 256, 6)
