"""Data-parallel sharding of image batches across the GPUs of one node.

The reference is single-process / single-device (``network.py:89``: one ``tf.Session``).
Images are independent (inference-mode BN), so the MI355X scale-out is pure data
parallelism: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI),
rank g classifies the contiguous shard ``[g*N/W, (g+1)*N/W)`` and ONE collective -- an
all-gather of the ``[n, C]`` float32 probabilities and ``[n]`` int64 class ids (32 bytes per
image) -- gives every rank the full result.  There is no other exchange on the data path.

The same code runs on CPU tensors over the "gloo" backend (used by the world_size-2 tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of ``n`` items; the first ``n % world_size`` ranks get one extra."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank %d / world_size %d" % (rank, world_size))
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(n: int, world_size: int):
    return [shard_bounds(n, world_size, r)[1] - shard_bounds(n, world_size, r)[0] for r in range(world_size)]


def all_gather_outputs(ids, probs, n_total: int, group=None):
    """All-gather ragged per-rank ``(ids [n_r], probs [n_r, C])`` torch tensors into
    ``(ids [n_total], probs [n_total, C])`` on every rank (rank order = batch order)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = shard_counts(n_total, world)
    rank = dist.get_rank(group)
    if ids.shape[0] != counts[rank] or probs.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d rows, its shard has %d" % (rank, ids.shape[0], counts[rank]))
    cmax = max(counts) if counts else 0
    c = probs.shape[1]
    pad_p = torch.zeros((cmax, c), dtype=probs.dtype, device=probs.device)
    pad_i = torch.zeros((cmax,), dtype=ids.dtype, device=ids.device)
    pad_p[:counts[rank]] = probs
    pad_i[:counts[rank]] = ids
    out_p = torch.empty((world * cmax, c), dtype=probs.dtype, device=probs.device)
    out_i = torch.empty((world * cmax,), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(out_p, pad_p, group=group)
    dist.all_gather_into_tensor(out_i, pad_i, group=group)
    if all(k == cmax for k in counts):
        return out_i, out_p
    keep = torch.cat([torch.arange(r * cmax, r * cmax + counts[r], device=probs.device) for r in range(world)])
    return out_i[keep], out_p[keep]


class DataParallelRoomNet:
    """Wraps a model exposing ``infer(batch) -> (ids, probs)`` (``RoomNet`` in optimized mode).
    Every rank calls ``infer`` with the same full batch (or only needs its own shard to be
    valid); each classifies its shard and the outputs are all-gathered."""

    def __init__(self, model, group=None, forward: Optional[Callable] = None, device=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._forward = forward or model.infer
        self.device = device

    def infer(self, im_batch) -> Tuple[np.ndarray, np.ndarray]:
        import torch
        n = len(im_batch)
        lo, hi = shard_bounds(n, self.world_size, self.rank)
        if hi > lo:
            ids, probs = self._forward(im_batch[lo:hi])
        else:
            ncls = getattr(self.model, "num_classes", 6)
            ids, probs = np.zeros((0,), np.int64), np.zeros((0, ncls), np.float32)
        dev = self.device if self.device is not None else "cpu"
        t_ids = torch.as_tensor(np.ascontiguousarray(ids), dtype=torch.int64).to(dev)
        t_probs = torch.as_tensor(np.ascontiguousarray(probs), dtype=torch.float32).to(dev)
        g_ids, g_probs = all_gather_outputs(t_ids, t_probs, n, self.group)
        return g_ids.cpu().numpy(), g_probs.cpu().numpy()
