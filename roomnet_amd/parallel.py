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


def result_buffers(n: int, num_classes: int, device):
    """One byte buffer holding ``probs [n, C]`` float32 followed by ``ids [n]`` int64, plus typed views of its two
    parts.  A forward pass writes through the views (or their ``data_ptr()``); the exchange moves the buffer."""
    import torch
    combo = torch.empty((n * (num_classes * 4 + 8),), dtype=torch.uint8, device=device)
    probs = combo[:n * num_classes * 4].view(torch.float32).view(n, num_classes)
    ids = combo[n * num_classes * 4:].view(torch.int64)
    return combo, probs, ids


def all_gather_packed(combo, group=None):
    """The data path's one collective: all-gather every rank's packed result buffer -> ``[world, bytes]``."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * combo.numel(),), dtype=torch.uint8, device=combo.device)   # flat: gloo insists on it
    dist.all_gather_into_tensor(out, combo, group=group)
    return out.view(world, combo.numel())


def unpack_results(packed_row, n: int, num_classes: int):
    """``(ids [n], probs [n, C])`` views of one rank's row of :func:`all_gather_packed`."""
    import torch
    probs = packed_row[:n * num_classes * 4].view(torch.float32).view(n, num_classes)
    ids = packed_row[n * num_classes * 4:n * (num_classes * 4 + 8)].view(torch.int64)
    return ids, probs


def all_gather_outputs(ids, probs, n_total: int, group=None):
    """All-gather ragged per-rank ``(ids [n_r], probs [n_r, C])`` torch tensors into
    ``(ids [n_total], probs [n_total, C])`` on every rank (rank order = batch order) with one collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = shard_counts(n_total, world)
    rank = dist.get_rank(group)
    if ids.shape[0] != counts[rank] or probs.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d rows, its shard has %d" % (rank, ids.shape[0], counts[rank]))
    cmax = max(counts) if counts else 0
    c = probs.shape[1]
    combo, pad_p, pad_i = result_buffers(cmax, c, probs.device)
    combo.zero_()
    pad_p[:counts[rank]] = probs.to(torch.float32)
    pad_i[:counts[rank]] = ids.to(torch.int64)
    packed = all_gather_packed(combo, group)
    parts = [unpack_results(packed[r], cmax, c) for r in range(world)]
    out_i = torch.cat([parts[r][0][:counts[r]] for r in range(world)])
    out_p = torch.cat([parts[r][1][:counts[r]] for r in range(world)])
    return out_i, out_p


class DataParallelRoomNet:
    """Wraps a model exposing ``infer(batch) -> (ids, probs)`` (``RoomNet`` in optimized mode).
    Every rank calls ``infer`` with the same full batch (or only needs its own shard to be
    valid); each classifies its shard and the outputs are all-gathered.

    With a real ``RoomNet`` the shard stays on the GPU: it is uploaded once, classified with
    ``rn_forward_u8_device`` straight into the packed result buffer and that buffer is what the collective
    moves -- library kernels and the all-gather run on one explicit stream, so they are stream-ordered without a
    host synchronisation.  Models without an engine (the gloo tests' stand-ins) go through ``infer`` on the host."""

    def __init__(self, model, group=None, forward: Optional[Callable] = None, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._forward = forward or model.infer
        self._engine = None
        self._stream = None
        if forward is None and hasattr(model, "_engine") and hasattr(model, "graph"):
            self._engine = model._engine()
            if device is None:
                device = torch.device("cuda", int(getattr(model, "device", 0)))
            self._stream = torch.cuda.Stream(device)
            self._engine.set_stream(self._stream.cuda_stream)
        self.device = device if device is not None else "cpu"

    def infer(self, im_batch) -> Tuple[np.ndarray, np.ndarray]:
        import torch
        n = len(im_batch)
        lo, hi = shard_bounds(n, self.world_size, self.rank)
        ncls = getattr(self.model, "num_classes", 6)
        if self._engine is not None:
            return self._infer_device(im_batch, n, lo, hi, ncls)
        return self._infer_host(im_batch, n, lo, hi, ncls)

    def _infer_host(self, im_batch, n, lo, hi, ncls):
        import torch
        if hi > lo:
            ids, probs = self._forward(im_batch[lo:hi])
        else:
            ids, probs = np.zeros((0,), np.int64), np.zeros((0, ncls), np.float32)
        t_ids = torch.as_tensor(np.ascontiguousarray(ids), dtype=torch.int64).to(self.device)
        t_probs = torch.as_tensor(np.ascontiguousarray(probs), dtype=torch.float32).to(self.device)
        g_ids, g_probs = all_gather_outputs(t_ids, t_probs, n, self.group)
        return g_ids.cpu().numpy(), g_probs.cpu().numpy()

    def _infer_device(self, im_batch, n, lo, hi, ncls):
        import torch
        eng = self._engine
        counts = shard_counts(n, self.world_size)
        cmax = max(counts) if counts else 0
        if cmax > eng.max_batch:
            raise ValueError("shard of %d images exceeds the engine's max_batch %d" % (cmax, eng.max_batch))
        # same feed rule as RoomNet.infer: uint8 (or integral values in [0, 255]) goes to the device path; a float feed
        # is refused by the 16-bit models with the reason, and takes the model's own infer() on a float32 model
        # (decided on the whole batch, so that every rank takes the same branch and the same collective)
        feed = np.asarray(im_batch)
        if feed.size and hasattr(self.model, "_as_feed"):
            feed = self.model._as_feed(feed)
        if feed.size and feed.dtype != np.uint8:
            return self._infer_host(im_batch, n, lo, hi, ncls)
        shard = feed[lo:hi].astype(np.uint8, copy=False)
        shard = np.ascontiguousarray(shard)
        with torch.cuda.stream(self._stream):
            combo, probs, ids = result_buffers(cmax, ncls, self.device)
            combo.zero_()
            if hi > lo:
                d_ims = torch.from_numpy(shard).to(self.device, non_blocking=False)
                eng.forward_u8_device(d_ims.data_ptr(), hi - lo, probs.data_ptr(), ids.data_ptr())
            packed = all_gather_packed(combo, self.group)
            parts = [unpack_results(packed[r], cmax, ncls) for r in range(self.world_size)]
            out_i = torch.cat([parts[r][0][:counts[r]] for r in range(self.world_size)])
            out_p = torch.cat([parts[r][1][:counts[r]] for r in range(self.world_size)])
            out_i, out_p = out_i.cpu(), out_p.cpu()
        return out_i.numpy(), out_p.numpy()
