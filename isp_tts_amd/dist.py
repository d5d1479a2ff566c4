"""Utterance sharding across GPUs and the one exchange of the forward path: gathering mel outputs.

Utterances are independent end to end (no cross-utterance op on the forward path), so data parallelism
needs no collective inside the model: weights are replicated (92.5 MB fp32), each rank (one process per GPU) runs its
shard, and the mel outputs are gathered once — `torch.distributed` all-gather, which is RCCL over xGMI with the "nccl"
backend on ROCm.  Message sizes are small (B_local * 80 * M * 4 B: 10.5 MB per rank at 64 x 512 frames), so the
exchange is one all-gather of equal-sized, padded blocks rather than a ring of bucketed pieces.

Everything here is backend-agnostic (the CPU test suite runs it over gloo with world_size 2).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def shard_by_cost(mel_len: Sequence[int], world: int, quad: float = 1.0 / 512.0) -> list[list[int]]:
    """Length-balanced assignment of utterances to ranks: longest first, each to the currently lightest rank.
    Cost of an utterance = m * (1 + quad * m): the GEMM/LayerNorm work is linear in frames, attention is quadratic
    (at m = 512 they weigh the same with the default `quad`).  Returns `world` lists of utterance indices, each sorted
    by decreasing length (so a rank's padded length is its first item's)."""
    lens = [int(v) for v in mel_len]
    order = sorted(range(len(lens)), key=lambda i: (-lens[i], i))
    shards: list[list[int]] = [[] for _ in range(world)]
    load = [0.0] * world
    for i in order:
        r = min(range(world), key=lambda k: (load[k], len(shards[k]), k))
        shards[r].append(i)
        load[r] += lens[i] * (1.0 + quad * lens[i])
    return shards


def all_gather_mel(mel: Tensor, dec_len: Tensor, group: Optional[dist.ProcessGroup] = None,
                   max_frames: Optional[int] = None, max_batch: Optional[int] = None):
    """Gathers every rank's `mel [B_local, C, M_local]` and `dec_len [B_local]`.

    Ranks may hold different batch sizes and padded lengths: blocks are zero-padded to (max_batch, C, max_frames).
    Passing `max_frames` / `max_batch` (known to the caller for fixed-shape batches) skips the tiny MAX all-reduce that
    otherwise agrees on them.  Returns (mel [world, max_batch, C, max_frames], dec_len [world, max_batch] with -1
    marking padding rows)."""
    world = dist.get_world_size(group)
    B, C, M = mel.shape
    if max_frames is None or max_batch is None:
        dims = torch.tensor([B, M], dtype=torch.int64, device=mel.device)
        dist.all_reduce(dims, op=dist.ReduceOp.MAX, group=group)
        max_batch, max_frames = int(dims[0]), int(dims[1])
    if (B, M) != (max_batch, max_frames):
        padded = mel.new_zeros((max_batch, C, max_frames))
        padded[:B, :, :M] = mel
        mel = padded
        dl = dec_len.new_full((max_batch,), -1)
        dl[:B] = dec_len
        dec_len = dl
    mel = mel.contiguous()
    out = mel.new_empty((world * max_batch, C, max_frames))      # concatenated along dim 0 (every backend accepts it)
    lens = dec_len.new_empty((world * max_batch,))
    dist.all_gather_into_tensor(out, mel, group=group)
    dist.all_gather_into_tensor(lens, dec_len.contiguous(), group=group)
    return out.view(world, max_batch, C, max_frames), lens.view(world, max_batch)


class MelGatherPipeline:
    """The same exchange for a stream of fixed-shape batches, overlapped with compute: batch i's gather runs on the
    process group's communication stream while batch i+1 is being computed (xGMI and the CUs work at the same time; a
    blocking all-gather of 8 x 10.5 MB per step would add its full ring time to every step).

    `submit(mel, dec_len)` copies the step's outputs into one of two staging buffers (the model's output buffers are
    overwritten by the next step, e.g. by a HIP-graph replay) and starts an asynchronous all-gather from it; it only
    blocks when that staging buffer's previous gather (two batches back) is still in flight.  `wait()` drains the
    pipeline and returns the latest (mel [world, B, C, M], dec_len [world, B])."""

    def __init__(self, batch: int, channels: int, frames: int, device, dtype: torch.dtype = torch.float32,
                 group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.shape = (batch, channels, frames)
        self.stage = [torch.empty(self.shape, dtype=dtype, device=device) for _ in range(2)]
        self.stage_len = [torch.empty((batch,), dtype=torch.int64, device=device) for _ in range(2)]
        self.out = [torch.empty((self.world * batch, channels, frames), dtype=dtype, device=device) for _ in range(2)]
        self.out_len = [torch.empty((self.world * batch,), dtype=torch.int64, device=device) for _ in range(2)]
        self.work: list = [None, None]
        self.count = 0

    def submit(self, mel: Tensor, dec_len: Tensor) -> None:
        assert tuple(mel.shape) == self.shape, "MelGatherPipeline is for fixed-shape batches (pad to the agreed shape)"
        k = self.count & 1
        if self.work[k] is not None:          # this staging pair is being read by the gather of two batches ago
            for w in self.work[k]:
                w.wait()
        self.stage[k].copy_(mel)
        self.stage_len[k].copy_(dec_len)
        self.work[k] = [dist.all_gather_into_tensor(self.out[k], self.stage[k], group=self.group, async_op=True),
                        dist.all_gather_into_tensor(self.out_len[k], self.stage_len[k], group=self.group, async_op=True)]
        self.count += 1

    def wait(self):
        for ws in self.work:
            if ws is not None:
                for w in ws:
                    w.wait()
        self.work = [None, None]
        if self.count == 0:
            return None
        k = (self.count - 1) & 1
        b, c, m = self.shape
        return self.out[k].view(self.world, b, c, m), self.out_len[k].view(self.world, b)


def unshard(gathered: Tensor, lens: Tensor, shards: list[list[int]]):
    """Restores the original utterance order after `shard_by_cost` + `all_gather_mel`.
    -> (mel [N, C, max_frames], dec_len [N])."""
    n = sum(len(s) for s in shards)
    mel = gathered.new_zeros((n, gathered.shape[2], gathered.shape[3]))
    dec = lens.new_zeros((n,))
    for r, idxs in enumerate(shards):
        if idxs:
            ii = torch.as_tensor(idxs, device=gathered.device)
            mel[ii] = gathered[r, : len(idxs)]
            dec[ii] = lens[r, : len(idxs)]
    return mel, dec
