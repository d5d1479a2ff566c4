"""Utterance sharding across GPUs and the one exchange of the forward path: gathering mel outputs.

Utterances are independent end to end (no cross-utterance op on the forward path), so data parallelism
needs no collective inside the model: weights are replicated (92.5 MB fp32), each rank (one process per GPU) runs its
shard, and the mel outputs are gathered once — a `torch.distributed` gather to one rank (or an all-gather when every
rank wants every utterance), which is RCCL over xGMI with the "nccl" backend on ROCm.  Message sizes are small
(B_local * 80 * M * 4 B: 10.5 MB per rank at 64 x 512 frames), so the exchange is one message of equal-sized, padded
blocks per rank rather than a ring of bucketed pieces.

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


def _cut_micro_batches(lens_sorted: Sequence[int], frame_budget: int, frame_multiple: int) -> list[tuple[int, int, int]]:
    """(start, stop, padded frames) of consecutive micro-batches over a decreasing length list: each holds as many
    utterances as fit `frame_budget` PADDED frames (n * M_pad, M_pad = its first = longest member)."""
    out, s = [], 0
    while s < len(lens_sorted):
        m_pad = (lens_sorted[s] + frame_multiple - 1) // frame_multiple * frame_multiple
        n = max(1, frame_budget // m_pad)
        out.append((s, min(s + n, len(lens_sorted)), m_pad))
        s += n
    return out


def shard_by_length_range(mel_len: Sequence[int], world: int, quad: float = 1.0 / 512.0, frame_budget: int = 32768,
                          frame_multiple: int = 8) -> list[list[int]]:
    """Cost-balanced assignment that keeps NEIGHBOURS IN LENGTH on the same rank: utterances sorted by decreasing length
    are cut into `world` contiguous ranges.  Rank 0 gets a few long utterances, the last rank many short ones, and every
    rank pads to a maximum close to its own members - `shard_by_cost` deals the whole length range to every rank, which at
    8 ranks x 32 utterances of 128..1024 frames pads 1.8 x the valid frames.  The cuts minimise the LARGEST rank cost,
    where a rank's cost counts what it will really run: its micro-batches (`_cut_micro_batches`) at their padded size,
    n * M_pad * (1 + quad * M_pad).  Exact, by dynamic programming over the (few hundred) cut positions.
    Returns `world` lists of utterance indices, each sorted by decreasing length (empty lists when there are fewer
    utterances than ranks)."""
    lens = [int(v) for v in mel_len]
    order = sorted(range(len(lens)), key=lambda i: (-lens[i], i))
    srt = [lens[i] for i in order]
    n = len(srt)

    def cost(i: int, j: int) -> float:     # padded cost of the range [i, j)
        return sum((b - a) * m * (1.0 + quad * m) for a, b, m in _cut_micro_batches(srt[i:j], frame_budget, frame_multiple))

    parts = min(world, n)
    inf = float("inf")
    best = [[inf] * (n + 1) for _ in range(parts + 1)]
    cut = [[0] * (n + 1) for _ in range(parts + 1)]
    best[0][0] = 0.0
    for k in range(1, parts + 1):
        for j in range(k, n - (parts - k) + 1):
            for i in range(k - 1, j):
                if best[k - 1][i] == inf:
                    continue
                c = max(best[k - 1][i], cost(i, j))
                if c < best[k][j]:
                    best[k][j], cut[k][j] = c, i
    bounds, j = [n], n
    for k in range(parts, 0, -1):
        j = cut[k][j]
        bounds.append(j)
    bounds.reverse()
    shards = [order[bounds[k]:bounds[k + 1]] for k in range(parts)]
    return shards + [[] for _ in range(world - parts)]


def plan_micro_batches(mel_len: Sequence[int], text_len: Sequence[int], world: int, frame_budget: int = 32768,
                       quad: float = 1.0 / 512.0, frame_multiple: int = 8, text_multiple: int = 4):
    """Variable-length batches (BASELINE config 4): `shard_by_length_range` gives every rank a contiguous length range;
    the rank then cuts its (length-sorted) shard into micro-batches whose PADDED size n * M_pad stays within
    `frame_budget` frames (32,768 = the 64 x 512 batch the kernels are tuned on), each padded only to ITS OWN longest
    member - few long utterances or many short ones per launch, never a long one padding out many short ones.
    Returns (shards, plans): plans[rank] = [(utterance indices, padded frames M, padded tokens L), ...]."""
    mel = [int(v) for v in mel_len]
    txt = [int(v) for v in text_len]
    shards = shard_by_length_range(mel, world, quad, frame_budget, frame_multiple)
    plans = []
    for idxs in shards:
        mbs = []
        for a, b, m_pad in _cut_micro_batches([mel[i] for i in idxs], frame_budget, frame_multiple):
            ii = idxs[a:b]
            l_pad = (max(txt[i] for i in ii) + text_multiple - 1) // text_multiple * text_multiple
            mbs.append((ii, m_pad, l_pad))
        plans.append(mbs)
    return shards, plans


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
    blocking exchange of 8 x 10.5 MB per step would add its full transfer time to every step).

    `dtype` = torch.bfloat16 sends the mel values rounded to bf16 (half the xGMI bytes: 5.2 instead of 10.5 MB per rank and
    step; what a bf16 compute path's mel is worth anyway), the lengths stay exact int64; the gathered mel comes back in
    that dtype.

    `root` = rank that collects (the "RCCL gather of mel outputs" of the north star: every other rank SENDS its 10.5 MB
    over its direct xGMI link to the root and receives nothing - one grouped send/recv, the root's seven links work in
    parallel), or None for an all-gather (every rank ends up with every utterance; 8 x the fabric traffic).  mel and
    dec_len travel as ONE packed message per step.

    `submit(mel, dec_len)` copies the step's outputs into one of two staging buffers (the model's output buffers are
    overwritten by the next step, e.g. by a HIP-graph replay) and starts the asynchronous exchange from it; it only
    blocks when that staging buffer's previous exchange (two batches back) is still in flight.  `wait()` drains the
    pipeline and returns the latest (mel [world, B, C, M], dec_len [world, B]) - on the root only when `root` is set
    (None elsewhere)."""

    def __init__(self, batch: int, channels: int, frames: int, device, dtype: torch.dtype = torch.float32,
                 group: Optional[dist.ProcessGroup] = None, root: Optional[int] = None):
        assert dtype in (torch.float32, torch.bfloat16), "the packed message carries int64 lengths behind fp32 / bf16 mel values"
        self.group, self.root = group, root
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.shape = (batch, channels, frames)
        per8 = 8 // torch.empty((), dtype=dtype).element_size()    # message elements per int64 length
        self.n_mel = batch * channels * frames                     # mel elements; the int64 lengths follow (8-byte aligned)
        assert self.n_mel % per8 == 0
        self.n_msg = self.n_mel + per8 * batch
        self.stage = [torch.empty((self.n_msg,), dtype=dtype, device=device) for _ in range(2)]
        self.receives = root is None or self.rank == root
        self.out = [torch.empty((self.world, self.n_msg), dtype=dtype, device=device) if self.receives else None
                    for _ in range(2)]
        self.work: list = [None, None]
        self.count = 0

    def submit(self, mel: Tensor, dec_len: Tensor) -> None:
        assert tuple(mel.shape) == self.shape, "MelGatherPipeline is for fixed-shape batches (pad to the agreed shape)"
        k = self.count & 1
        if self.work[k] is not None:          # this staging buffer is being read by the exchange of two batches ago
            self.work[k].wait()
        self.stage[k][: self.n_mel].view(self.shape).copy_(mel)
        self.stage[k][self.n_mel:].view(torch.int64).copy_(dec_len)
        if self.root is None:
            self.work[k] = dist.all_gather_into_tensor(self.out[k].view(-1), self.stage[k], group=self.group, async_op=True)
        else:
            rows = list(self.out[k].unbind(0)) if self.receives else None
            dst = self.root if self.group is None else dist.get_global_rank(self.group, self.root)
            self.work[k] = dist.gather(self.stage[k], rows, dst=dst, group=self.group, async_op=True)
        self.count += 1

    def wait(self):
        for w in self.work:
            if w is not None:
                w.wait()
        self.work = [None, None]
        if self.count == 0 or not self.receives:
            return None
        k = (self.count - 1) & 1
        b, c, m = self.shape
        out = self.out[k]
        return (out[:, : self.n_mel].reshape(self.world, b, c, m),
                out[:, self.n_mel:].contiguous().view(torch.int64).view(self.world, b))


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
