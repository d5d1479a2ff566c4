"""Batch ingest for the forward path: collated host batches -> device tensors, and length bucketing.

The reference's collator (tts/data/collator.py:27-95) hands the Trainer a dict of pageable CPU tensors - `text_vector`
int64 [B, L], `text_vector_len` int64 [B], `mel` fp32 [B, 80, M], `mel_len` int64 [B], `pitch` / `energy` fp32 [B, M] -
which accelerate then moves to the device synchronously, batch by batch, on the compute stream
(experiments/trainer.py:538-545).  At MI355X speeds a 64 x 512-frame batch is 10.8 MB in and computes in 2.4 ms, so a
pageable, synchronous copy (~1.5 ms) would cost more than half the step.  Here:

  * `BatchIngest`: a ring of PINNED staging buffers and device buffers sized for the largest batch; `submit()` copies a
    collated batch into a pinned slot (host memcpy) and queues its host-to-device copies on a dedicated copy stream;
    `get()` makes the compute stream wait for that slot's event and returns device views - the copy of batch k+1 runs
    under the compute of batch k.  `AcousticModel.prepare_inputs` maps the field names (model.py:244-259).
  * `bucket_by_length`: the DataLoader-side counterpart of `dist.plan_micro_batches` - orders a pool of samples into
    batches of neighbouring lengths whose PADDED size stays within a frame budget, so the collator pads to a maximum
    close to every member.

PyTorch is plumbing here (pinned memory, streams, events); nothing in this file computes.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import Tensor

FIELDS = (("text_vector", torch.int64), ("text_vector_len", torch.int64), ("mel", torch.float32), ("mel_len", torch.int64),
          ("pitch", torch.float32), ("energy", torch.float32))


def _numel(shape) -> int:
    n = 1
    for v in shape:
        n *= int(v)
    return n


def bucket_by_length(mel_len: Sequence[int], frame_budget: int = 64 * 512, max_batch: Optional[int] = None) -> list[list[int]]:
    """Sample indices grouped into batches of neighbouring lengths (longest first): a batch takes as many samples as keep
    its padded size n * longest within `frame_budget` frames (and within `max_batch` samples).  Every index appears once."""
    order = sorted(range(len(mel_len)), key=lambda i: (-int(mel_len[i]), i))
    batches, s = [], 0
    while s < len(order):
        longest = max(1, int(mel_len[order[s]]))
        n = max(1, frame_budget // longest)
        if max_batch is not None:
            n = min(n, max_batch)
        batches.append(order[s:s + n])
        s += n
    return batches


class BatchIngest:
    """Double-buffered (by default) staging of collated batches.  Shapes may vary from batch to batch up to the maxima
    given here; the returned tensors are views of the slot's device buffers, valid until that slot is submitted again
    (`slots` submissions later)."""

    def __init__(self, device, max_batch: int, max_text: int, max_mel: int, mel_dim: int = 80, slots: int = 2):
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.slots = slots
        caps = {"text_vector": max_batch * max_text, "text_vector_len": max_batch, "mel": max_batch * mel_dim * max_mel,
                "mel_len": max_batch, "pitch": max_batch * max_mel, "energy": max_batch * max_mel}
        self.host = [{k: torch.empty(caps[k], dtype=dt, pin_memory=self.on_gpu) for k, dt in FIELDS} for _ in range(slots)]
        self.dev = [{k: torch.empty(caps[k], dtype=dt, device=self.device) for k, dt in FIELDS} for _ in range(slots)]
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.copied = [torch.cuda.Event() if self.on_gpu else None for _ in range(slots)]
        self.consumed = [torch.cuda.Event() if self.on_gpu else None for _ in range(slots)]
        self.shapes: list = [None] * slots
        self.head = 0          # next slot to submit into
        self.tail = 0          # next slot to hand out
        self.pending = 0
        self.handed_out = [False] * slots   # get() returned this slot's device views and done() has not been called for it

    def submit(self, batch: dict) -> None:
        """Stages one collated batch (the reference collator's dict; `pitch` / `energy` required, as in the recipes)."""
        assert self.pending < self.slots, "every slot holds a batch that has not been taken with get() yet"
        k = self.head
        # contract: get() -> queue the compute that reads the views -> done(); only then may the slot be staged again (the
        # `consumed` event that protects the device buffers is recorded by done() alone)
        assert not self.handed_out[k], ("BatchIngest: slot %d was handed out by get() and done() has not been called for it - "
                                        "its device buffers may still be read by queued work" % k)
        if self.on_gpu:
            # the HOST only waits for this pinned slot's own previous upload (long finished); the slot's DEVICE buffers are
            # protected on the copy stream below - waiting here for `consumed` would hold the host until the forward that is
            # running has finished, and the next batch's staging + launch would then start on an idle GPU
            self.copied[k].synchronize()
        shapes = {}
        for name, dt in FIELDS:
            t = batch[name]
            assert t.dtype == dt and t.device.type == "cpu", f"{name}: expected a CPU {dt} tensor"
            n = t.numel()
            assert n <= self.host[k][name].numel(), f"{name}: {tuple(t.shape)} exceeds the ingest capacity"
            self.host[k][name][:n].view(t.shape).copy_(t)          # pageable -> pinned (host memcpy)
            shapes[name] = tuple(t.shape)
        self.shapes[k] = shapes
        if self.on_gpu:
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.consumed[k])      # the compute that read this slot's device buffers (no-op at first)
                for name, _ in FIELDS:
                    n = _numel(shapes[name])
                    self.dev[k][name][:n].copy_(self.host[k][name][:n], non_blocking=True)
                self.copied[k].record(self.copy_stream)
        else:
            for name, _ in FIELDS:
                n = _numel(shapes[name])
                self.dev[k][name][:n].copy_(self.host[k][name][:n])
        self.head = (k + 1) % self.slots
        self.pending += 1

    def get(self) -> dict:
        """Device tensors of the oldest staged batch, ordered behind its copy on the CURRENT stream."""
        assert self.pending > 0, "nothing staged"
        k = self.tail
        if self.on_gpu:
            torch.cuda.current_stream().wait_event(self.copied[k])
        out = {name: self.dev[k][name][: _numel(self.shapes[k][name])].view(self.shapes[k][name]) for name, _ in FIELDS}
        self.tail = (k + 1) % self.slots
        self.pending -= 1
        self._last = k
        self.handed_out[k] = True
        return out

    def done(self) -> None:
        """Marks the batch last returned by get() as consumed by everything queued on the current stream so far; its slot
        may be overwritten by a later submit()."""
        if self.on_gpu:
            self.consumed[self._last].record()
        self.handed_out[self._last] = False


def model_inputs(batch: dict) -> dict:
    """The reference's `AcousticModel.prepare_inputs` (model.py:244-259): collator field names -> forward kwargs."""
    return {"text": batch["text_vector"], "text_len": batch["text_vector_len"], "mel": batch["mel"],
            "mel_len": batch["mel_len"], "pitch": batch["pitch"], "energy": batch["energy"]}
