"""Per-dtype cache of kernel-ready weight images (fused [Wq;Wkv], bf16 copies, packed W2, conv weights as GEMM weights).

A captured HIP graph (`graph.GraphedForward`) holds RAW POINTERS to these tensors, so two rules keep replays valid:
  * one slot PER DTYPE: running the other precision (bench.py's fp32 line on the model whose bf16 graph is captured)
    builds a second image and leaves the first alive;
  * replacing an existing image - a source parameter changed (`load()`, an optimizer step) - bumps `replacements()`;
    a graph compares it with the value at capture and refuses to replay against freed weights.
"""
from __future__ import annotations

from typing import Callable, Sequence

_replacements = 0


def replacements() -> int:
    """How many times an existing staged image has been rebuilt (process-wide)."""
    return _replacements


class StagedWeights:
    def __init__(self):
        self._slots: dict = {}

    def get(self, tag, params: Sequence, build: Callable[[], object]):
        """The image for `tag` (a dtype, or any hashable), rebuilt by `build()` when a parameter's storage, version or
        device differs from the ones it was built from."""
        global _replacements
        key = tuple((p.data_ptr(), p._version, p.device) for p in params)
        slot = self._slots.get(tag)
        if slot is None or slot[0] != key:
            if slot is not None:
                _replacements += 1
            slot = (key, build())
            self._slots[tag] = slot
        return slot[1]
