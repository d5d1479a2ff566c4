"""Monotonic Alignment Search entry points (tts/modules/aligner/mas.py:29-35 and cuda_mas.py:11-46 of the reference).

`b_mas` keeps the reference's numpy signature (host arrays in, int16 host array out) and `mas_device` is the
tensor-level call the Aligner uses.  Both run `ispk_mas_f32` (csrc/mas.hip) on the GPU; there is no CPU path here
(the CPU restatement lives in oracle/ and is test infrastructure).  Unlike the reference's CPU branch
(alignment.py:308) the input is never mutated.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

from ... import runtime


def mas_device(attn_logits: Tensor, text_len: Tensor, mel_len: Tensor, want_dur: bool = True, want_path: bool = False):
    """attn_logits fp32 [B, M, L] on the GPU; lengths int64 [B].  -> (attn_hard int16 [B,M,L], dur int64 [B,L], path)."""
    return runtime.mas(attn_logits, text_len, mel_len, want_dur=want_dur, want_path=want_path)


def b_mas(b_attn_map: np.ndarray, in_lens: np.ndarray, out_lens: np.ndarray) -> np.ndarray:
    """Reference signature: `b_mas(b_attn_map float32 [B,M,L], in_lens (text), out_lens (mel)) -> int16 [B,M,L]`."""
    if not torch.cuda.is_available():
        raise runtime.IspkError("b_mas runs on the GPU (ispk_mas_f32); no GPU is visible and there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.from_numpy(np.ascontiguousarray(b_attn_map, dtype=np.float32)).to(dev)
    tl = torch.from_numpy(np.ascontiguousarray(in_lens).astype(np.int64)).to(dev)
    ml = torch.from_numpy(np.ascontiguousarray(out_lens).astype(np.int64)).to(dev)
    hard, _, _ = runtime.mas(x, tl, ml, want_dur=False)
    return hard.cpu().numpy()
