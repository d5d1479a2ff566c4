"""`Attend`: the core attention call (tts/modules/transformer/attend.py of the reference).

The reference's `efficient_attn` (attend.py:49-122) expands the single K/V head to H heads, materialises the ALiBi
bias as a [B,H,N,N] fp32 tensor with masked keys filled, and calls SDPA.  Here one HIP kernel
(`ispk_alibi_mqa_attn_*`, csrc/attention.hip) takes q, the shared k/v, the H slopes and the key lengths.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import runtime


class AttentionIntermediates(NamedTuple):
    queries: Tensor
    keys: Tensor
    values: Tensor
    qk_similarities: Optional[Tensor] = None


class Attend(nn.Module):
    def __init__(self, *, dropout: float = 0., causal: bool = False, scale: Optional[float] = None):
        super().__init__()
        if causal:
            raise NotImplementedError("causal attention is unused by the acoustic-model recipes and not built")
        self.causal, self.scale, self.dropout = causal, scale, dropout
        # like the reference's efficient path (attend.py:115-120), the kernel always uses 1/sqrt(head_dim)

    def forward(self, q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor] = None,
                attn_bias: Optional[Tensor] = None, offset: int = 0, slopes: Optional[Tensor] = None):
        """q [B,H,N,64]; k, v [B,N,64] (one shared head); mask [B,1,1,N] bool, True = attend, must be a length
        (prefix) mask as every mask on this path is; `attn_bias` [H,N,N], if given instead of `slopes`, must be the
        symmetric ALiBi bias -slope_h*|i-j| (its only producer in the reference: attention.py:152) — only the slopes
        are read back from it."""
        if self.training and self.dropout > 0:
            raise NotImplementedError("attention dropout (training) is outside the forward-path scope")
        if k.ndim != 3 or v.ndim != 3 or q.shape[-1] != 64:
            raise NotImplementedError("only one shared K/V head with head_dim 64 is built (recipes: one_kv_head)")
        B, H, N, _ = q.shape
        if k.shape[1] != N:
            raise NotImplementedError("cross-attention / KV-cache lengths are not built")
        if slopes is None:
            if attn_bias is None:
                slopes = q.new_zeros(H, dtype=torch.float32)
            else:
                slopes = -attn_bias[:, 0, 1].float() if N > 1 else q.new_zeros(H, dtype=torch.float32)
        key_len = mask.reshape(B, -1, N)[:, 0].sum(-1) if mask is not None else None
        qm = q.transpose(1, 2).contiguous()  # [B,N,H,64]
        k, v = k.contiguous(), v.contiguous()
        out = runtime.alibi_mqa_attention_raw(qm, H * 64, k, v, 64, slopes, key_len, B, N, H)
        inter = AttentionIntermediates(queries=q.detach(), keys=k.detach(), values=v.detach())
        return out.view(B, N, H, 64).transpose(1, 2), inter
