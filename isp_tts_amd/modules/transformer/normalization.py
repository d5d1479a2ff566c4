"""LayerNorm / AdaptiveLayerNorm on ispk_layernorm_* (tts/modules/transformer/normalization.py of the reference)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import runtime


class LayerNorm(nn.LayerNorm):
    """normalization.py:20-27: affine LayerNorm, eps 1e-5; `condition` is accepted and ignored like the reference."""

    def __init__(self, dim: int, bias: bool = True, eps: float = 1e-5):
        super().__init__(normalized_shape=dim, eps=eps)
        if not bias:
            self.bias = None

    def forward(self, x: Tensor, condition: Optional[Tensor] = None, row_mask: Optional[Tensor] = None,
                out_dtype: torch.dtype = torch.float32) -> Tensor:
        return runtime.layernorm(x, self.weight, self.bias, row_mask=row_mask, eps=self.eps, out_dtype=out_dtype)


class AdaptiveLayerNorm(nn.Module):
    """normalization.py:37-61: Linear(cond) * LN_noaffine(x) + Linear(cond); scale init (W=0, b=1), shift init 0.
    The two condition projections (32 -> D) run on ispk_linear_small_f32; scale/shift/mask are fused into the
    normalisation kernel.  A 2-D condition [B, C] conditions each batch item, [1, 1, C] is broadcast (:57)."""

    def __init__(self, dim: int, condition_dim: int, bias: bool = True, eps: float = 1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Linear(condition_dim, dim)
        nn.init.zeros_(self.weight.weight)
        nn.init.ones_(self.weight.bias)
        self.bias = None
        if bias:
            self.bias = nn.Linear(condition_dim, dim)
            nn.init.zeros_(self.bias.weight)
            nn.init.zeros_(self.bias.bias)

    def forward(self, x: Tensor, condition: Optional[Tensor] = None, row_mask: Optional[Tensor] = None,
                out_dtype: torch.dtype = torch.float32, scale_shift: Optional[tuple] = None) -> Tensor:
        """`scale_shift` = (scale [Bc,D], shift [Bc,D]) already projected from the condition (the Transformer projects
        the condition for ALL of its adaptive norms in one launch)."""
        if scale_shift is not None:
            rows_per_batch = x.numel() // (x.shape[0] * x.shape[-1])
            return runtime.layernorm(x, None, None, scale_shift[0], scale_shift[1], rows_per_batch, row_mask, self.eps,
                                     out_dtype)
        if condition is None:  # reference: weight 1, bias 0 -> plain non-affine LN
            return runtime.layernorm(x, None, None, row_mask=row_mask, eps=self.eps, out_dtype=out_dtype)
        cond = condition.reshape(-1, condition.shape[-1]).float().contiguous()
        if cond.shape[0] not in (1, x.shape[0]):
            raise ValueError(f"condition batch {cond.shape[0]} must be 1 or {x.shape[0]}")
        scale = runtime.linear_small(cond, self.weight.weight, self.weight.bias)
        shift = runtime.linear_small(cond, self.bias.weight, self.bias.bias) if self.bias is not None else None
        rows_per_batch = x.numel() // (x.shape[0] * x.shape[-1])
        return runtime.layernorm(x, None, None, scale, shift, rows_per_batch, row_mask, self.eps, out_dtype)

    def extra_repr(self) -> str:
        return f"bias={self.bias is not None}"
