"""ALiBi slopes and the flow-matching time embedding (tts/modules/transformer/embeddings.py of the reference).

The reference materialises an integer bias tensor `-|i-j|` of shape [1,N,N], multiplies it by the per-head slopes
into [H,N,N] and expands it over the batch (embeddings.py:51-72, attend.py:90-100).  Here the bias never exists as a
tensor: the attention kernel computes `-slope_h * |i-j|` in registers.  These classes therefore only own the
(learned) slopes and keep the reference's parameter/buffer names.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import runtime


class ALiBiPositionalBias(nn.Module):
    def __init__(self, heads: int, total_heads: int, symmetric: bool = True):
        super().__init__()
        if not symmetric:
            raise NotImplementedError("asymmetric ALiBi is unused by the recipes and not built (embeddings.py:71-72)")
        self.heads, self.total_heads, self.symmetric = heads, total_heads, symmetric
        slopes = torch.tensor(self._compute_slopes(heads), dtype=torch.float32).view(-1, 1, 1)
        self.register_buffer("slopes", slopes, persistent=False)  # non-persistent like the reference (:35)

    @staticmethod
    def _compute_slopes(heads: int) -> list[float]:
        """embeddings.py:38-49: 2^(-8(i+1)/n) for n a power of two; otherwise the closest lower power of two
        followed by every other slope of the next power of two."""
        def pow2(n):
            start = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
            return [start * start ** i for i in range(n)]
        if math.log2(heads).is_integer():
            return pow2(heads)
        n = 2 ** math.floor(math.log2(heads))
        return pow2(n) + pow2(2 * n)[0::2][: heads - n]

    def get_slopes(self) -> Tensor:
        return self.slopes

    def get_bias(self, i: int, j: int, k: int = 0) -> Tensor:
        """Integer bias -|j - (i + k)| as the reference returns it (embeddings.py:51-54).  Only for callers that
        inspect `AttentionSharedIntermediates.rel_pos_bias`; the kernels never read it."""
        ia = torch.arange(k, i + k, dtype=torch.int, device=self.slopes.device)
        ja = torch.arange(j, dtype=torch.int, device=self.slopes.device)
        return -torch.abs(ja[None, None, :] - ia[None, :, None])

    def head_slopes(self) -> Tensor:
        """fp32 [total_heads] slopes for the kernel (heads without ALiBi get slope 0, embeddings.py:66-67)."""
        logs = getattr(self, "learned_logslopes", None)
        if logs is not None and logs.is_cuda and logs.dtype == torch.float32:     # one libispk launch (exp + zero padding)
            return runtime.exp_pad(logs, max(self.total_heads, logs.numel()))
        s = self.get_slopes().reshape(-1).to(torch.float32)
        if self.total_heads > s.numel():
            s = torch.cat([s, s.new_zeros(self.total_heads - s.numel())])
        return s.contiguous()


class LearnedALiBiPositionalBias(ALiBiPositionalBias):
    """embeddings.py:75-82: slopes = exp(learned_logslopes), initialised at log(default slopes)."""

    def __init__(self, heads: int, total_heads: int, symmetric: bool = True):
        super().__init__(heads, total_heads, symmetric)
        self.learned_logslopes = nn.Parameter(torch.log(self.slopes))

    def get_slopes(self) -> Tensor:
        return self.learned_logslopes.exp()


class SinusoidalEmbedding(nn.Module):
    """embeddings.py:85-128 (theta ** -(i/half) frequencies, `freq_scale` persistent buffer, optional raw position)."""

    def __init__(self, dim: int, theta: float = 10000, freq_scale: float = 1., with_positions: bool = False):
        super().__init__()
        assert dim % 2 == 0
        self.dim, self.theta, self.with_positions = dim, theta, with_positions
        self.register_buffer("freq_scale", torch.ones(1) * freq_scale, persistent=True)
        half = dim // 2
        self.register_buffer("inv_freq", theta ** -(torch.arange(half).float() / half), persistent=False)

    def forward(self, x: Tensor) -> Tensor:
        raise NotImplementedError("SinusoidalEmbedding only owns the `freq_scale` / `inv_freq` buffers here (the reference's "
                                  "state_dict names): the sinusoid is evaluated inside ispk_time_embedding_f32 "
                                  "(TimePositionalEmbedding.forward); there is no PyTorch path to fall back to")


class TimePositionalEmbedding(nn.Module):
    """embeddings.py:131-157: sinusoid -> Linear -> SiLU -> Linear, as one kernel."""

    def __init__(self, freq_dim: int = 256, emb_dim: int = 512, theta: float = 1000., freq_scale: float = 1000.,
                 with_steps: bool = False):
        super().__init__()
        self.freq_emb = SinusoidalEmbedding(freq_dim, theta=theta, freq_scale=freq_scale, with_positions=with_steps)
        self.mlp = nn.Sequential(nn.Linear(freq_dim + int(with_steps), emb_dim, bias=True), nn.SiLU(),
                                 nn.Linear(emb_dim, emb_dim, bias=True))

    def forward(self, x: Tensor) -> Tensor:
        """x [...] time values -> [..., emb_dim]; one kernel (sinusoid, Linear + SiLU, Linear: `ispk_time_embedding_f32`)."""
        fe = self.freq_emb
        if not fe.with_positions:
            raise NotImplementedError("built for the flow predictor's embedding (with_steps=True, temporal_adaptor.py:87-89)")
        return runtime.time_embedding(x, fe.inv_freq, fe.freq_scale, self.mlp[0].weight, self.mlp[0].bias,
                                      self.mlp[2].weight, self.mlp[2].bias)
