"""Aligner: ConvAttention front-end + MAS (tts/models/acoustic/modules/alignment.py of the reference).

Both halves run on hand-written kernels: the MAS binarisation (`ispk_mas_f32`) and the convolutional
front-end that PRODUCES the MAS input (csrc/aligner.hip): padded channel-last activations so that every
Conv1d is one MFMA GEMM over overlapping rows, a fused masked instance norm, and one fused kernel for
QKᵀ + log-softmax + analytic diagonal prior + masked softmax, with the reference's exact order of operations
(Appendix A items 8-9).  The nn.Module layout below only exists to keep the reference's parameter names.
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import NamedTuple, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import runtime
from ..staging import StagedWeights
from ..modules.aligner import mas_device
from ..modules.constructor import Constructor


_NOT_A_PATH = ("{} only owns parameters (the reference's state_dict names); its arithmetic runs inside the fused HIP "
               "kernels that ConvAttention launches (csrc/aligner.hip) - there is no PyTorch/MIOpen path to fall back to")


class MaskedInstanceNorm1d(nn.Module):
    """modules/normalization.py:160-208 (`_masked_norm`, instance).  Parameter names weight / bias like
    nn.InstanceNorm1d; the normalisation itself is `ispk_masked_instnorm_f32` (statistics over valid positions only,
    biased variance, eps 1e-5, every position normalised, affine)."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))

    def forward(self, *args, **kwargs):
        raise NotImplementedError(_NOT_A_PATH.format("MaskedInstanceNorm1d"))


class ConvBlock1D(nn.Module):
    """alignment.py:40-83: x*mask -> Conv1d (no bias when normalised) -> activation -> masked norm -> dropout(eval: id).
    Holds `conv.weight` / `norm.{weight,bias}`; ConvAttention runs the block as one MFMA GEMM over overlapping rows of a
    padded channel-last buffer (GELU in the epilogue) plus one fused masked-instance-norm kernel."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 1, activation: str = "relu",
                 normalization: Optional[str] = "batch", bias: bool = True):
        super().__init__()
        if normalization not in (None, "instance"):
            raise NotImplementedError("only instance normalisation (the recipes' choice) is built")
        if activation not in ("gelu", "linear"):
            raise NotImplementedError("only gelu / linear activations (the recipes' choice) are built")
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, padding=(kernel_size - 1) // 2,
                              bias=bias and normalization is None)
        self.gelu = activation == "gelu"
        self.norm = MaskedInstanceNorm1d(out_channels) if normalization is not None else None

    def forward(self, *args, **kwargs):
        raise NotImplementedError(_NOT_A_PATH.format("ConvBlock1D"))


class ConvAttention(nn.Module, Constructor):
    def __init__(self, mel_dim: int, text_dim: int = 512, attention_dim: int = 80, key_kernel_size: int = 3,
                 query_kernel_size=(3, 3), dropout: float = 0.0, normalization: Optional[str] = "instance",
                 activation: str = "relu", attention_prior: bool = True):
        super().__init__()
        self.mel_dim, self.text_dim = mel_dim, text_dim
        self.scale = attention_dim ** -0.5
        if isinstance(query_kernel_size, int):
            query_kernel_size = [query_kernel_size] * 2
        self.key_proj = nn.ModuleList([
            ConvBlock1D(text_dim, text_dim * 2, key_kernel_size, activation, normalization, bias=False),
            ConvBlock1D(text_dim * 2, attention_dim, 1, "linear", None, bias=False)])
        self.query_proj = nn.ModuleList([
            ConvBlock1D(mel_dim, mel_dim * 2, query_kernel_size[0], activation, normalization, bias=False),
            ConvBlock1D(mel_dim * 2, mel_dim, query_kernel_size[1], activation, normalization, bias=False),
            ConvBlock1D(mel_dim, attention_dim, 1, "linear", None, bias=False)])
        self.attention_prior = attention_prior
        self.compute_dtype = torch.float32
        self._cache = StagedWeights()

    def _staged(self):
        """Conv weights as GEMM weights [O][k*C] (tap-major, matching the padded channel-last window)."""
        ps = [c.conv.weight for c in list(self.key_proj) + list(self.query_proj)]
        dt = self.compute_dtype

        def build():
            if dt == torch.float16:   # split fp16 planes [2, O, k*C]
                w2d = [runtime.split_f16(p.detach().permute(0, 2, 1).reshape(p.shape[0], -1).float().contiguous()) for p in ps]
            else:
                w2d = [p.detach().permute(0, 2, 1).reshape(p.shape[0], -1).to(dt).contiguous() for p in ps]
            return w2d[:2], w2d[2:]
        return self._cache.get(dt, ps, build)

    def project_queries(self, queries: Tensor, query_len: Tensor) -> Tensor:
        """The mel-side half of alignment.py:159-208 (query_proj: conv-GELU-norm, conv-GELU-norm, 1x1 conv).  It needs
        nothing from the text encoder, so the model runs it on a second stream beside the encoder stack."""
        _, wq = self._staged()
        gelu, dt = runtime.EP_GELU, self.compute_dtype
        if dt == torch.float16:   # split-fp16 path: conv operands as hi / lo planes, conv outputs and statistics fp32
            q = runtime.pad_rows(queries.float(), query_len, channel_first=True, out_dtype=dt)       # split planes out
            for i in (0, 1):
                q = runtime.conv5_padded_split(q, wq[i], gelu)
                q = runtime.masked_instnorm(q, self.query_proj[i].norm.weight, self.query_proj[i].norm.bias, query_len,
                                            out_dtype=dt)
            return runtime.conv5_padded_split(q, wq[2])
        q = runtime.pad_rows(queries.float(), query_len, channel_first=True, out_dtype=dt)
        q = runtime.conv5_padded(q, wq[0], gelu)
        q = runtime.masked_instnorm(q, self.query_proj[0].norm.weight, self.query_proj[0].norm.bias, query_len,
                                    out_dtype=dt)
        q = runtime.conv5_padded(q, wq[1], gelu)
        q = runtime.masked_instnorm(q, self.query_proj[1].norm.weight, self.query_proj[1].norm.bias, query_len,
                                    out_dtype=dt)
        return runtime.conv5_padded(q, wq[2])

    def forward(self, queries: Tensor, keys: Tensor, query_len: Tensor, key_len: Tensor, q_proj: Optional[Tensor] = None):
        """alignment.py:159-208.  queries (mel) [B, mel_dim, M] channel-first as the collator gives it; keys (encoder
        output) [B, text_dim, L] — a transposed view of the encoder's [B, L, text_dim] is read in place.
        -> (attn_soft, attn_logits), both [B, M, L] fp32.  11 launches: 2 pads, 5 conv GEMMs (GELU fused), 3 masked
        instance norms, 1 fused scores kernel.  q_proj: `project_queries(queries, query_len)` when already computed."""
        if not self.attention_prior:
            raise NotImplementedError("the aligner is built with its diagonal prior (recipes: attention_prior default)")
        wk, _ = self._staged()
        max_q, max_k = queries.shape[2], keys.shape[2]
        gelu, dt = runtime.EP_GELU, self.compute_dtype   # bf16 path: bf16 conv operands, fp32 conv outputs / statistics
        if dt == torch.float16:
            k = runtime.pad_rows(keys.float(), key_len, channel_first=True, out_dtype=dt)
            k = runtime.conv5_padded_split(k, wk[0], gelu)
            k = runtime.masked_instnorm(k, self.key_proj[0].norm.weight, self.key_proj[0].norm.bias, key_len, out_dtype=dt)
            k = runtime.conv5_padded_split(k, wk[1])
            q = q_proj if q_proj is not None else self.project_queries(queries, query_len)
            return runtime.aligner_scores(q, k, key_len, query_len, max_q, max_k)
        k = runtime.pad_rows(keys.float(), key_len, channel_first=True, out_dtype=dt)
        k = runtime.conv5_padded(k, wk[0], gelu)
        k = runtime.masked_instnorm(k, self.key_proj[0].norm.weight, self.key_proj[0].norm.bias, key_len, out_dtype=dt)
        k = runtime.conv5_padded(k, wk[1])
        q = q_proj if q_proj is not None else self.project_queries(queries, query_len)
        return runtime.aligner_scores(q, k, key_len, query_len, max_q, max_k, fast=dt == torch.bfloat16)


class AlignerOutput(NamedTuple):
    attn_soft: Tensor
    attn_logits: Tensor
    attn_hard: Tensor
    attn_hard_duration: Tensor


class Aligner(nn.Module, Constructor):
    def __init__(self, mel_dim: int, text_dim: int = 512, attention_dim: int = 80, key_kernel_size: int = 3,
                 query_kernel_size=(3, 3), dropout: float = 0.0, normalization: Optional[str] = "instance",
                 activation: str = "relu", attention_prior: bool = True):
        super().__init__()
        self.attention = ConvAttention(mel_dim=mel_dim, text_dim=text_dim, attention_dim=attention_dim,
                                       key_kernel_size=key_kernel_size, query_kernel_size=query_kernel_size,
                                       dropout=dropout, normalization=normalization, activation=activation,
                                       attention_prior=attention_prior)

    def forward(self, mel: Tensor, enc_text: Tensor, mel_len: Tensor, text_len: Tensor,
                q_proj: Optional[Tensor] = None, mas_stream=None) -> AlignerOutput:
        """alignment.py:259-289.  The duration fix-up of :278-282 (an item whose durations do not sum to mel_len gets the
        difference added to column 0) happens inside the MAS kernel, on the device and unconditionally - the reference
        tests `torch.all(...)` on the host first; adding a zero difference gives the same result without the round trip."""
        attn_soft, attn_logits = self.attention(mel, enc_text, mel_len, text_len, q_proj=q_proj)
        if mas_stream is not None and attn_logits.is_cuda:
            # `mas_stream`: MAS (one wavefront per utterance: 64 waves on a 1024-SIMD chip for 80 us) runs on that stream
            # beside whatever the caller launches next; the caller joins the stream before it reads attn_hard / durations
            mas_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(mas_stream):
                attn_hard, dur = self.binarize_attention_parallel(attn_logits, text_len, mel_len, return_duration=True)
            attn_logits.record_stream(mas_stream)
        else:
            attn_hard, dur = self.binarize_attention_parallel(attn_logits, text_len, mel_len, return_duration=True)
        return AlignerOutput(attn_soft=attn_soft, attn_logits=attn_logits, attn_hard=attn_hard, attn_hard_duration=dur)

    @torch.no_grad()
    def binarize_attention_parallel(self, attn_logits: Tensor, text_len: Tensor, mel_len: Tensor,
                                    return_duration: bool = False):
        """alignment.py:291-331: replaces both the numba-CPU and the numba-CUDA branch by the wavefront MAS kernel.
        Returns the int16 one-hot [B,M,L] on the logits' device (and the int64 column sums when asked)."""
        hard, dur, _ = mas_device(attn_logits.detach(), text_len, mel_len, want_dur=return_duration)
        return (hard, dur) if return_duration else hard
