"""`Attention`: ALiBi multi-query attention block (tts/modules/transformer/attention.py:33-176 of the reference).

Same constructor arguments, forward signature, return triple and `state_dict` keys (`to_q.weight`, `to_kv.weight`,
`rel_pos.learned_logslopes`, `to_out.weight`; no biases).  Execution:
  1. ONE MFMA GEMM for to_q and to_kv together (D -> H*64 + 128), giving a [B,N,Q|K|V] buffer;
  2. `ispk_alibi_mqa_attn_*` directly on that buffer (K/V tile shared by all heads, bias/mask in registers);
  3. to_out as a GEMM whose epilogue applies `* mask` (attention.py:172) and, when the enclosing TransformerLayer
     asks for it, the residual add of transformer.py:91.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import NamedTuple, Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import runtime
from ...staging import StagedWeights
from ..constructor import Constructor, ModuleConfig
from .attend import Attend, AttentionIntermediates
from .embeddings import LearnedALiBiPositionalBias


class AttentionSharedIntermediates(NamedTuple):
    rel_pos_bias: Optional[Tensor] = None


@dataclass
class AttentionConfig(ModuleConfig):
    dim: int = 256
    heads: int = 4
    head_dim: Optional[int] = 64
    causal: bool = False
    dropout: float = 0.
    one_kv_head: bool = False
    context_dim: Optional[int] = None
    alibi_pos_bias: bool = False
    alibi_heads: Optional[int] = None
    alibi_symmetric: bool = True


class Attention(nn.Module, Constructor):
    def __init__(self, dim: int = 256, heads: int = 4, head_dim: Optional[int] = 64, causal: bool = False,
                 dropout: float = 0., one_kv_head: bool = False, context_dim: Optional[int] = None,
                 alibi_pos_bias: bool = False, alibi_heads: Optional[int] = None, alibi_symmetric: bool = True):
        super().__init__()
        head_dim = head_dim or dim // heads
        if head_dim != 64 or not one_kv_head or causal or context_dim not in (None, dim) or heads > 8:
            raise NotImplementedError(
                "built for the recipes' attention only: head_dim 64, one_kv_head, non-causal self-attention, <= 8 heads "
                f"(got head_dim={head_dim}, one_kv_head={one_kv_head}, causal={causal}, context_dim={context_dim}, "
                f"heads={heads})")
        self.heads, self.causal, self.dim, self.head_dim, self.one_kv_head = heads, causal, dim, head_dim, one_kv_head
        self.out_dim = self.q_dim = head_dim * heads
        self.kv_dim = head_dim
        self.to_q = nn.Linear(dim, self.q_dim, bias=False)
        self.to_kv = nn.Linear(dim, 2 * self.kv_dim, bias=False)
        self.scale = head_dim ** -0.5
        self.rel_pos = None
        if alibi_pos_bias:
            alibi_heads = heads if alibi_heads is None else alibi_heads
            assert alibi_heads <= heads, "number of ALiBi heads must be less than the total number of heads"
            self.rel_pos = LearnedALiBiPositionalBias(heads=alibi_heads, total_heads=heads, symmetric=alibi_symmetric)
        self.attend = Attend(causal=causal, dropout=dropout, scale=self.scale)
        self.to_out = nn.Linear(self.out_dim, dim, bias=False)
        self.compute_dtype = torch.float32
        self._cache = StagedWeights()

    # weights staged for the kernels (fused [to_q; to_kv], optional bf16 copies): one image per dtype, rebuilt when a
    # parameter changes (staging.StagedWeights)
    def _staged(self, dtype: torch.dtype):
        ps = (self.to_q.weight, self.to_kv.weight, self.to_out.weight) + \
             ((self.rel_pos.learned_logslopes,) if self.rel_pos is not None else ())

        def build():
            with torch.no_grad():
                if dtype == torch.float16:   # split fp16 planes [2, N, K] (hi, lo) for the split-fp16 kernels
                    wqkv = runtime.split_f16(torch.cat([self.to_q.weight, self.to_kv.weight], dim=0).float().contiguous())
                    wo = runtime.split_f16(self.to_out.weight.detach().float().contiguous())
                elif self.to_q.weight.is_cuda:   # libispk launches only: a training step re-stages after every update
                    wqkv = runtime.cat0([self.to_q.weight, self.to_kv.weight], dtype)
                    wo = runtime.cast_bf16(self.to_out.weight.detach()) if dtype == torch.bfloat16 else self.to_out.weight.detach()
                else:
                    wqkv = torch.cat([self.to_q.weight, self.to_kv.weight], dim=0).to(dtype).contiguous()
                    wo = self.to_out.weight.detach().to(dtype).contiguous()
                slopes = (self.rel_pos.head_slopes() if self.rel_pos is not None
                          else torch.zeros(self.heads, device=wo.device)).detach()
            return wqkv, wo, slopes
        return self._cache.get(dtype, ps, build)

    def _chunked_wo(self) -> Tensor:
        """to_out's weight as [out_dim / 32][dim][32] chunks (ispk_attn_out_ffn_bf16), staged once per weight version."""
        return self._cache.get("woc", (self.to_out.weight,),
                               lambda: runtime.ffn_chunk_w2(self._staged(torch.bfloat16)[1]))

    def _chunked_wqkv(self) -> Tensor:
        """[to_q; to_kv] as k-step chunks [dim / 16][heads * 64 + 128][16] (ispk_attn_out_ffn_qkv_bf16: the PREVIOUS layer's
        kernel applies this layer's attention_norm and q/kv projection), staged once per weight version."""
        return self._cache.get("wqc", (self.to_q.weight, self.to_kv.weight),
                               lambda: runtime.chunk_k16(self._staged(torch.bfloat16)[0]))

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, context: Optional[Tensor] = None,
                context_mask: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                cache: Optional[AttentionIntermediates] = None,
                shared_cache: Optional[AttentionSharedIntermediates] = None, *, key_len: Optional[Tensor] = None,
                residual: Optional[Tensor] = None, prenorm: Optional[tuple] = None, defer_out: bool = False,
                qkv: Optional[Tensor] = None):
        """x [B,N,dim] (fp32, or bf16 when compute_dtype is bf16); mask [B,N] bool, True = valid, a length mask.
        `key_len` (int64 [B]) may be passed to skip recomputing mask.sum(1); `residual` (fp32 [B,N,dim]) fuses
        `residual + mask * to_out(...)` into the output GEMM.  Returns (out, AttentionIntermediates,
        AttentionSharedIntermediates) like the reference; `rel_pos_bias` is None because no bias tensor exists.
        `prenorm` = (row_stats | None, weight, bias, eps): x is the fp32 input of the LayerNorm that precedes this block
        and the q/kv GEMM applies that LayerNorm while staging x (bf16 path; statistics from the producing kernel, or
        computed by the GEMM's own waves when None).  `defer_out` (bf16 path): the first element is the attention output
        BEFORE `to_out` (bf16 [B,N,heads*64]) - the caller's next kernel applies `to_out`, mask and residual itself
        (`runtime.attn_out_ffn`).  `qkv` (bf16 [B,N,heads*64+128]): the q/kv rows of norm(x), already produced by the previous
        layer's kernel - no projection here."""
        if context is not None or context_mask is not None or attention_mask is not None or cache is not None:
            raise NotImplementedError("cross-attention, explicit attention masks and KV caches are not on the "
                                      "acoustic-model forward path and are not built")
        b, n, _ = x.shape
        dt = self.compute_dtype
        wqkv, wo, slopes = self._staged(dt)
        if mask is not None and key_len is None:
            key_len = mask.sum(dim=1)
        if dt == torch.float16:   # split-fp16 path (x fp32, or already split planes [2, B, N, dim])
            if x.dtype == torch.float16:
                b, n = x.shape[1], x.shape[2]
                xs = x
            else:
                xs = runtime.split_f16(x.float().contiguous())
            qkv = runtime.gemm_split(xs, wqkv)
            o = runtime.alibi_mqa_attention_split(qkv, self.heads, slopes, key_len)
            out = runtime.gemm_split(o, wo, resid=residual, mask=mask, flags=runtime.EP_MASK_ACC if mask is not None else 0)
            hq = self.heads * 64
            inter = AttentionIntermediates(queries=qkv[..., :hq].view(b, n, self.heads, 64).transpose(1, 2),
                                           keys=qkv[..., hq:hq + 64], values=qkv[..., hq + 64:])
            return out, inter, AttentionSharedIntermediates(rel_pos_bias=None)
        if qkv is not None:
            assert dt == torch.bfloat16 and qkv.dtype == dt and qkv.shape == (b, n, self.heads * 64 + 128)
        elif prenorm is not None:
            assert dt == torch.bfloat16 and x.dtype == torch.float32
            qkv = runtime.gemm_lnin(x, prenorm[0], prenorm[1], prenorm[2], wqkv, ln_eps=prenorm[3])
        else:
            if x.dtype != dt:
                x = runtime.cast_bf16(x) if dt == torch.bfloat16 else x.float()
            qkv = runtime.gemm(x, wqkv)                                        # [B,N,H*64+128]
        o = runtime.alibi_mqa_attention(qkv, self.heads, slopes, key_len)      # [B,N,H*64]
        if defer_out:
            assert dt == torch.bfloat16
            out = o
        else:
            flags = runtime.EP_MASK_ACC if mask is not None else 0
            out = runtime.gemm(o, wo, resid=residual, mask=mask, flags=flags, out_dtype=torch.float32)
        hq = self.heads * 64
        inter = AttentionIntermediates(queries=qkv[..., :hq].view(b, n, self.heads, 64).transpose(1, 2),
                                       keys=qkv[..., hq:hq + 64], values=qkv[..., hq + 64:])
        return out, inter, AttentionSharedIntermediates(rel_pos_bias=None)
