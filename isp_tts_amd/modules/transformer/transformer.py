"""`TransformerLayer` / `Transformer` (tts/modules/transformer/transformer.py:37-211 of the reference).

Pre-norm layer, as launched here (7 kernels, activations make one HBM round trip between them):
    h  = LN/AdaLN(x)                                   ispk_layernorm
    qkv = h · [Wq;Wkv]ᵀ                                ispk_gemm
    o  = ALiBi-MQA(qkv)                                ispk_alibi_mqa_attn
    x1 = x + mask * (o · Woᵀ)                          ispk_gemm, epilogue MASK_ACC + residual   (transformer.py:91)
    h2 = mask * LN/AdaLN(x1)                           ispk_layernorm with row mask             (:97-102)
    f  = gelu(h2 · W1ᵀ)                                ispk_gemm, epilogue GELU
    y  = mask * (x1 + f · W2ᵀ)                         ispk_gemm, epilogue residual + MASK_OUT   (:105-110)
The residual stream (x, x1, y) is always fp32; with compute_dtype = bf16 the GEMM/attention operands are bf16.

bf16 path, decoder-sized batches (dim 384 = heads * 64, >= 16,385 rows): a layer is TWO launches -
    o  = ALiBi-MQA(qkv)                                ispk_alibi_mqa_attn_bf16      (qkv comes from the previous layer's kernel)
    y, qkv_next = ...                                  ispk_attn_out_ffn_qkv_bf16:   x1 = x + mask * (o · Woᵀ) in the accumulators,
                                                       y = mask * (x1 + FFN(LN(x1))), qkv_next = LN_next(y) · [Wq;Wkv]_nextᵀ
(first layer: q/kv by ispk_gemm_bf16_lnin; last layer: ispk_attn_out_ffn_bf16).  Small batches: LayerNorm folded into the split
feed-forward's combine pass, see `FeedForward.forward_prenorm_split`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import NamedTuple, Optional, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import runtime
from ...staging import StagedWeights
from ..constructor import Constructor, ModuleConfig
from .attend import AttentionIntermediates
from .attention import Attention, AttentionConfig, AttentionSharedIntermediates
from .feedforward import FeedForward, FeedForwardConfig
from .normalization import AdaptiveLayerNorm, LayerNorm


class TransformerLayerIntermediates(NamedTuple):
    attention: Optional[AttentionIntermediates] = None


class TransformerLayerOutput(NamedTuple):
    out: Tensor
    intermediates: Optional[TransformerLayerIntermediates] = None
    shared_intermediates: Optional[AttentionSharedIntermediates] = None
    next_normed: Optional[Tensor] = None   # LN_next(out) when the feed-forward kernel produced it (not in the reference)


@dataclass
class TransformerLayerConfig(ModuleConfig):
    dim: int = 384
    attention: Union[AttentionConfig, dict] = field(default_factory=AttentionConfig)
    feed_forward: Union[FeedForwardConfig, dict] = field(default_factory=FeedForwardConfig)
    pre_norm: bool = True
    adaptive_norm: bool = False
    condition_dim: Optional[int] = None


class TransformerLayer(nn.Module, Constructor):
    # Path switches are plain class attributes (set them on the class or on an instance; nothing reads the environment).
    # LayerNorm applied by the consuming GEMM's own waves (ispk_gemm_bf16_lnin with row_stats = NULL), decoder-sized batches:
    lnin_self = True
    # to_out + residual + feed_forward_norm + feed-forward + residual as ONE kernel (ispk_attn_out_ffn_bf16; x1 never reaches
    # memory), bf16 path, dim 384 = heads * 64, decoder-sized batches:
    proj_ffn = True
    final_norm_fused = True    # the stack's final LayerNorm from the last layer's fused kernel (ispk_attn_out_ffn_norm_bf16)
    # ... and in every workgroup of the split feed-forward (ispk_attn_out_ffn_split_bf16) - where that pays: with 2 splits (16,384
    # decoder rows = 32 utterances per GPU: 1.512 -> 1.498 ms per step); with 4 - 8 splits per row block the repeated projection
    # costs more than the to_out launch it saves (6,400 rows: 2.004 -> 2.012 ms; 8 utterances per GPU: 1.014 -> 1.066 ms)
    proj_ffn_split = True
    proj_ffn_split_min_rows = 8192

    def __init__(self, dim: int = 384, attention=None, feed_forward=None, pre_norm: bool = True,
                 adaptive_norm: bool = False, condition_dim: Optional[int] = None):
        super().__init__()
        if not pre_norm:
            raise NotImplementedError("post-norm layers are unused by the recipes and not built")
        assert not adaptive_norm or condition_dim is not None
        self.pre_norm, self.adaptive_norm = pre_norm, adaptive_norm
        # LayerNorm inside the consuming GEMM with statistics computed by the GEMM's own waves: pays from decoder-sized
        # batches on (33.0 vs 22.1 + 15.0 us at 32,768 rows); at 6,400 rows the output is split over many workgroups that
        # each repeat the fp32 staging, and the separate 4.9-us LayerNorm is cheaper (18.6 vs 9.4 + 4.9 us)
        self.lnin_self_min_rows = 128 * 128
        norm = (lambda: AdaptiveLayerNorm(dim, condition_dim=condition_dim)) if adaptive_norm else (lambda: LayerNorm(dim))
        self.attention_norm = norm()
        self.attention = Attention.init(attention if attention is not None else AttentionConfig(), dim=dim)
        self.feed_forward_norm = norm()
        self.feed_forward = FeedForward.init(feed_forward if feed_forward is not None else FeedForwardConfig(), dim=dim)

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, context: Optional[Tensor] = None,
                context_mask: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                adaptive_condition: Optional[Tensor] = None, cache: Optional[TransformerLayerIntermediates] = None,
                shared_cache: Optional[AttentionSharedIntermediates] = None, *, key_len: Optional[Tensor] = None,
                ada: Optional[tuple] = None, normed: Optional[Tensor] = None, next_norm: Optional[tuple] = None,
                skip_out: bool = False):
        """`normed`: attention_norm(x) when the previous layer's feed-forward kernel already produced it; `next_norm`
        = (weight, bias, eps, apply_mask, dtype) of the norm that will consume this layer's output - if the fused
        feed-forward kernel can emit it, the output carries it in `next_normed` (bf16 path, decoder-sized batches).
        `skip_out`: the caller consumes only `next_normed` (the last layer of a stack whose output goes through the final norm) -
        a kernel that produces the norm itself may then leave `out` unwritten (None)."""
        assert not self.adaptive_norm or adaptive_condition is not None or ada is not None, \
            "`adaptive_condition` should be provided for AdaptiveLayerNorm"
        if cache is not None:
            raise NotImplementedError("KV caches are not on the acoustic-model forward path")
        x = x.float().contiguous()
        cdt = self.attention.compute_dtype
        if mask is not None and key_len is None:
            key_len = mask.sum(dim=1)
        if cdt == torch.float16:
            if context is not None or attention_mask is not None:
                raise NotImplementedError("cross-attention / explicit attention masks are not on the forward path")
            return self._forward_split(x, mask, key_len, adaptive_condition, ada)
        kw1 = {"scale_shift": ada[0]} if ada is not None else {}
        kw2 = {"scale_shift": ada[1]} if ada is not None else {}
        handed = normed is not None and normed.dtype == torch.float32 and normed.shape[-1] == 2
        qkv_in = (normed is not None and cdt == torch.bfloat16 and normed.dtype == torch.bfloat16
                  and normed.shape[-1] == self.attention.heads * 64 + 128 != x.shape[-1])   # q/kv rows from the previous layer's kernel
        own = (normed is None and ada is None and isinstance(self.attention_norm, nn.LayerNorm) and context is None
               and attention_mask is None and x.shape[-1] in (256, 384) and self.attention_norm.weight is not None
               and self.attention_norm.bias is not None and self.lnin_self
               and x.numel() // x.shape[-1] >= self.lnin_self_min_rows)
        final = next_norm is not None and next_norm[4] != "stats"    # the stack's final norm: only the split path's combine serves it
        fuse = (cdt == torch.bfloat16 and self.proj_ffn and ada is None and context is None and attention_mask is None
                and self.attention.out_dim == x.shape[-1] and self.feed_forward.proj_ok(x, self.feed_forward_norm))
        # small batches: the same prologue in every workgroup of the split feed-forward (ispk_attn_out_ffn_split_bf16)
        split_ok = (ada is None and (next_norm is None or next_norm[4] == "stats" or next_norm[4] in (cdt, torch.float32))
                    and self.feed_forward.split_ok(x, self.feed_forward_norm))
        fuse_split = (not fuse and split_ok and cdt == torch.bfloat16 and self.proj_ffn_split and context is None
                      and attention_mask is None and self.attention.out_dim == x.shape[-1] == 384
                      and x.numel() // x.shape[-1] >= self.proj_ffn_split_min_rows)
        if qkv_in:
            x1, inter, shared = self.attention(x, mask=mask, key_len=key_len, residual=x, defer_out=fuse or fuse_split, qkv=normed)
        elif cdt == torch.bfloat16 and (handed or own):
            # attention_norm inside the q/kv GEMM, applied while it stages x: with the row statistics the previous layer's
            # feed-forward kernel handed over, or (first layer of a stack) computed by the GEMM's own waves
            an = self.attention_norm
            x1, inter, shared = self.attention(x, mask=mask, key_len=key_len, residual=x, defer_out=fuse or fuse_split,
                                               prenorm=(normed if handed else None, an.weight, an.bias, an.eps))
        else:
            h = normed if normed is not None else self.attention_norm(x, adaptive_condition, out_dtype=cdt, **kw1)
            x1, inter, shared = self.attention(h, mask=mask, context=context, context_mask=context_mask,
                                               attention_mask=attention_mask, key_len=key_len, residual=x, defer_out=fuse or fuse_split)
        hn = None
        if fuse:    # (x1 is the attention output before to_out)
            fin = final and self.final_norm_fused and next_norm[0] is not None and next_norm[1] is not None
            y, hn = self.feed_forward.forward_proj_prenorm(x, x1, self.attention._chunked_wo(), self.feed_forward_norm, mask=mask,
                                                           next_norm=next_norm if (fin or not final) else None,
                                                           want_out=not (fin and skip_out))
            return TransformerLayerOutput(out=y, intermediates=TransformerLayerIntermediates(attention=inter),
                                          shared_intermediates=shared, next_normed=hn)
        if ada is None and self.feed_forward.prenorm_ok(x1, self.feed_forward_norm):
            # feed_forward_norm inside the fused feed-forward kernel (its waves own whole rows); the `* mask` of :102
            # cannot reach a kept value because the same mask multiplies the block's output (:110)
            y, hn = self.feed_forward.forward_prenorm(x1, self.feed_forward_norm, mask=mask, next_norm=None if final else next_norm)
            return TransformerLayerOutput(out=y, intermediates=TransformerLayerIntermediates(attention=inter),
                                          shared_intermediates=shared, next_normed=hn)
        if split_ok:
            # small batches: feed_forward_norm + feed-forward split over the inner dimension, then ONE pass that adds the
            # partial products, the residual and the mask and already applies the norm that consumes the result
            nn_ = None if next_norm is None else (next_norm[0], next_norm[1], next_norm[2], next_norm[3],
                                                  torch.float32 if next_norm[4] == torch.float32 else cdt)
            if fuse_split:    # (x1 is the attention output before to_out)
                y, hn = self.feed_forward.forward_prenorm_split(x, self.feed_forward_norm, mask=mask, next_norm=nn_,
                                                                attn_proj=(x1, self.attention._chunked_wo()))
            else:
                y, hn = self.feed_forward.forward_prenorm_split(x1, self.feed_forward_norm, mask=mask, next_norm=nn_)
            return TransformerLayerOutput(out=y, intermediates=TransformerLayerIntermediates(attention=inter),
                                          shared_intermediates=shared, next_normed=hn)
        if ada is None and self.feed_forward.prenorm_unfused_ok(x1, self.feed_forward_norm):
            # small batches (two-GEMM feed-forward): feed_forward_norm inside the first Linear's GEMM
            y = self.feed_forward.forward_prenorm_unfused(x1, self.feed_forward_norm, mask=mask)
            return TransformerLayerOutput(out=y, intermediates=TransformerLayerIntermediates(attention=inter),
                                          shared_intermediates=shared, next_normed=None)
        h2 = self.feed_forward_norm(x1, adaptive_condition, row_mask=mask, out_dtype=cdt, **kw2)
        y = self.feed_forward(h2, residual=x1, mask=mask)
        return TransformerLayerOutput(out=y, intermediates=TransformerLayerIntermediates(attention=inter),
                                      shared_intermediates=shared, next_normed=hn)


    def _norm_split(self, norm, x: Tensor, condition: Optional[Tensor], scale_shift: Optional[tuple],
                    row_mask: Optional[Tensor]) -> Tensor:
        """LayerNorm / AdaptiveLayerNorm with the result as split fp16 planes (ispk_layernorm_f32_split)."""
        if isinstance(norm, AdaptiveLayerNorm):
            if scale_shift is None and condition is not None:
                cond = condition.reshape(-1, condition.shape[-1]).float().contiguous()
                scale_shift = (runtime.linear_small(cond, norm.weight.weight, norm.weight.bias),
                               runtime.linear_small(cond, norm.bias.weight, norm.bias.bias) if norm.bias is not None else None)
            if scale_shift is None:
                return runtime.layernorm_split(x, None, None, row_mask=row_mask, eps=norm.eps)
            rows_per_batch = x.numel() // (x.shape[0] * x.shape[-1])
            return runtime.layernorm_split(x, None, None, scale_shift[0], scale_shift[1], rows_per_batch, row_mask, norm.eps)
        return runtime.layernorm_split(x, norm.weight, norm.bias, row_mask=row_mask, eps=norm.eps)

    def _forward_split(self, x: Tensor, mask: Optional[Tensor], key_len: Optional[Tensor],
                       adaptive_condition: Optional[Tensor], ada: Optional[tuple]) -> "TransformerLayerOutput":
        """The layer on the split-fp16 kernels (csrc/split.hip: fp32-grade products, three fp16 MFMAs each) - the same seven
        launches as the exact-fp32 path, the operands of every product carried as hi / lo fp16 planes:
            h  = LN(x) -> planes;  qkv = h [Wq;Wkv]^T (fp32);  o = attention(qkv) -> planes
            x1 = x + mask * (o Wo^T);  h2 = mask * LN(x1) -> planes;  f = gelu(h2 W1^T) -> planes;  y = mask * (x1 + f W2^T)"""
        att, ff = self.attention, self.feed_forward
        wqkv, wo, slopes = att._staged(torch.float16)
        w1, w2 = ff._staged(torch.float16)
        b, n, _ = x.shape
        h = self._norm_split(self.attention_norm, x, adaptive_condition, None if ada is None else ada[0], None)
        qkv = runtime.gemm_split(h, wqkv)
        o = runtime.alibi_mqa_attention_split(qkv, att.heads, slopes, key_len)
        x1 = runtime.gemm_split(o, wo, resid=x, mask=mask, flags=runtime.EP_MASK_ACC if mask is not None else 0)
        h2 = self._norm_split(self.feed_forward_norm, x1, adaptive_condition, None if ada is None else ada[1], mask)
        f = runtime.gemm_split(h2, w1, bias=ff.net[0].bias, flags=ff.act_flag, out_split=True)
        y = runtime.gemm_split(f, w2, bias=ff.net[3].bias, resid=x1, mask=mask,
                               flags=runtime.EP_MASK_OUT if mask is not None else 0)
        hq = att.heads * 64
        inter = AttentionIntermediates(queries=qkv[..., :hq].view(b, n, att.heads, 64).transpose(1, 2),
                                       keys=qkv[..., hq:hq + 64], values=qkv[..., hq + 64:])
        return TransformerLayerOutput(out=y, intermediates=TransformerLayerIntermediates(attention=inter),
                                      shared_intermediates=AttentionSharedIntermediates(rel_pos_bias=None), next_normed=None)


class TransformerOutput(NamedTuple):
    out: Tensor
    intermediates: Optional[list] = None


@dataclass
class TransformerConfig(ModuleConfig):
    dim: int = 384
    depth: int = 6
    transformer_layer: Union[TransformerLayerConfig, dict] = field(default_factory=TransformerLayerConfig)
    emb_dim: Optional[int] = None
    use_abs_pos_emb: bool = True
    adaptive_norm: bool = False
    condition_dim: Optional[int] = None


class Transformer(nn.Module, Constructor):
    def __init__(self, dim: int = 384, depth: int = 6, transformer_layer=None, emb_dim: Optional[int] = None,
                 use_abs_pos_emb: bool = True, adaptive_norm: bool = False, condition_dim: Optional[int] = None):
        super().__init__()
        self.dim = dim
        self.emb_dim = emb_dim = emb_dim or dim
        self.adaptive_norm = adaptive_norm
        layer_cfg = transformer_layer if transformer_layer is not None else TransformerLayerConfig()
        self.layers = nn.ModuleList([
            TransformerLayer.init(layer_cfg, dim=dim, adaptive_norm=adaptive_norm, condition_dim=condition_dim)
            for _ in range(depth)])
        if use_abs_pos_emb and self.layers[0].attention.rel_pos is None:
            raise NotImplementedError("absolute sinusoidal position embeddings (no ALiBi) are unused by the recipes")
        self.pos_emb = None
        self.project_emb = nn.Linear(emb_dim, dim) if emb_dim != dim else nn.Identity()  # transformer.py:170
        self.norm = nn.LayerNorm(dim)                                                      # transformer.py:172
        self._ada_cache = StagedWeights()

    def _ada_all(self, condition: Tensor):
        """Projects the condition for every AdaptiveLayerNorm of the stack in ONE launch (2 norms x 2 Linears per layer
        would otherwise be 4*depth tiny launches): concatenated [4*depth*dim, cond] weight, sliced per norm."""
        norms = [n for layer in self.layers for n in (layer.attention_norm, layer.feed_forward_norm)]
        ps = [p for n in norms for p in (n.weight.weight, n.weight.bias, n.bias.weight, n.bias.bias)]

        def build():
            with torch.no_grad():
                return (torch.cat([torch.cat([n.weight.weight, n.bias.weight]) for n in norms]).contiguous(),
                        torch.cat([torch.cat([n.weight.bias, n.bias.bias]) for n in norms]).contiguous())
        w_all, b_all = self._ada_cache.get("ada", ps, build)
        cond = condition.reshape(-1, condition.shape[-1]).float().contiguous()
        proj = runtime.linear_small(cond, w_all, b_all)       # [Bc, 2*len(norms)*dim]
        d = self.dim
        parts = [(proj[:, (2 * i) * d:(2 * i + 1) * d], proj[:, (2 * i + 1) * d:(2 * i + 2) * d])
                 for i in range(len(norms))]
        return [(parts[2 * li], parts[2 * li + 1]) for li in range(len(self.layers))]

    # The LayerNorm that consumes a layer's output moves only its STATISTICS: the fused feed-forward kernel writes (mean, rstd) per row and the
    # next layer's q/kv GEMM (ispk_gemm_bf16_lnin) normalises while it stages its fp32 input - no normalised copy in HBM.
    # ON: the feed-forward kernel is unchanged in time (106.6 us), q/kv goes 21.8 -> 26.6 us and the 15.0-us LayerNorm
    # launch disappears: 2.636 -> 2.587 ms per step with one batch in flight, 2.162 -> 2.147 with two.
    stats_layernorm = True

    def set_compute_dtype(self, dtype: torch.dtype):
        """fp32 (exact-fp32 MFMAs), bf16 (throughput path) or fp16 = the split-fp16 path: fp32-grade products as three fp16
        MFMAs over hi / lo terms (csrc/split.hip)."""
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        for layer in self.layers:
            layer.attention.compute_dtype = dtype
            layer.feed_forward.compute_dtype = dtype
        return self

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, context: Optional[Tensor] = None,
                context_mask: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                adaptive_condition: Optional[Tensor] = None, return_intermediates: bool = False, *,
                key_len: Optional[Tensor] = None, projected: Optional[Tensor] = None,
                out_dtype: torch.dtype = torch.float32, final_norm: bool = True):
        """`final_norm=False`: `.out` is the last layer's raw output - for a caller whose next kernel applies `self.norm` itself
        (the flow predictor's head, `runtime.flow_head`).
        `projected` lets a caller that already holds project_emb(x) (e.g. the Euler loop, which re-projects only the
        3 flow channels per step) skip the projection.  `out_dtype=torch.bfloat16` makes the final LayerNorm emit bf16
        for a bf16 consumer (the decoder's to_mel GEMM on the bf16 path)."""
        if projected is not None:
            out = projected
        elif isinstance(self.project_emb, nn.Identity):
            out = x.float()
        else:
            out = runtime.linear(x.float(), self.project_emb.weight, self.project_emb.bias)   # (strided rows are fine)
        if mask is not None and key_len is None:
            key_len = mask.sum(dim=1)
        intermediates = []
        ada = self._ada_all(adaptive_condition) if (self.adaptive_norm and adaptive_condition is not None) else None
        # bf16, plain LayerNorm: a layer's fused feed-forward kernel also emits the row statistics of the LayerNorm that
        # consumes its output (the next layer's attention_norm), which that layer's q/kv GEMM applies while it stages x
        cdt = self.layers[0].attention.compute_dtype
        stats = (self.stats_layernorm and not self.adaptive_norm
                 and cdt == torch.bfloat16 and context is None and attention_mask is None)
        chain = stats
        normed = None
        for li, layer in enumerate(self.layers):
            nxt = None
            if chain:
                if li + 1 < len(self.layers):
                    nn_ = self.layers[li + 1].attention_norm
                    nxt = (nn_.weight, nn_.bias, nn_.eps, False, "stats", self.layers[li + 1].attention)
                elif final_norm and out_dtype in (torch.float32, torch.bfloat16):
                    # last layer: the stack's own final norm (row-masked, transformer.py:205-206) - the split feed-forward's
                    # combine pass applies it from the same read (small batches), the fused layer kernel from the registers that
                    # store the rows (decoder-sized batches, ispk_attn_out_ffn_norm_bf16); the plain fused kernel ignores it
                    nxt = (self.norm.weight, self.norm.bias, self.norm.eps, mask is not None, out_dtype)
                if nxt is not None and (nxt[0] is None or nxt[1] is None):
                    nxt = None
            res = layer(out, mask=mask, context=context, context_mask=context_mask, attention_mask=attention_mask,
                        adaptive_condition=adaptive_condition, key_len=key_len, ada=None if ada is None else ada[li],
                        normed=normed, next_norm=nxt, skip_out=final_norm and li + 1 == len(self.layers))
            out, normed = res.out, res.next_normed
            if return_intermediates:
                intermediates.append(res.intermediates)
        if not final_norm:
            return TransformerOutput(out=out, intermediates=intermediates)
        if normed is None:
            if out_dtype == torch.float16:   # split fp16 planes [2, B, N, D] for a split-fp16 consumer GEMM
                normed = runtime.layernorm_split(out, self.norm.weight, self.norm.bias, row_mask=mask, eps=self.norm.eps)
            else:
                normed = runtime.layernorm(out, self.norm.weight, self.norm.bias, row_mask=mask, eps=self.norm.eps,
                                           out_dtype=out_dtype)
        return TransformerOutput(out=normed, intermediates=intermediates)
