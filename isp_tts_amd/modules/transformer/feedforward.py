"""`FeedForward`: Linear -> GELU(erf) -> Linear (tts/modules/transformer/feedforward.py:20-40 of the reference).

Two MFMA GEMMs; the exact-erf GELU (layers.py:29) is the first GEMM's epilogue, and the second GEMM's epilogue can
carry the residual add and row mask of transformer.py:105,110.  Dropout is identity in eval (the forward path).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import runtime
from ...staging import StagedWeights
from ..constructor import Constructor, ModuleConfig

_ACTS = {"gelu": runtime.EP_GELU, "swish": runtime.EP_SILU, "linear": 0}


@dataclass
class FeedForwardConfig(ModuleConfig):
    dim: int = 384
    inner_dim: int = 1536
    dropout: float = 0.0
    activation: str = "relu"
    bias: bool = False
    glu: bool = False


class FeedForward(nn.Module, Constructor):
    # path switches: class attributes (override on the class or an instance; nothing reads the environment)
    prenorm_fused = True   # norm -> feed-forward -> residual as one kernel (ispk_ffn_bf16_prenorm) when a caller offers it
    lnin_self = True       # two-GEMM path: feed_forward_norm applied by the first GEMM's own waves (ispk_gemm_bf16_lnin)
    next_qkv = True        # ispk_attn_out_ffn_qkv_bf16: the next layer's attention_norm + q/kv projection as the kernel's epilogue
    pair_kernel = True     # dim 384: the eight-wave kernel (ispk_ffn_bf16_prenorm2, csrc/ffn2.hip) instead of the four-wave one

    def __init__(self, dim: int = 384, inner_dim: int = 1536, dropout: float = 0.0, activation: str = "relu",
                 bias: bool = False, glu: bool = False):
        super().__init__()
        if glu or activation not in _ACTS:
            raise NotImplementedError(f"built for the recipes' feed-forward (gelu, no GLU); got activation="
                                      f"{activation!r}, glu={glu}")
        self.act_flag = _ACTS[activation]
        self.dropout_p = dropout
        # same container layout as the reference so the keys are net.0.weight / net.3.weight
        self.net = nn.Sequential(nn.Linear(dim, inner_dim, bias=bias), nn.GELU() if activation == "gelu" else nn.Identity(),
                                 nn.Dropout(dropout) if dropout > 0. else nn.Identity(),
                                 nn.Linear(inner_dim, dim, bias=bias))
        self.compute_dtype = torch.float32
        # the fused kernel gives a workgroup 128 rows and the whole inner dimension: with 128 row blocks or fewer (<= 16,384 rows:
        # half the chip) the split form - several workgroups per row block, each a slice of the inner dimension, + one combine
        # pass - fills the CUs instead (dim 384); other dims take the two-GEMM path there
        self.fused_min_rows = 128 * 128 + 1
        self._cache = StagedWeights()

    def _staged(self, dtype: torch.dtype):
        ps = (self.net[0].weight, self.net[3].weight)
        if dtype == torch.float16:   # split fp16 planes [2, N, K] for the split-fp16 kernels
            return self._cache.get(dtype, ps, lambda: (runtime.split_f16(ps[0].detach().float().contiguous()),
                                                       runtime.split_f16(ps[1].detach().float().contiguous())))
        if dtype == torch.bfloat16 and ps[0].is_cuda and ps[0].dtype == torch.float32:   # libispk launches (re-staged every training step)
            return self._cache.get(dtype, ps, lambda: (runtime.cast_bf16(ps[0].detach()), runtime.cast_bf16(ps[1].detach())))
        return self._cache.get(dtype, ps, lambda: (ps[0].detach().to(dtype).contiguous(),
                                                   ps[1].detach().to(dtype).contiguous()))

    def _packed_w2(self) -> Tensor:
        """W2 in the fused kernel's chunk-contiguous layout, staged once per weight version."""
        return self._cache.get("w2p", (self.net[3].weight,),
                               lambda: runtime.ffn_pack_w2(self._staged(torch.bfloat16)[1]))

    def _chunked_w2(self) -> Tensor:
        """W2 as chunk-contiguous [inner/32][D][32] blocks (ispk_ffn_bf16_prenorm2), staged once per weight version."""
        return self._cache.get("w2c", (self.net[3].weight,),
                               lambda: runtime.ffn_chunk_w2(self._staged(torch.bfloat16)[1]))

    def prenorm_ok(self, x: Tensor, norm) -> bool:
        """Can `forward_prenorm` run the block (norm -> feed-forward -> residual) as one kernel for this input?"""
        rows = x.numel() // x.shape[-1]
        return (self.prenorm_fused and isinstance(norm, nn.LayerNorm)
                and norm.weight is not None and norm.bias is not None and x.dtype == torch.float32
                and self.compute_dtype == torch.bfloat16 and self.act_flag == runtime.EP_GELU and x.shape[-1] in (256, 384)
                and rows >= self.fused_min_rows and self.net[0].bias is None
                and not (self.training and self.dropout_p > 0))

    # small batches (text encoder: 6,400 rows = 50 row blocks on 256 CUs): inner dimension split over several workgroups per
    # row block, partial products combined (+ residual, mask, the consumer's LayerNorm) by a second launch.
    # 22.4 (FFN2) + 19.8 (FFN1) + 7.1 (LayerNorm) + 5.8 (next LayerNorm) us -> measured in DESIGN.md
    split_small = True
    split_min_rows = 128       # (one row block: at 8 utterances per GPU the text encoder has 800 rows)
    max_splits = 8             # (16 measured no faster at 800 rows: the combine pass reads every partial)

    def split_ok(self, x: Tensor, norm) -> bool:
        rows = x.numel() // x.shape[-1]
        return (self.split_small and self.pair_kernel and isinstance(norm, nn.LayerNorm) and norm.weight is not None
                and norm.bias is not None and x.dtype == torch.float32 and self.compute_dtype == torch.bfloat16
                and self.act_flag == runtime.EP_GELU and x.shape[-1] == 384 and self.split_min_rows <= rows < self.fused_min_rows
                and self.net[0].bias is None and self.net[3].bias is None and self.net[0].weight.shape[0] % 64 == 0
                and not (self.training and self.dropout_p > 0))

    def forward_prenorm_split(self, x: Tensor, norm, *, mask: Optional[Tensor] = None, next_norm: Optional[tuple] = None,
                              attn_proj: Optional[tuple] = None):
        """(y, LN_next(y) | None) for small batches: ispk_ffn_bf16_prenorm2_split + ispk_ffn_combine_ln_f32.  `next_norm` =
        (weight, bias, eps, apply_mask, dtype) of the norm that consumes y.  `attn_proj` = (attention output before to_out,
        Attention._chunked_wo()): x is the LAYER's input and the kernel applies to_out, mask and residual itself."""
        w1, _ = self._staged(torch.bfloat16)
        rows, chunks = x.numel() // x.shape[-1], w1.shape[0] // 32
        blocks = (rows + 127) // 128
        splits = 1
        for s in (2, 3, 4, 6, 8, 12, 16):       # as many workgroups as fit one round of the 256 CUs, at least 2 chunks each
            if s <= self.max_splits and chunks % s == 0 and chunks // s >= 2 and blocks * s <= 256:
                splits = s
        return runtime.ffn_prenorm2_split(x, norm.weight, norm.bias, w1, self._chunked_w2(), mask, splits, next_norm=next_norm,
                                          norm_eps=norm.eps, attn_proj=attn_proj)

    def prenorm_unfused_ok(self, x: Tensor, norm) -> bool:
        """Two-GEMM path (e.g. an activation the fused kernel lacks): can the first Linear's GEMM apply `norm` itself?
        Only from decoder-sized batches on - below, the separate LayerNorm launch is cheaper (28.8 vs 19.4 + 4.9 us at
        6,400 rows: every workgroup of the split output repeats the fp32 staging)."""
        return (self.lnin_self and isinstance(norm, nn.LayerNorm) and norm.weight is not None
                and norm.bias is not None and x.dtype == torch.float32 and self.compute_dtype == torch.bfloat16
                and x.shape[-1] in (256, 384) and x.numel() // x.shape[-1] >= self.fused_min_rows
                and not (self.training and self.dropout_p > 0))

    def forward_prenorm_unfused(self, x: Tensor, norm, *, mask: Optional[Tensor] = None) -> Tensor:
        """y = [mask] * (x + W2 act(W1 norm(x))) as two GEMMs, the LayerNorm applied by the first one while it stages x
        (ispk_gemm_bf16_lnin, statistics by its own waves); the `* mask` of transformer.py:102 is dead under the output mask."""
        w1, w2 = self._staged(torch.bfloat16)
        hidden = runtime.gemm_lnin(x, None, norm.weight, norm.bias, w1, bias=self.net[0].bias, flags=self.act_flag,
                                   ln_eps=norm.eps)
        return runtime.gemm(hidden, w2, bias=self.net[3].bias, resid=x, mask=mask,
                            flags=runtime.EP_MASK_OUT if mask is not None else 0, out_dtype=torch.float32)

    def forward_prenorm(self, x: Tensor, norm, *, mask: Optional[Tensor] = None, next_norm: Optional[tuple] = None):
        """y = [mask] * (x + feed_forward(norm(x))) in one kernel (ispk_ffn_bf16_prenorm), x fp32; with `next_norm` =
        (.., eps, .., "stats") also the output rows' (mean, rstd) for the next layer's q/kv GEMM.  Returns (y, stats)."""
        w1, _ = self._staged(torch.bfloat16)
        want = next_norm is not None and next_norm[4] == "stats"
        if self.pair_kernel and x.shape[-1] == 384 and self.net[3].bias is None and w1.shape[0] % 32 == 0:
            res = runtime.ffn_prenorm2(x, norm.weight, norm.bias, w1, self._chunked_w2(), mask=mask,
                                       flags=runtime.EP_MASK_OUT if mask is not None else 0, norm_eps=norm.eps,
                                       want_stats=want, stats_eps=next_norm[2] if want else 1e-5)
            return res if want else (res, None)
        res = runtime.ffn_prenorm(x, norm.weight, norm.bias, w1, self._packed_w2(), mask=mask, bias2=self.net[3].bias,
                                  flags=runtime.EP_MASK_OUT if mask is not None else 0, norm_eps=norm.eps,
                                  want_stats=want, stats_eps=next_norm[2] if want else 1e-5)
        return res if want else (res, None)

    def proj_ok(self, x: Tensor, norm) -> bool:
        """Can `forward_proj_prenorm` run to_out + residual + norm + feed-forward + residual as one kernel (x: the layer's input)?"""
        return (self.prenorm_ok(x, norm) and self.pair_kernel and x.shape[-1] == 384 and self.net[3].bias is None
                and self.net[0].weight.shape[0] % 32 == 0 and self.net[0].weight.shape[0] >= 64)

    def forward_proj_prenorm(self, x: Tensor, attn_out: Tensor, woc: Tensor, norm, *, mask: Optional[Tensor] = None,
                             next_norm: Optional[tuple] = None, want_out: bool = True):
        """(y, stats | qkv | None) with x1 = x + [mask] * to_out(attn_out), y = [mask] * (x1 + feed_forward(norm(x1))) in one
        kernel (ispk_attn_out_ffn_bf16): x1 exists only in the kernel's accumulators.  `woc` = Attention._chunked_wo().
        `next_norm` = (weight, bias, eps, _, "stats"[, next layer's Attention]): with the Attention given and 6 heads the second
        result is that layer's q/kv rows (bf16 [..., 512]) instead of the row statistics.  `next_norm` = (weight, bias, eps,
        apply_mask, torch dtype) - the STACK's final norm: the second result is LN_final(y) in that dtype, and with
        `want_out=False` y itself is not stored (first result None)."""
        w1, _ = self._staged(torch.bfloat16)
        if next_norm is not None and next_norm[4] in (torch.float32, torch.bfloat16):
            # the stack's final LayerNorm from the same kernel: (y | None, LN_final(y)); y itself only if the caller wants it
            return runtime.attn_out_ffn(x, attn_out, woc, norm.weight, norm.bias, w1, self._chunked_w2(), mask=mask, norm_eps=norm.eps,
                                        final_norm=next_norm[:5], want_out=want_out)
        want = next_norm is not None and next_norm[4] == "stats"
        nxt_attn = next_norm[5] if want and len(next_norm) > 5 else None
        if nxt_attn is not None and self.next_qkv and nxt_attn.heads * 64 + 128 == 512 and nxt_attn.dim == x.shape[-1]:
            # the next layer's attention_norm + q/kv projection from the same kernel: (y, qkv)
            return runtime.attn_out_ffn(x, attn_out, woc, norm.weight, norm.bias, w1, self._chunked_w2(), mask=mask,
                                        norm_eps=norm.eps,
                                        next_qkv=(next_norm[0], next_norm[1], next_norm[2], nxt_attn._chunked_wqkv()))
        res = runtime.attn_out_ffn(x, attn_out, woc, norm.weight, norm.bias, w1, self._chunked_w2(), mask=mask, norm_eps=norm.eps,
                                   want_stats=want, stats_eps=next_norm[2] if want else 1e-5)
        return res if want else (res, None)

    def forward(self, x: Tensor, *, residual: Optional[Tensor] = None, mask: Optional[Tensor] = None) -> Tensor:
        if self.training and self.dropout_p > 0:
            raise NotImplementedError("feed-forward dropout (training) is outside the forward-path scope")
        dt = self.compute_dtype
        w1, w2 = self._staged(dt)
        if dt == torch.float16:   # split-fp16 path: x fp32 or split planes
            xs = x if x.dtype == torch.float16 else runtime.split_f16(x.float().contiguous())
            hidden = runtime.gemm_split(xs, w1, bias=self.net[0].bias, flags=self.act_flag, out_split=True)
            return runtime.gemm_split(hidden, w2, bias=self.net[3].bias, resid=residual, mask=mask,
                                      flags=runtime.EP_MASK_OUT if mask is not None else 0)
        if x.dtype != dt:
            x = runtime.cast_bf16(x) if dt == torch.bfloat16 else x.float()
        rows = x.numel() // x.shape[-1]
        if (dt == torch.bfloat16 and self.act_flag == runtime.EP_GELU and x.shape[-1] in (256, 384)
                and rows >= self.fused_min_rows):
            # one kernel for Linear -> GELU -> Linear (+ residual, mask): the hidden activations never reach HBM
            return runtime.ffn_fused(x, w1, self._packed_w2(), resid=residual, mask=mask, bias1=self.net[0].bias, bias2=self.net[3].bias,
                                     flags=runtime.EP_MASK_OUT if mask is not None else 0)
        hidden = runtime.gemm(x, w1, bias=self.net[0].bias, flags=self.act_flag)
        flags = runtime.EP_MASK_OUT if mask is not None else 0
        return runtime.gemm(hidden, w2, bias=self.net[3].bias, resid=residual, mask=mask, flags=flags,
                            out_dtype=torch.float32)
