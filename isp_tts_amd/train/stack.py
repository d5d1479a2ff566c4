"""A `Transformer` stack as ONE autograd node: the forward keeps what the backward needs, the backward is
the kernels of csrc/backward.hip plus the forward's own NT GEMM on transposed weights.

Scope of this first cut: plain LayerNorm stacks (TextEncoder, MelDecoder: transformer.py:174-211 with emb_dim == dim),
fp32, ALiBi multi-query attention, exact-erf GELU, no Linear biases.  Dropout as the recipes train with it (attention
probabilities and feed-forward activations, p = 0.1): the masks are a hash of (seed, element index) evaluated inside the
kernels, forward and backward alike - nothing is stored; a step's seeds come from torch's CPU generator (torch.manual_seed
makes a run reproducible), the draw sequence differs from torch's own dropout kernels.

Forward per layer (transformer.py:62-118 as the inference path launches it, csrc/gemm.hip epilogues), m = row mask:
    h   = LN1(x)                 qkv = h Wqkv^T            o = ALiBi-MQA(qkv)          x1 = x + m (o Wo^T)
    h2  = m LN2(x1)              u   = h2 W1^T             a = gelu(u)                 y  = m (x1 + a W2^T)
and out = m LN_f(y_last).  Backward, given dy (rows of padded positions are exactly zero: the final LayerNorm's backward
masks them, and every step below keeps zero rows zero):
    da = (m dy) W2        dW2 = (m dy)^T a        du = da gelu'(u)       dW1 = du^T h2       dh2 = du W1
    dx1 = dy + LN2'(x1, m dh2)                    dWo = (m dx1)^T o      do  = (m dx1) Wo
    dqkv, dlogslopes = attention'(qkv, o, do)     dWqkv = dqkv^T h       dh  = dqkv Wqkv
    dx  = dx1 + LN1'(x, dh)
Saved per layer: x, x1, u (fp32) and h, qkv, o, h2, a (bf16 under AMP): ~0.5 GB per decoder layer at 64 x 512 frames, small
beside 288 GB of HBM, so nothing is recomputed except the LayerNorm / softmax statistics.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .. import runtime
from ..modules.transformer.transformer import Transformer
from .images import layer_images


def _check(tr: Transformer) -> None:
    if tr.adaptive_norm:
        raise NotImplementedError("training backward: adaptive-norm stacks (the flow predictor) are not built")
    if not isinstance(tr.project_emb, torch.nn.Identity) and tr.project_emb.in_features > 8:
        raise NotImplementedError("training backward: input projections with more than 8 features are not built")
    for layer in tr.layers:
        att, ff = layer.attention, layer.feed_forward
        if ff.net[0].bias is not None or ff.net[3].bias is not None or ff.act_flag != runtime.EP_GELU:
            raise NotImplementedError("training backward: feed-forward with biases / non-GELU activation is not built")


def stack_parameters(tr: Transformer) -> list:
    """The parameters the node differentiates, in the order `TransformerStackFunction` takes and returns them."""
    ps = [] if isinstance(tr.project_emb, torch.nn.Identity) else [tr.project_emb.weight, tr.project_emb.bias]
    for layer in tr.layers:
        att, ff = layer.attention, layer.feed_forward
        ps += [layer.attention_norm.weight, layer.attention_norm.bias, att.to_q.weight, att.to_kv.weight,
               att.rel_pos.learned_logslopes, att.to_out.weight, layer.feed_forward_norm.weight, layer.feed_forward_norm.bias,
               ff.net[0].weight, ff.net[3].weight]
    return ps + [tr.norm.weight, tr.norm.bias]


def _mm(a: Tensor, w32: Tensor, w16: Optional[Tensor], out_dtype: torch.dtype = torch.float32, **kw) -> Tensor:
    """a . w^T.  fp32 step (`w16` None): fp32 operands and result.  AMP step: `a` is ALREADY bf16 - written in bf16 by the
    kernel that produced it (LayerNorm, GELU, one cast of an attention / residual-gradient tensor that the weight gradient
    shares) - `w16` the staged bf16 weight, fp32 accumulation; the result fp32 or, where it is only a GEMM operand again, bf16."""
    if w16 is None:
        return runtime.gemm(a, w32, **kw)
    assert a.dtype == torch.bfloat16
    return runtime.gemm(a, w16, out_dtype=out_dtype, **kw)


def _deliver(param: Tensor, producer, *args, **kw):
    """A parameter's gradient.  When the parameter's .grad is a preallocated buffer of an optimizer arena
    (`FlatParameters` marks it `_ispk_grad_arena`), the producing kernel writes - or adds, if something has been delivered
    since the arena was zeroed - straight into it and autograd is handed None: no AccumulateGrad launch per parameter
    (206 element-wise ATen adds per step otherwise).  Else the gradient is returned for autograd to place."""
    if not param.requires_grad:       # frozen after the arena was built: no gradient, and nothing written behind autograd's back
        return None
    g = param.grad
    if g is not None and getattr(g, "_ispk_grad_arena", False) and producer is runtime.gemm_tn:
        runtime.gemm_tn(*args, out=g, accumulate=getattr(g, "_ispk_dirty", False), **kw)
        g._ispk_dirty = True
        return None
    return producer(*args, **kw)


class TransformerStackFunction(torch.autograd.Function):
    """out = Transformer(x, mask).out with every parameter of the stack as a differentiable input.  `amp`: the eight
    Linear GEMMs of a layer (forward and the dX ones of the backward) take bf16 operands - the reference trains under
    autocast (recipes/default.yaml:56) - while LayerNorm, attention, GELU, every weight gradient and the residual stream stay
    fp32."""

    @staticmethod
    def forward(ctx, tr: Transformer, x: Tensor, mask: Optional[Tensor], amp: bool, key_len: Optional[Tensor], *params: Tensor):
        _check(tr)
        ctx.want_dx = x.requires_grad          # (asked before the copy below: a tensor made inside forward never requires grad)
        x = x.float().contiguous()
        if key_len is None and mask is not None:        # (the callers that know the lengths pass them: no reduction launch)
            key_len = mask.sum(dim=1)
        proj = not isinstance(tr.project_emb, torch.nn.Identity)
        if proj:      # transformer.py:170, :189: a Linear from the few input features (the adaptor's pitch / energy pair)
            ctx.proj_in = x
            x = runtime.linear(x, tr.project_emb.weight, tr.project_emb.bias)
        tape, out = [], x
        images = layer_images(tr, amp)
        base_seed = runtime.draw_seed()      # a host generator that follows torch's seed: torch.manual_seed reproduces a run
        for li, layer in enumerate(tr.layers):
            att, ff, an, fn = layer.attention, layer.feed_forward, layer.attention_norm, layer.feed_forward_norm
            im = images[li]                     # this step's weight images (train/images.py)
            wqkv, wo, w1, w2 = (None,) * 4 if amp else (im["wqkv"], im["wo"], im["w1"], im["w2"])
            wqkv16, wo16, w116, w216 = (im["wqkv"], im["wo"], im["w1"], im["w2"]) if amp else (None,) * 4
            slopes = im["slopes"]
            adt = torch.bfloat16 if amp else torch.float32        # dtype of the tensors that are GEMM operands only
            h = runtime.layernorm(out, an.weight, an.bias, eps=an.eps, out_dtype=adt)
            qkv = _mm(h, wqkv, wqkv16, out_dtype=adt)        # AMP: bf16 q / k / v, as SDPA sees them under autocast
            p_att = float(att.attend.dropout) if layer.training else 0.0
            p_ff = float(ff.dropout_p) if layer.training else 0.0
            seed_att, seed_ff = base_seed + 2 * li, base_seed + 2 * li + 1
            lse = None
            if p_att > 0 or amp:   # dropped attention probabilities (attend.py:118); the rows' log-sum-exp is kept for the
                # backward.  Under AMP forward and backward both take bf16 operands (and must share the statistics)
                o, lse = runtime.alibi_mqa_attention_train(qkv, att.heads, slopes, key_len, p_att, seed_att)
            else:
                o = runtime.alibi_mqa_attention(qkv, att.heads, slopes, key_len)
            x1 = _mm(o, wo, wo16, resid=out, mask=mask, flags=runtime.EP_MASK_ACC if mask is not None else 0)
            h2 = runtime.layernorm(x1, fn.weight, fn.bias, row_mask=mask, eps=fn.eps, out_dtype=adt)
            if amp and h2.shape[-1] in (256, 384):      # one launch: u (bf16, autocast's Linear output) and a = dropout(gelu(u))
                u, a = runtime.gemm_gelu_train(h2, w116, p_ff, seed_ff)
            else:
                u = _mm(h2, w1, w116, out_dtype=adt)
                a = runtime.gelu(u, p_ff, seed_ff, out_dtype=adt)        # GELU, then nn.Dropout (feedforward.py:35)
            y = _mm(a, w2, w216, resid=x1, mask=mask, flags=runtime.EP_MASK_OUT if mask is not None else 0)
            tape.append((out, h, qkv, o, x1, h2, u, a, lse, p_att, seed_att, p_ff, seed_ff))
            out = y
        final = runtime.layernorm(out, tr.norm.weight, tr.norm.bias, row_mask=mask, eps=tr.norm.eps)
        ctx.tr, ctx.mask, ctx.key_len, ctx.tape, ctx.last, ctx.amp, ctx.params = tr, mask, key_len, tape, out, amp, params
        return final

    @staticmethod
    def backward(ctx, dfinal: Tensor):
        tr, mask, key_len, amp = ctx.tr, ctx.mask, ctx.key_len, ctx.amp
        mflag = runtime.EP_MASK_OUT if mask is not None else 0
        grads: list = []
        # (AMP: every LayerNorm backward also leaves its dx as bf16 rows - the operand of the dX GEMM and weight gradient below it)
        dy, dgf, dbf, *rest = runtime.layernorm_bwd(ctx.last, dfinal.float().contiguous(), tr.norm.weight, row_mask=mask,
                                                    eps=tr.norm.eps, bf16_copy=amp)
        dyg = rest[0] if amp else dy
        images = layer_images(tr, amp)         # (the forward's images: same parameter versions)
        for li in reversed(range(len(tr.layers))):
            layer = tr.layers[li]
            xin, h, qkv, o, x1, h2, u, a, lse, p_att, seed_att, p_ff, seed_ff = ctx.tape[li]
            att, ff, an, fn = layer.attention, layer.feed_forward, layer.attention_norm, layer.feed_forward_norm
            im = images[li]
            slopes = im["slopes"]
            wqkv_t, wo_t, w1_t, w2_t = (None,) * 4 if amp else (im["wqkv_t"], im["wo_t"], im["w1_t"], im["w2_t"])
            wqkv_t16, wo_t16, w1_t16, w2_t16 = (im["wqkv_t"], im["wo_t"], im["w1_t"], im["w2_t"]) if amp else (None,) * 4
            gdt = torch.bfloat16 if amp else torch.float32
            # Under AMP every tensor that is ONLY a GEMM operand lives in bf16 - a (forward), da / du (their producers write
            # bf16; q / k / v, the attention output, dO and dqkv: the attention kernels read and write bf16), and ONE bf16 copy each of
            # the fp32 residual-stream gradients that a dX GEMM and a weight gradient both read (dy, dx1).
            # feed-forward block
            # (rows of padded positions are exactly zero in dy and dx1 - see the module header - so the weight gradients' row mask
            # changes nothing; the AMP step leaves it out and gets the LDS-DMA kernel, which has no mask path)
            wmask = None if amp else mask
            dw2 = _deliver(ff.net[3].weight, runtime.gemm_tn, dyg, a, row_mask=wmask, bf16=amp)       # [dim, inner]
            if amp and dyg.shape[-1] in (256, 384):     # (m dy) W2 and the GELU / dropout derivative in the GEMM's epilogue
                du = runtime.gemm_gelu_bwd(dyg, w2_t16, u, mask, p_ff, seed_ff)
            else:
                da = _mm(dyg, w2_t, w2_t16, out_dtype=gdt, mask=mask, flags=mflag)                     # (m dy) W2
                du = runtime.gelu_bwd(da, u, out=da, dropout_p=p_ff, seed=seed_ff)
            dw1 = _deliver(ff.net[0].weight, runtime.gemm_tn, du, h2, bf16=amp)                         # [inner, dim]
            dh2 = _mm(du, w1_t, w1_t16)
            dx1, dg2, db2, *rest = runtime.layernorm_bwd(x1, dh2, fn.weight, row_mask=mask, dx=dy, add_to_dx=True, eps=fn.eps,
                                                         bf16_copy=amp)
            # attention block
            dx1g = rest[0] if amp else dx1
            dwo = _deliver(att.to_out.weight, runtime.gemm_tn, dx1g, o, row_mask=wmask, bf16=amp)      # [dim, heads*64]
            d_o = _mm(dx1g, wo_t, wo_t16, out_dtype=gdt, mask=mask, flags=mflag)
            dqkv, dls = runtime.alibi_mqa_attention_bwd(qkv, o, d_o, att.heads, slopes, key_len, lse=lse, dropout_p=p_att,
                                                        seed=seed_att)                     # AMP: bf16 in, bf16 out
            dwqkv = runtime.gemm_tn(dqkv, h, bf16=amp)                                      # [heads*64 + 128, dim]
            dh = _mm(dqkv, wqkv_t, wqkv_t16)
            dy, dg1, db1, *rest = runtime.layernorm_bwd(xin, dh, an.weight, dx=dx1, add_to_dx=True, eps=an.eps, bf16_copy=amp)
            dyg = rest[0] if amp else dy
            hq = att.heads * 64
            ls = att.rel_pos.learned_logslopes
            grads = [dg1, db1, dwqkv[:hq], dwqkv[hq:], dls[:ls.numel()].view_as(ls), dwo, dg2, db2, dw1, dw2] + grads
        ctx.tape = None
        if not isinstance(tr.project_emb, torch.nn.Identity):
            # dy is the gradient of the projected input: d W = dy^T x over a handful of input features, d b = column sums;
            # the raw features are targets (no gradient wanted)
            grads = [runtime.smallk_wgrad(dy, ctx.proj_in), runtime.colsum(dy)] + grads
            # d x = dy W (a handful of outputs): only when the features themselves carry a gradient (the soft averages do)
            dy = runtime.linear_small(dy, runtime.transpose(tr.project_emb.weight.detach()), None) if ctx.want_dx else None
        # small gradients (norm weights, slopes, ...) go to the optimizer arena in ONE launch; the big ones were delivered above
        return (None, dy, None, None, None, *runtime.deliver_grads(list(zip(ctx.params, grads + [dgf, dbf]))))


def transformer_train_forward(tr: Transformer, x: Tensor, mask: Optional[Tensor] = None, amp: bool = False,
                              key_len: Optional[Tensor] = None) -> Tensor:
    """`tr(x, mask).out` as a differentiable node (gradients reach x and every parameter of the stack); `amp` = bf16
    operands for the Linear GEMMs (see TransformerStackFunction)."""
    return TransformerStackFunction.apply(tr, x, mask, amp, key_len, *stack_parameters(tr))


class ForkFunction(torch.autograd.Function):
    """x -> (x, x) for a tensor with two differentiable consumers: the two gradients are added by libispk launches here, so
    the autograd engine never has to (its own accumulation is an ATen add)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1: Optional[Tensor], g2: Optional[Tensor]):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        g1, g2 = g1.float().contiguous(), g2.float().contiguous()
        out = torch.empty_like(g1)
        runtime.segments([(g1, out, runtime.SEG_COPY)])
        runtime.segments([(g2, out, runtime.SEG_ADD)])
        return out


def fork(x: Tensor):
    return ForkFunction.apply(x) if x.requires_grad and x.is_cuda else (x, x)


class ToMelFunction(torch.autograd.Function):
    """mel = mask * (dec W^T + b), stored [B, 80, T] (model.py:167-168; forward = the inference path's `runtime.to_mel`).
    Backward: masked gradient rows, then d dec = g W (NT GEMM on W^T), dW = g^T dec, db = column sums of g."""

    @staticmethod
    def forward(ctx, dec: Tensor, weight: Tensor, bias: Tensor, mask: Optional[Tensor], amp: bool = False):
        """`amp`: bf16 operands for the three GEMMs (autocast covers this Linear too), fp32 mel and gradients."""
        dec = dec.float().contiguous()
        w = weight.detach()
        if amp:
            dec, w = runtime.cast_bf16(dec), runtime.cast_bf16(w)
        ctx.save_for_backward(dec, weight)
        ctx.mask, ctx.params, ctx.amp = mask, (weight, bias), amp
        return runtime.to_mel(dec, w, bias.detach(), mask)

    @staticmethod
    def backward(ctx, dmel: Tensor):
        dec, weight = ctx.saved_tensors
        g = runtime.mel_grad_rows(dmel.float(), ctx.mask)                 # [B, T, 80]
        db = runtime.colsum(g)
        wt = runtime.transpose(weight.detach())
        if ctx.amp:
            g, wt = runtime.cast_bf16(g), runtime.cast_bf16(wt)
        d_dec = runtime.gemm(g, wt, out_dtype=torch.float32)               # [B, T, dim]
        dw = runtime.gemm_tn(g, dec)                                       # [80, dim]
        return (d_dec, *runtime.deliver_grads(list(zip(ctx.params, (dw, db)))), None, None)


def mel_decoder_train_forward(model, dec_in: Tensor, dec_mask: Optional[Tensor], amp: bool = False) -> Tensor:
    """MelDecoder + to_mel of `AcousticModel` (model.py:165-168) as differentiable nodes: dec_in [B, T, dim] -> mel [B, 80, T]."""
    dec = transformer_train_forward(model.decoder, dec_in, dec_mask, amp)
    return ToMelFunction.apply(dec, model.to_mel.weight, model.to_mel.bias, dec_mask)


class LengthRegulateFunction(torch.autograd.Function):
    """out[b] = A[b] x[b] (LengthRegulator, soft branch: temporal_adaptor.py:411-436; forward = `runtime.length_regulate`).
    Backward: d x[b] = A[b]^T d out[b] (one batched transposed product) and, when the alignment carries a gradient (the
    path by which the mel loss reaches the aligner), d A[b] = d out[b] x[b]^T (one batched NT product)."""

    @staticmethod
    def forward(ctx, x: Tensor, alignment: Tensor, durations: Tensor, frames: int):
        out, dec_len, dec_mask = runtime.length_regulate(x, durations, alignment, frames, max_len=frames)
        ctx.set_materialize_grads(False)         # (no zero tensors for the two non-differentiable outputs)
        ctx.save_for_backward(alignment, x)
        ctx.want_da = alignment.requires_grad
        ctx.mark_non_differentiable(dec_len, dec_mask)
        return out, dec_len, dec_mask

    @staticmethod
    def backward(ctx, d_out: Tensor, _dl, _dm):
        alignment, x = ctx.saved_tensors
        d_out = d_out.float().contiguous()
        d_a = runtime.gemm_batched(d_out, x, out=torch.empty_like(alignment)) if ctx.want_da else None
        return runtime.gemm_tn_batched(alignment, d_out), d_a, None, None


class MaskedLinearResidualFunction(torch.autograd.Function):
    """out = residual + mask * (h W^T + b): the output Linear of the adaptor's `TransformerTemporalModule`
    (temporal_adaptor.py:43-59) fused with the caller's `enc_out + embedding(...)` (:297), as the forward path launches it."""

    @staticmethod
    def forward(ctx, h: Tensor, weight: Tensor, bias: Tensor, mask: Optional[Tensor], residual: Tensor):
        h = h.float().contiguous()
        ctx.save_for_backward(h, weight)
        ctx.mask, ctx.params = mask, (weight, bias)
        return runtime.gemm(h, weight.detach(), bias=bias.detach(), mask=mask, flags=runtime.EP_MASK_ACC if mask is not None else 0,
                            resid=residual)

    @staticmethod
    def backward(ctx, dy: Tensor):
        h, weight = ctx.saved_tensors
        dy, mask = dy.float().contiguous(), ctx.mask
        dh = runtime.gemm(dy, runtime.transpose(weight.detach()), mask=mask, flags=runtime.EP_MASK_OUT if mask is not None else 0)
        dw, db = runtime.deliver_grads(list(zip(ctx.params, (runtime.gemm_tn(dy, h, row_mask=mask), runtime.colsum(dy, mask)))))
        return dh, dw, db, None, dy


class EmbedTokensFunction(torch.autograd.Function):
    """emb = table[text] (model.py:131, nn.Embedding with padding_idx 0) - forward by `runtime.embed_tokens`, backward a
    deterministic row gather-sum per vocabulary entry (no gradient for the padding row)."""

    @staticmethod
    def forward(ctx, text: Tensor, table: Tensor, text_len: Tensor):
        emb, mask = runtime.embed_tokens(text, table.detach(), text_len)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(text)
        ctx.vocab, ctx.table = table.shape[0], table
        ctx.mark_non_differentiable(mask)
        return emb, mask

    @staticmethod
    def backward(ctx, d_emb: Tensor, _dm):
        (text,) = ctx.saved_tensors
        (dt,) = runtime.deliver_grads([(ctx.table, runtime.embedding_bwd(text, d_emb.float().contiguous(), ctx.vocab, padding_idx=0))])
        return None, dt, None


def acoustic_mel_train_forward(model, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Tensor, energy: Tensor,
                               amp: bool = False) -> Tensor:
    """The teacher-forced forward of `AcousticModel` (model.py:116-174) as a differentiable chain for the MEL loss:
    text embedding -> TextEncoder -> [aligner, frozen] -> adaptor embedding stack + length regulator -> MelDecoder -> to_mel.
    Gradients reach the text embedding table, the encoder, the adaptor's embedding module, the decoder and to_mel - every
    parameter the mel loss reaches outside the aligner (which the reference feeds a DETACHED encoder output, model.py:139,
    and trains through attn_soft: that path, d A = d out x^T, is not built) and the flow predictor (whose outputs the
    mel loss does not see).  -> mel [B, 80, M]."""
    ad = model.temporal_adaptor
    emb, enc_mask = EmbedTokensFunction.apply(text, model.text_embedding.weight, text_len)
    enc_out = transformer_train_forward(model.encoder, emb, enc_mask, amp)
    with torch.no_grad():
        attn_soft, _ = model.aligner.attention(mel, enc_out.detach().transpose(1, 2), mel_len, text_len)
        feats = runtime.soft_average(attn_soft, pitch, energy, None, text_len)        # pitch / energy targets (:257-269)
    emod = ad.embedding
    h = transformer_train_forward(emod.transformer, feats[..., 1:3], enc_mask, amp)
    x = MaskedLinearResidualFunction.apply(h, emod.linear_layer.weight, emod.linear_layer.bias, enc_mask, enc_out)
    dec_in, dec_len, dec_mask = LengthRegulateFunction.apply(x, attn_soft, mel_len.view(-1, 1), mel.shape[2])
    dec = transformer_train_forward(model.decoder, dec_in, dec_mask, amp)
    return ToMelFunction.apply(dec, model.to_mel.weight, model.to_mel.bias, dec_mask)
