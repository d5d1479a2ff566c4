"""Training backward of the flow predictor (`FlowTransformerTemporalModule`, temporal_adaptor.py:72-147): the flow-matching
loss's gradient for the predictor's own parameters - time-embedding MLP, the 12 AdaptiveLayerNorm condition projections, the
387 -> 256 input projection, the 3 x 256 adaptive-norm stack, the 256 -> 3 output Linear - and for the encoder output that
conditions it.  Forward launches are the inference path's; every backward is a kernel of csrc/backward.hip / csrc/train.hip
(fp32, dropout by in-kernel masks as in `stack.py`).

    time_emb = TimeMLP(t)                                     TimeEmbeddingFunction
    ss       = [scale | shift] of all norms = Linear(time_emb)  AdaProjectionFunction      (Transformer._ada_all)
    proj     = cond W[:, 3:]^T + b + x_t W[:, :3]^T           ProjectSplitFunction        (transformer.py:170 on cat([x_t, cond]))
    out      = AdaLN stack(proj; ss), final LayerNorm         AdaptiveStackFunction
    raw      = out Wl^T + bl                                  SmallOutputLinearFunction
    loss     = masked mean of (raw m - flow)^2                FlowLossFunction            (temporal_adaptor.py:145-146)
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .. import runtime
from ..modules.transformer.transformer import Transformer
from .images import layer_images
from .stack import _mm


class TimeEmbeddingFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t: Tensor, inv_freq: Tensor, freq_scale: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor):
        ctx.save_for_backward(t, inv_freq, freq_scale, w0, b0, w1)
        ctx.params = (w0, b0, w1, b1)
        return runtime.time_embedding(t, inv_freq, freq_scale, w0.detach(), b0.detach(), w1.detach(), b1.detach())

    @staticmethod
    def backward(ctx, d_out: Tensor):
        t, inv_freq, freq_scale, w0, b0, w1 = ctx.saved_tensors
        dw0, db0, dw1, db1 = runtime.time_embedding_bwd(t, inv_freq, freq_scale, w0.detach(), b0.detach(), w1.detach(), d_out)
        return (None, None, None, *runtime.deliver_grads(list(zip(ctx.params, (dw0, db0, dw1, db1)))))


def _ada_norms(tr: Transformer) -> list:
    return [n for layer in tr.layers for n in (layer.attention_norm, layer.feed_forward_norm)]


def ada_parameters(tr: Transformer) -> list:
    return [p for n in _ada_norms(tr) for p in (n.weight.weight, n.weight.bias, n.bias.weight, n.bias.bias)]


class AdaProjectionFunction(torch.autograd.Function):
    """Scale and shift rows of every AdaptiveLayerNorm of the stack from the condition, one launch (Transformer._ada_all):
    ss[:, (2i) D : (2i+1) D] = scale of norm i, ss[:, (2i+1) D : (2i+2) D] = its shift."""

    @staticmethod
    def forward(ctx, tr: Transformer, cond: Tensor, *params: Tensor):
        norms = _ada_norms(tr)
        w_all = runtime.cat0([w for n in norms for w in (n.weight.weight, n.bias.weight)])      # one launch each
        b_all = runtime.cat0([b for n in norms for b in (n.weight.bias, n.bias.bias)])
        cond = cond.float().contiguous()
        ctx.save_for_backward(cond, w_all)
        ctx.dim, ctx.n, ctx.params = tr.dim, len(norms), params
        return runtime.linear_small(cond, w_all, b_all)

    @staticmethod
    def backward(ctx, d_ss: Tensor):
        cond, w_all = ctx.saved_tensors
        d_ss = d_ss.float().contiguous()
        dw = runtime.gemm_tn(d_ss, cond)                              # [2 n D, cond_dim]
        db = runtime.colsum(d_ss)
        # d cond [B, cond_dim] = d_ss w_all: a reduction over the 2 n D stacked outputs with only B x cond_dim results - as a
        # weight-gradient-shaped product (rows = the reduction index, split over row ranges) instead of one workgroup's K loop
        d_cond = (runtime.gemm_tn(runtime.transpose(d_ss), w_all) if d_ss.shape[0] % 4 == 0 and w_all.shape[1] % 4 == 0
                  else runtime.gemm(d_ss, runtime.transpose(w_all)))
        d, grads = ctx.dim, []
        for i in range(ctx.n):
            grads += [dw[(2 * i) * d:(2 * i + 1) * d], db[(2 * i) * d:(2 * i + 1) * d],
                      dw[(2 * i + 1) * d:(2 * i + 2) * d], db[(2 * i + 1) * d:(2 * i + 2) * d]]
        return (None, d_cond, *runtime.deliver_grads(list(zip(ctx.params, grads))))


class ProjectSplitFunction(torch.autograd.Function):
    """proj = cat([x_t, cond]) W^T + b (transformer.py:170, :189) without the concatenation: the 384 condition channels are
    one GEMM, the 3 flow channels a K = 3 update on its result.  x_t (noise / target mix) gets no gradient."""

    @staticmethod
    def forward(ctx, x_t: Tensor, cond: Tensor, weight: Tensor, bias: Tensor):
        k = x_t.shape[-1]
        cond = cond.float().contiguous()
        x_t = x_t.float().contiguous()
        wc = runtime.copy2d(weight.detach()[:, k:], torch.empty((weight.shape[0], weight.shape[1] - k), dtype=torch.float32,
                                                                device=weight.device))
        ctx.save_for_backward(x_t, cond, wc)
        ctx.k, ctx.params = k, (weight, bias)
        cp = runtime.gemm(cond, wc, bias=bias.detach())
        return runtime.linear_small(x_t, weight.detach()[:, :k], None, resid=cp)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x_t, cond, wc = ctx.saved_tensors
        dy = dy.float().contiguous()
        k = ctx.k
        dw = torch.empty((wc.shape[0], k + wc.shape[1]), dtype=torch.float32, device=dy.device)
        runtime.gemm_tn(dy, cond, out=dw[:, k:])
        runtime.copy2d(runtime.smallk_wgrad(dy, x_t), dw[:, :k])
        return (None, runtime.gemm(dy, runtime.transpose(wc)), *runtime.deliver_grads(list(zip(ctx.params, (dw, runtime.colsum(dy))))))


class SmallOutputLinearFunction(torch.autograd.Function):
    """raw = out W^T + b with a handful of outputs (the predictor's 256 -> 3 `linear_layer`, temporal_adaptor.py:98)."""

    @staticmethod
    def forward(ctx, h: Tensor, weight: Tensor, bias: Tensor):
        h = h.float().contiguous()
        ctx.save_for_backward(h, weight)
        ctx.params = (weight, bias)
        return runtime.linear_small(h, weight.detach(), bias.detach())

    @staticmethod
    def backward(ctx, d_raw: Tensor):
        h, weight = ctx.saved_tensors
        d_raw = d_raw.float().contiguous()
        dh = runtime.linear_small(d_raw, runtime.transpose(weight.detach()), None)     # [rows, 3] x [3 -> dim]
        dw = runtime.transpose(runtime.smallk_wgrad(h, d_raw))                          # ([dim, 3])^T
        return (dh, *runtime.deliver_grads(list(zip(ctx.params, (dw, runtime.colsum(d_raw))))))


class FlowLossFunction(torch.autograd.Function):
    """The flow-matching loss of temporal_adaptor.py:145-146 on the predictor's raw output (value from `runtime.flow_finish`);
    also hands out the kernel's other two results - x_pred = (x0 + pred_flow) * mask (:150, a no-grad value in the reference) and
    the duration estimate clamp(exp(x_pred[..., 0]) - 1, 0) - as non-differentiable outputs."""

    @staticmethod
    def forward(ctx, raw: Tensor, flow: Tensor, x0: Tensor, mask: Tensor):
        raw = raw.float().contiguous()
        pred, dur, _, loss = runtime.flow_finish(raw, flow, x0, mask)
        ctx.save_for_backward(raw, flow, mask)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(pred, dur)
        return loss.reshape(()), pred, dur

    @staticmethod
    def backward(ctx, g: Tensor, _dp=None, _dd=None):
        raw, flow, mask = ctx.saved_tensors
        return runtime.scale_(runtime.flow_loss_bwd(raw, flow, mask), g.reshape(1)), None, None, None


def adaptive_stack_parameters(tr: Transformer) -> list:
    ps = []
    for layer in tr.layers:
        att, ff = layer.attention, layer.feed_forward
        ps += [att.to_q.weight, att.to_kv.weight, att.rel_pos.learned_logslopes, att.to_out.weight, ff.net[0].weight, ff.net[3].weight]
    return ps + [tr.norm.weight, tr.norm.bias]


class AdaptiveStackFunction(torch.autograd.Function):
    """The adaptive-norm stack (transformer.py:174-211 with AdaptiveLayerNorm, normalization.py:37-61) on an already projected
    input, as one autograd node: like `stack.TransformerStackFunction`, the two norms of a layer taking their per-utterance
    scale / shift rows from `ss` (AdaProjectionFunction) and returning those rows' gradients."""

    @staticmethod
    def forward(ctx, tr: Transformer, x: Tensor, mask: Optional[Tensor], amp: bool, ss: Tensor, ctx_key_len: Optional[Tensor],
                *params: Tensor):
        x, ss = x.float().contiguous(), ss.float().contiguous()
        B, L, D = x.shape
        key_len = ctx_key_len if ctx_key_len is not None else (mask.sum(dim=1) if mask is not None else None)
        base_seed = runtime.draw_seed()
        images = layer_images(tr, amp)
        tape, out = [], x
        for li, layer in enumerate(tr.layers):
            att, ff = layer.attention, layer.feed_forward
            im = images[li]
            wqkv, wo, w1, w2 = (None,) * 4 if amp else (im["wqkv"], im["wo"], im["w1"], im["w2"])
            wqkv16, wo16, w116, w216 = (im["wqkv"], im["wo"], im["w1"], im["w2"]) if amp else (None,) * 4
            slopes = im["slopes"]
            s1, t1 = ss[:, (4 * li) * D:(4 * li + 1) * D], ss[:, (4 * li + 1) * D:(4 * li + 2) * D]
            s2, t2 = ss[:, (4 * li + 2) * D:(4 * li + 3) * D], ss[:, (4 * li + 3) * D:(4 * li + 4) * D]
            adt = torch.bfloat16 if amp else torch.float32        # dtype of the tensors that are GEMM / attention operands only
            h = runtime.layernorm(out, None, None, s1, t1, L, None, layer.attention_norm.eps, out_dtype=adt)
            qkv = _mm(h, wqkv, wqkv16, out_dtype=adt)
            p_att = float(att.attend.dropout) if layer.training else 0.0
            p_ff = float(ff.dropout_p) if layer.training else 0.0
            seed_att, seed_ff = base_seed + 2 * li, base_seed + 2 * li + 1
            lse = None
            if p_att > 0 or amp:
                o, lse = runtime.alibi_mqa_attention_train(qkv, att.heads, slopes, key_len, p_att, seed_att)
            else:
                o = runtime.alibi_mqa_attention(qkv, att.heads, slopes, key_len)
            x1 = _mm(o, wo, wo16, resid=out, mask=mask, flags=runtime.EP_MASK_ACC if mask is not None else 0)
            h2 = runtime.layernorm(x1, None, None, s2, t2, L, mask, layer.feed_forward_norm.eps, out_dtype=adt)
            if amp and h2.shape[-1] in (256, 384):      # one launch: u (bf16, autocast's Linear output) and a = dropout(gelu(u))
                u, a = runtime.gemm_gelu_train(h2, w116, p_ff, seed_ff)
            else:
                u = _mm(h2, w1, w116, out_dtype=adt)
                a = runtime.gelu(u, p_ff, seed_ff, out_dtype=adt)
            y = _mm(a, w2, w216, resid=x1, mask=mask, flags=runtime.EP_MASK_OUT if mask is not None else 0)
            tape.append((out, h, qkv, o, x1, h2, u, a, lse, p_att, seed_att, p_ff, seed_ff))
            out = y
        final = runtime.layernorm(out, tr.norm.weight, tr.norm.bias, row_mask=mask, eps=tr.norm.eps)
        ctx.tr, ctx.mask, ctx.key_len, ctx.tape, ctx.last, ctx.amp, ctx.ss, ctx.params = tr, mask, key_len, tape, out, amp, ss, params
        return final

    @staticmethod
    def backward(ctx, dfinal: Tensor):
        tr, mask, key_len, amp, ss = ctx.tr, ctx.mask, ctx.key_len, ctx.amp, ctx.ss
        B, L, D = ctx.last.shape
        mflag = runtime.EP_MASK_OUT if mask is not None else 0
        d_ss = torch.empty_like(ss)
        grads: list = []
        dy, dgf, dbf = runtime.layernorm_bwd(ctx.last, dfinal.float().contiguous(), tr.norm.weight, row_mask=mask, eps=tr.norm.eps)
        images = layer_images(tr, amp)
        for li in reversed(range(len(tr.layers))):
            layer = tr.layers[li]
            xin, h, qkv, o, x1, h2, u, a, lse, p_att, seed_att, p_ff, seed_ff = ctx.tape[li]
            att, ff = layer.attention, layer.feed_forward
            im = images[li]
            slopes = im["slopes"]
            wqkv_t, wo_t, w1_t, w2_t = (None,) * 4 if amp else (im["wqkv_t"], im["wo_t"], im["w1_t"], im["w2_t"])
            wqkv_t16, wo_t16, w1_t16, w2_t16 = (im["wqkv_t"], im["wo_t"], im["w1_t"], im["w2_t"]) if amp else (None,) * 4
            c = 4 * li * D
            gdt = torch.bfloat16 if amp else torch.float32        # as in stack.py: GEMM-only tensors live in bf16 under AMP
            dyg = runtime.cast_bf16(dy) if amp else dy
            wmask = None if amp else mask          # (padded rows of dy / dx1 are exactly zero: stack.py)
            dw2 = runtime.gemm_tn(dyg, a, row_mask=wmask, bf16=amp)
            if amp and dyg.shape[-1] in (256, 384):
                du = runtime.gemm_gelu_bwd(dyg, w2_t16, u, mask, p_ff, seed_ff)
            else:
                da = _mm(dyg, w2_t, w2_t16, out_dtype=gdt, mask=mask, flags=mflag)
                du = runtime.gelu_bwd(da, u, out=da, dropout_p=p_ff, seed=seed_ff)
            dw1 = runtime.gemm_tn(du, h2, bf16=amp)
            dh2 = _mm(du, w1_t, w1_t16)
            dx1 = runtime.adaln_bwd(x1, dh2, ss[:, c + 2 * D:c + 3 * D], mask, dy, True, d_ss[:, c + 2 * D:c + 3 * D],
                                    d_ss[:, c + 3 * D:c + 4 * D], layer.feed_forward_norm.eps)
            dx1g = runtime.cast_bf16(dx1) if amp else dx1
            dwo = runtime.gemm_tn(dx1g, o, row_mask=wmask, bf16=amp)
            d_o = _mm(dx1g, wo_t, wo_t16, out_dtype=gdt, mask=mask, flags=mflag)
            dqkv, dls = runtime.alibi_mqa_attention_bwd(qkv, o, d_o, att.heads, slopes, key_len, lse=lse, dropout_p=p_att,
                                                        seed=seed_att)
            dwqkv = runtime.gemm_tn(dqkv, h, bf16=amp)
            dh = _mm(dqkv, wqkv_t, wqkv_t16)
            dy = runtime.adaln_bwd(xin, dh, ss[:, c:c + D], None, dx1, True, d_ss[:, c:c + D], d_ss[:, c + D:c + 2 * D],
                                   layer.attention_norm.eps)
            hq = att.heads * 64
            ls = att.rel_pos.learned_logslopes
            grads = [dwqkv[:hq], dwqkv[hq:], dls[:ls.numel()].view_as(ls), dwo, dw1, dw2] + grads
        ctx.tape = None
        return (None, dy, None, None, d_ss, None, *runtime.deliver_grads(list(zip(ctx.params, grads + [dgf, dbf]))))


def flow_predictor_loss(pred, cond: Tensor, targets: Tensor, mask: Tensor, noise: Tensor, time_steps: Tensor,
                        amp: bool = False, key_len: Optional[Tensor] = None, return_pred: bool = False):
    """The flow loss of `FlowTransformerTemporalModule.forward` (temporal_adaptor.py:105-147) as a differentiable scalar:
    gradients reach every parameter of the predictor `pred` and `cond` (the encoder output; pass it detached for
    `detach_inputs`).  `targets` [B, L, 3] are constants (:112), `noise` / `time_steps` the step's x0 and t.
    `return_pred`: -> (loss, x_pred [B, L, 3], duration estimate [B, L]) - the module's other (no-grad) outputs (:148-152)."""
    tr = pred.transformer
    if mask.ndim == 3:
        mask = mask[..., 0]
    te, fe = pred.time_embedding, pred.time_embedding.freq_emb
    x_t, flow = runtime.flow_mix(noise, targets.detach(), time_steps, pred.sigma)
    time_emb = TimeEmbeddingFunction.apply(time_steps, fe.inv_freq, fe.freq_scale, te.mlp[0].weight, te.mlp[0].bias,
                                           te.mlp[2].weight, te.mlp[2].bias)
    ss = AdaProjectionFunction.apply(tr, time_emb, *ada_parameters(tr))
    proj = ProjectSplitFunction.apply(x_t, cond, tr.project_emb.weight, tr.project_emb.bias)
    out = AdaptiveStackFunction.apply(tr, proj, mask, amp, ss, key_len, *adaptive_stack_parameters(tr))
    raw = SmallOutputLinearFunction.apply(out, pred.linear_layer.weight, pred.linear_layer.bias)
    loss, x_pred, duration = FlowLossFunction.apply(raw, flow, noise, mask)
    return (loss, x_pred, duration) if return_pred else loss
