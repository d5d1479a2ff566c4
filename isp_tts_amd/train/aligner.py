"""Training backward of the aligner front-end (`ConvAttention`, alignment.py:159-208) and of the two adaptor steps through
which the mel loss reaches it (`TemporalAverager` :446-449, the alignment operand of `LengthRegulator` :419-421).

Forward launches are the inference path's, with the GELUs of the convolution blocks as separate passes (their
pre-activations are kept).  Buffers are channel-last and padded, [B][T+4][C]: "padded" buffers hold frame t at row t + 2
(two zero rows either side), convolution OUTPUTS hold frame t at row t, their last four rows per utterance being scratch.
Backward of a block Conv1d(k) -> GELU -> masked instance norm (-> next block's mask):
    d(norm out) -> ispk_masked_instnorm_bwd_f32 -> ispk_gelu_bwd_f32 -> d u (zero past the utterance's length)
    d W[o][(k, c)] = sum_rows d u[row][o] * window(row)[(k, c)]          one ispk_gemm_tn_f32 over the overlapping rows
    d x = the same conv GEMM run over the padded d u with flipped taps       (only the second query block needs it)
"""
from __future__ import annotations

import torch
from torch import Tensor

from .. import runtime


def _windows(xpad: Tensor, rows: int, taps: int, first_row: int = 0) -> Tensor:
    """[rows, taps * C] view of a padded channel-last buffer: row r = the `taps` consecutive buffer rows from first_row + r."""
    C = xpad.shape[-1]
    flat = xpad.reshape(-1, C)
    return torch.as_strided(flat, (rows, taps * C), (C, 1), flat.storage_offset() + first_row * C)


def _w2d(conv_weight: Tensor) -> Tensor:
    """Conv1d weight [O, C, k] -> GEMM weight [O, k * C] (tap-major, as ConvAttention._staged)."""
    w = conv_weight.detach()
    if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous():
        return runtime.permute021(w).reshape(w.shape[0], -1)              # a libispk launch (re-done after every update)
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def _w2d_grad(g: Tensor, conv_weight: Tensor) -> Tensor:
    o, c, k = conv_weight.shape
    if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous():
        return runtime.permute021(g.view(o, k, c))
    return g.view(o, k, c).permute(0, 2, 1).contiguous()


class ConvAttentionFunction(torch.autograd.Function):
    """(attn_soft, attn_logits) = ConvAttention(mel, keys) with its 13 parameters as differentiable inputs.  The keys (the
    detached encoder output, model.py:139) and the mel spectrogram get no gradient."""

    @staticmethod
    def forward(ctx, att, mel: Tensor, keys_t: Tensor, mel_len: Tensor, text_len: Tensor, amp: bool, *params: Tensor):
        """`amp`: the five convolutions (and their backward GEMMs) take bf16 operands - autocast covers conv1d in the reference's
        step (recipes/default.yaml:56) - with fp32 outputs; GELU, the instance norms and the scores stay fp32."""
        kb0, kb1 = att.key_proj
        qb0, qb1, qb2 = att.query_proj
        M, L = mel.shape[2], keys_t.shape[2]
        dt = torch.bfloat16 if amp else torch.float32
        cw = (lambda w: runtime.cast_bf16(w)) if amp else (lambda w: w)
        wk0, wk1 = cw(_w2d(kb0.conv.weight)), cw(_w2d(kb1.conv.weight))
        wq0, wq1, wq2 = cw(_w2d(qb0.conv.weight)), cw(_w2d(qb1.conv.weight)), cw(_w2d(qb2.conv.weight))
        kp0 = runtime.pad_rows(keys_t.float(), text_len, channel_first=True, out_dtype=dt)
        u0 = runtime.conv5_padded(kp0, wk0)
        y0 = runtime.gelu(u0)
        kp1 = runtime.masked_instnorm(y0, kb0.norm.weight, kb0.norm.bias, text_len, out_dtype=dt)
        k_enc = runtime.conv5_padded(kp1, wk1)
        qp0 = runtime.pad_rows(mel.float(), mel_len, channel_first=True, out_dtype=dt)
        u1 = runtime.conv5_padded(qp0, wq0)
        y1 = runtime.gelu(u1)
        qp1 = runtime.masked_instnorm(y1, qb0.norm.weight, qb0.norm.bias, mel_len, out_dtype=dt)
        u2 = runtime.conv5_padded(qp1, wq1)
        y2 = runtime.gelu(u2)
        qp2 = runtime.masked_instnorm(y2, qb1.norm.weight, qb1.norm.bias, mel_len, out_dtype=dt)
        q_enc = runtime.conv5_padded(qp2, wq2)
        soft, logits = runtime.aligner_scores(q_enc, k_enc, text_len, mel_len, M, L)
        ctx.att, ctx.M, ctx.L, ctx.amp = att, M, L, amp
        ctx.save_for_backward(mel_len, text_len, kp0, u0, y0, kp1, k_enc, qp0, u1, y1, qp1, u2, y2, qp2, q_enc, soft, logits)
        return soft, logits

    @staticmethod
    def backward(ctx, d_soft, d_logits):
        att, M, L, amp = ctx.att, ctx.M, ctx.L, ctx.amp
        cw = (lambda w: runtime.cast_bf16(w)) if amp else (lambda w: w)          # GEMM operands: bf16 under AMP
        (mel_len, text_len, kp0, u0, y0, kp1, k_enc, qp0, u1, y1, qp1, u2, y2, qp2, q_enc, soft, logits) = ctx.saved_tensors
        kb0, kb1 = att.key_proj
        qb0, qb1, qb2 = att.query_proj
        B = soft.shape[0]
        dev = soft.device
        dS, dSt = runtime.aligner_scores_bwd(logits, soft, d_soft, d_logits, text_len, mel_len, att.scale)
        # d q_enc[b] = dS[b] k_enc[b], d k_enc[b] = dS[b]^T q_enc[b]: batched transposed products straight into zeroed
        # conv-output-space buffers (frame t at row t; the rows past T stay / become zero: the padding columns of dS are zero)
        dq_buf = runtime.zeros((B, M + 4, 128), torch.float32, dev)
        dk_buf = runtime.zeros((B, L + 4, 128), torch.float32, dev)
        runtime.gemm_tn_batched(dSt, k_enc[:, :L], out=dq_buf[:, :dSt.shape[2]])
        runtime.gemm_tn_batched(dS, q_enc[:, :M], out=dk_buf[:, :dS.shape[2]])

        def conv1_bwd(d_buf: Tensor, xpad: Tensor, conv_weight: Tensor):
            """1 x 1 convolution reading frame t at padded row t + 2: -> (d W [O, C, 1], d x in conv-output row space)."""
            rows = d_buf.shape[0] * d_buf.shape[1] - 2
            dg = cw(d_buf.reshape(-1, d_buf.shape[2]))
            dw = runtime.gemm_tn(dg[:rows], _windows(xpad, rows, 1, first_row=2))
            w = _w2d(conv_weight)                                                    # [O, C]
            dx = runtime.gemm(dg, cw(runtime.transpose(w)), out_dtype=torch.float32).view(d_buf.shape[0], d_buf.shape[1], -1)
            return _w2d_grad(dw, conv_weight), dx

        def block_bwd(d_normed: Tensor, y: Tensor, u: Tensor, xpad: Tensor, block, lengths: Tensor, want_dx: bool):
            """Conv1d(5) -> GELU -> masked instance norm, given d(norm out) in conv-output row space."""
            d_y, dnw, dnb = runtime.masked_instnorm_bwd(y, d_normed.contiguous(), block.norm.weight, lengths)
            d_u = runtime.gelu_bwd(d_y, u, out=d_y)
            Bc, TP, O = d_u.shape
            rows = Bc * TP - 4
            dw = runtime.gemm_tn(cw(d_u.reshape(-1, O))[:rows], _windows(xpad, rows, 5))
            dx = None
            if want_dx:
                # d xpad[r] = sum_k d u[r - k] W_k: the conv GEMM over d u with four zero rows in front and the taps flipped
                C = xpad.shape[-1]
                g = runtime.zeros((Bc * TP + 4 + 4, O), torch.bfloat16 if amp else torch.float32, d_u.device)
                runtime.segments([(d_u.reshape(-1, O), g[4:4 + Bc * TP], runtime.SEG_BF16 if amp else runtime.SEG_COPY)])
                wf = cw(runtime.conv_weight_flip(block.conv.weight))      # [C][(j, o)] = W[o][c][4 - j]
                dx = runtime.gemm(_windows(g, Bc * TP, 5), wf, out_dtype=torch.float32)   # [B (T+4), C]: padded row space
            return _w2d_grad(dw, block.conv.weight), dnw, dnb, dx

        # key side: conv1 (768 -> 128), then the first block (its input, the detached encoder output, needs no gradient)
        dwk1, d_kp1 = conv1_bwd(dk_buf, kp1, kb1.conv.weight)
        dwk0, dnk_w, dnk_b, _ = block_bwd(d_kp1, y0, u0, kp0, kb0, text_len, False)
        # query side: conv1 (80 -> 128), block 2 (160 -> 80, passes a gradient on), block 1 (80 -> 160)
        dwq2, d_qp2 = conv1_bwd(dq_buf, qp2, qb2.conv.weight)
        dwq1, dnq1_w, dnq1_b, d_qp1 = block_bwd(d_qp2, y2, u2, qp1, qb1, mel_len, True)
        # d qp1 is in PADDED row space (frame t at row t + 2): shift by two rows into the conv-output convention
        Bq, TPq = qp1.shape[0], qp1.shape[1]
        shifted = runtime.zeros((Bq * TPq + 2, qp1.shape[2]), torch.float32, dev)
        runtime.segments([(d_qp1.reshape(Bq * TPq, -1), shifted[:Bq * TPq], runtime.SEG_COPY)])
        d_n1 = shifted[2:].view(Bq, TPq, -1)
        dwq0, dnq0_w, dnq0_b, _ = block_bwd(d_n1, y1, u1, qp0, qb0, mel_len, False)
        grads = {id(kb0.conv.weight): dwk0, id(kb0.norm.weight): dnk_w, id(kb0.norm.bias): dnk_b, id(kb1.conv.weight): dwk1,
                 id(qb0.conv.weight): dwq0, id(qb0.norm.weight): dnq0_w, id(qb0.norm.bias): dnq0_b,
                 id(qb1.conv.weight): dwq1, id(qb1.norm.weight): dnq1_w, id(qb1.norm.bias): dnq1_b, id(qb2.conv.weight): dwq2}
        ps = aligner_parameters(att)
        return (None, None, None, None, None, None, *runtime.deliver_grads([(p, grads[id(p)]) for p in ps]))


def aligner_parameters(att) -> list:
    kb0, kb1 = att.key_proj
    qb0, qb1, qb2 = att.query_proj
    return [kb0.conv.weight, kb0.norm.weight, kb0.norm.bias, kb1.conv.weight, qb0.conv.weight, qb0.norm.weight, qb0.norm.bias,
            qb1.conv.weight, qb1.norm.weight, qb1.norm.bias, qb2.conv.weight]


def conv_attention_train(att, mel: Tensor, keys_t: Tensor, mel_len: Tensor, text_len: Tensor, amp: bool = False):
    """-> (attn_soft, attn_logits), differentiable with respect to the aligner's parameters; `amp`: bf16 convolution operands."""
    return ConvAttentionFunction.apply(att, mel, keys_t, mel_len, text_len, amp, *aligner_parameters(att))


class SoftAverageFunction(torch.autograd.Function):
    """feats[b][l] = (0, soft average of pitch, of energy) over attn_soft (TemporalAverager, temporal_adaptor.py:446-449);
    gradient flows to attn_soft only (the dense targets are data)."""

    @staticmethod
    def forward(ctx, attn_soft: Tensor, pitch: Tensor, energy: Tensor, text_len: Tensor):
        ctx.save_for_backward(attn_soft, pitch, energy, text_len)
        return runtime.soft_average(attn_soft, pitch, energy, None, text_len)

    @staticmethod
    def backward(ctx, d_feats: Tensor):
        attn_soft, pitch, energy, text_len = ctx.saved_tensors
        return runtime.soft_average_bwd(attn_soft, pitch, energy, d_feats, text_len), None, None, None
