"""The teacher-forced training forward of `AcousticModel` (model.py:116-174) with the loss terms the built backward covers.

`acoustic_train_outputs` is the forward itself (what `AcousticModel.forward` returns when gradients are enabled: the reference's
training loop drives it unchanged); `acoustic_train_forward` adds the reference's dict of losses (`AcousticModelLoss`, loss.py:140-182):
  * "model/mel_loss" and "adaptor/flow_loss" are differentiable - their gradients reach every parameter outside the aligner:
    text embedding, TextEncoder, the adaptor's embedding module and flow predictor, MelDecoder, to_mel;
  * "aligner/attention_loss" (CTC) and "aligner/kl_loss" (binarisation) train the aligner front-end (train/aligner.py), which
    the mel loss also reaches through attn_soft in the length regulator (the pitch / energy averages are detached, as in the
    reference); with `train_aligner=False` the aligner is frozen and the two terms are values.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .. import runtime
from .loss import AttentionBinarizationLoss, AttentionCTCLoss, MelLoss, sum_losses
from .aligner import conv_attention_train
from .predictor import flow_predictor_loss
from .stack import (EmbedTokensFunction, LengthRegulateFunction, MaskedLinearResidualFunction, ToMelFunction, fork,
                    transformer_train_forward)


def acoustic_train_outputs(model, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Tensor, energy: Tensor,
                           flow_noise: Optional[Tensor] = None, flow_time: Optional[Tensor] = None, amp: bool = False,
                           train_aligner: bool = True):
    """`AcousticModel.forward` (model.py:116-174) as a chain of autograd nodes whose forward AND backward are HIP kernels ->
    the reference's `AcousticModelOutput`: `mel`, `adaptor_output.losses["flow_loss"]`, `aligner_output.attn_logits` and
    `aligner_output.attn_soft` carry the graph, so that the reference's loop body - `outputs = model(**inputs)`;
    `loss, losses = criterion(inputs=inputs, outputs=outputs)`; `optimizer.step(loss)` (experiments/trainer.py:543-549) -
    back-propagates through these kernels.  This is what `AcousticModel.forward` returns when gradients are enabled.
    `train_aligner=False`: the aligner front-end is frozen (its outputs are values)."""
    from ..acoustic.alignment import AlignerOutput
    from ..acoustic.model import AcousticModelOutput
    from ..acoustic.temporal_adaptor import TemporalAdaptorOutput
    ad = model.temporal_adaptor
    emb, enc_mask = EmbedTokensFunction.apply(text, model.text_embedding.weight, text_len)
    enc_out = transformer_train_forward(model.encoder, emb, enc_mask, amp, key_len=text_len)
    keys_t = enc_out.detach().transpose(1, 2)          # model.py:139: the aligner sees the DETACHED encoder output
    if train_aligner:
        attn_soft, attn_logits = conv_attention_train(model.aligner.attention, mel, keys_t, mel_len, text_len, amp)
        attn_soft, attn_soft_kl = fork(attn_soft)     # two consumers: the length regulator and the binarisation loss
    with torch.no_grad():
        if not train_aligner:
            attn_soft, attn_logits = model.aligner.attention(mel, keys_t, mel_len, text_len)
            attn_soft_kl = attn_soft
        attn_hard, dur = model.aligner.binarize_attention_parallel(attn_logits.detach(), text_len, mel_len, return_duration=True)
        targets = runtime.soft_average(attn_soft.detach(), pitch, energy, dur, text_len)       # [log1p duration, pitch, energy]
    # The averaged pitch / energy enter the embedding stack DETACHED (temporal_adaptor.py:284, :292 `pitch_target.detach()`,
    # `energy_target.detach()`), like the predictor's targets (:112): the mel loss reaches attn_soft - and through it the
    # aligner - only by way of the length regulator (:300).
    feats = targets
    b, l = text.shape
    x0 = flow_noise if flow_noise is not None else torch.randn(b, l, 3, device=text.device)
    t = flow_time if flow_time is not None else torch.rand(b, device=text.device)
    if ad.predictor.detach_inputs:
        cond = enc_out.detach()
    else:
        enc_out, cond = fork(enc_out)               # two consumers: the embedding module's residual and the flow predictor
    flow_loss, x_pred, duration_pred = flow_predictor_loss(ad.predictor, cond, targets, enc_mask, x0, t, amp, key_len=text_len,
                                                           return_pred=True)
    emod = ad.embedding
    # pitch / energy columns as a contiguous [B, L, 2] (a strided copy launch, not a .contiguous())
    pe = runtime.copy2d(feats.reshape(-1, 3)[:, 1:3], torch.empty((b * l, 2), dtype=torch.float32, device=feats.device)).view(b, l, 2)
    h = transformer_train_forward(emod.transformer, pe, enc_mask, amp, key_len=text_len)
    x = MaskedLinearResidualFunction.apply(h, emod.linear_layer.weight, emod.linear_layer.bias, enc_mask, enc_out)
    dec_in, dec_len, dec_mask = LengthRegulateFunction.apply(x, attn_soft, mel_len.view(-1, 1), mel.shape[2])
    dec = transformer_train_forward(model.decoder, dec_in, dec_mask, amp, key_len=dec_len)
    mel_out = ToMelFunction.apply(dec, model.to_mel.weight, model.to_mel.bias, dec_mask, amp)
    adaptor = TemporalAdaptorOutput(enc_out=dec_in, log_duration=x_pred[..., 0], duration=duration_pred, dec_lengths=dec_len,
                                    pitch=x_pred[..., 1], energy=x_pred[..., 2], pitch_target=feats[..., 1],
                                    energy_target=feats[..., 2], losses={"flow_loss": flow_loss}, dec_mask=dec_mask)
    aligner = AlignerOutput(attn_soft=attn_soft_kl, attn_logits=attn_logits, attn_hard=attn_hard, attn_hard_duration=dur)
    return AcousticModelOutput(mel=mel_out, adaptor_output=adaptor, aligner_output=aligner)


def acoustic_train_forward(model, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Tensor, energy: Tensor,
                           flow_noise: Optional[Tensor] = None, flow_time: Optional[Tensor] = None, amp: bool = False,
                           train_aligner: bool = True):
    """-> (mel_out [B, 80, M], loss, losses): `acoustic_train_outputs` under the reference's total loss.  `train_aligner`: the
    aligner front-end is a differentiable node too (train/aligner.py) and loss = mel + flow + CTC + binarisation, the
    reference's total (loss.py:140-182); otherwise the aligner is frozen, loss = mel + flow and the two attention terms are
    values."""
    out = acoustic_train_outputs(model, text, text_len, mel, mel_len, pitch, energy, flow_noise, flow_time, amp, train_aligner)
    al = out.aligner_output
    mel_loss = MelLoss()(out.mel, mel, mel_len)
    flow_loss = out.adaptor_output.losses["flow_loss"]
    with torch.set_grad_enabled(train_aligner):
        ctc = AttentionCTCLoss()(al.attn_logits, text_len, mel_len)
        kl = AttentionBinarizationLoss()(al.attn_soft, al.attn_hard)
    losses = {"model/mel_loss": mel_loss, "adaptor/flow_loss": flow_loss, "aligner/attention_loss": ctc, "aligner/kl_loss": kl}
    total = sum_losses([mel_loss, flow_loss, ctc, kl] if train_aligner else [mel_loss, flow_loss])
    return out.mel, total, losses
