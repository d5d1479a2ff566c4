"""`MelLoss` (models/acoustic/loss.py:22-35) with forward value and gradient from one kernel."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import runtime


class _MelLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mel_out: Tensor, mel_target: Tensor, mel_len: Tensor):
        # value and d loss / d mel_out in the same pass (the gradient is scaled by the incoming grad in backward)
        loss, grad = runtime.mel_loss(mel_out, mel_target, mel_len, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss: Tensor):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None


class MelLoss(torch.nn.Module):
    """loss.py:22-35: masked mean-squared error per utterance (utils/functions.py:44-58), mean over the batch, times
    `weight`; `skip_steps` as modules/loss.py:27-31."""

    def __init__(self, weight: float = 1.0, skip_steps: int = 0):
        super().__init__()
        self.weight, self.skip_steps = weight, skip_steps

    def forward(self, mels_out: Tensor, mels_target: Tensor, mel_lengths: Tensor, step=None):
        if step is not None and step < self.skip_steps:
            return 0.
        return self.weight * _MelLossFunction.apply(mels_out, mels_target, mel_lengths)


class _BinLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, soft: Tensor, hard: Tensor, eps: float):
        loss, grad = runtime.attn_bin_loss(soft, hard, eps, want_grad=True)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss: Tensor):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None


class AttentionBinarizationLoss(torch.nn.Module):
    """loss.py:80-107: -log(clamp(attn_soft[attn_hard == 1], eps)).sum() / attn_hard.sum(), times `weight`;
    `attn_hard` is the aligner's int16 one-hot MAS output."""

    def __init__(self, weight: float = 1.0, skip_steps: int = 0, eps: float = 1e-6):
        super().__init__()
        self.weight, self.skip_steps, self.eps = weight, skip_steps, eps

    def forward(self, soft_attention: Tensor, hard_attention: Tensor, step=None):
        if step is not None and step < self.skip_steps:
            return 0.
        soft = soft_attention.reshape(-1, *soft_attention.shape[-2:])
        hard = hard_attention.reshape(-1, *hard_attention.shape[-2:])
        return self.weight * _BinLossFunction.apply(soft, hard, self.eps)


class _CTCLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, text_len: Tensor, mel_len: Tensor, blank: float):
        loss, grad = runtime.attn_ctc_loss(logits, text_len, mel_len, blank, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss: Tensor):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None, None


class AttentionCTCLoss(torch.nn.Module):
    """loss.py:39-77: CTC ("forward-sum") loss of the aligner's logits against the text positions 1 .. text_len, a blank class
    with logit `blank_logprob` in front, nn.CTCLoss(zero_infinity=True) semantics, times `weight`."""

    def __init__(self, blank_logprob: float = -1, weight: float = 1.0, skip_steps: int = 0):
        super().__init__()
        self.blank_logprob, self.weight, self.skip_steps = blank_logprob, weight, skip_steps

    def forward(self, attn_logits: Tensor, text_lengths: Tensor, mel_lengths: Tensor, step=None):
        if step is not None and step < self.skip_steps:
            return 0.
        return self.weight * _CTCLossFunction.apply(attn_logits, text_lengths, mel_lengths, float(self.blank_logprob))
