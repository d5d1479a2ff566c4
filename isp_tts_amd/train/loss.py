"""The loss terms of `AcousticModelLoss` (models/acoustic/loss.py:22-182), value and gradient of each from kernels."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .. import runtime


def _times(grad: Tensor, grad_loss: Tensor) -> Tensor:
    """grad * grad_loss (the scalar that reaches a loss term in backward), in place on the saved gradient, as a libispk launch."""
    if grad.is_cuda and grad.dtype == torch.float32 and grad.is_contiguous() and grad_loss.dtype == torch.float32:
        return runtime.scale_(grad, grad_loss.reshape(1))
    return grad * grad_loss


def _weighted(weight: float, loss: Tensor) -> Tensor:
    return loss if weight == 1.0 else weight * loss


class _SumLossesFunction(torch.autograd.Function):
    """total = sum of scalar loss terms (loss.py:140-182 adds them one by one): one launch; every term gets the incoming gradient."""

    @staticmethod
    def forward(ctx, *terms: Tensor):
        ctx.n = len(terms)
        return runtime.sum_scalars([t.reshape(1) for t in terms])

    @staticmethod
    def backward(ctx, g: Tensor):
        return (g,) * ctx.n


def sum_losses(terms) -> Tensor:
    """Sum of scalar loss terms (Python numbers - skipped criteria return 0. - are added on the host side of the result)."""
    tensors = [t for t in terms if isinstance(t, Tensor)]
    extra = sum(float(t) for t in terms if not isinstance(t, Tensor))
    if not tensors:
        return extra
    if all(t.is_cuda and t.dtype == torch.float32 for t in tensors) and len(tensors) <= 8:
        total = _SumLossesFunction.apply(*tensors) if len(tensors) > 1 else tensors[0]
    else:
        total = tensors[0]
        for t in tensors[1:]:
            total = total + t
    return total if extra == 0.0 else total + extra


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
        return _times(grad, grad_loss), None, None


class MelLoss(torch.nn.Module):
    """loss.py:22-35: masked mean-squared error per utterance (utils/functions.py:44-58), mean over the batch, times
    `weight`; `skip_steps` as modules/loss.py:27-31."""

    def __init__(self, weight: float = 1.0, skip_steps: int = 0):
        super().__init__()
        self.weight, self.skip_steps = weight, skip_steps

    def forward(self, mels_out: Tensor, mels_target: Tensor, mel_lengths: Tensor, step=None):
        if step is not None and step < self.skip_steps:
            return 0.
        return _weighted(self.weight, _MelLossFunction.apply(mels_out, mels_target, mel_lengths))


class _BinLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, soft: Tensor, hard: Tensor, eps: float):
        loss, grad = runtime.attn_bin_loss(soft, hard, eps, want_grad=True)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss: Tensor):
        (grad,) = ctx.saved_tensors
        return _times(grad, grad_loss), None, None


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
        return _weighted(self.weight, _BinLossFunction.apply(soft, hard, self.eps))


class _CTCLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, text_len: Tensor, mel_len: Tensor, blank: float):
        loss, grad = runtime.attn_ctc_loss(logits, text_len, mel_len, blank, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss: Tensor):
        (grad,) = ctx.saved_tensors
        return _times(grad, grad_loss), None, None, None


class AttentionCTCLoss(torch.nn.Module):
    """loss.py:39-77: CTC ("forward-sum") loss of the aligner's logits against the text positions 1 .. text_len, a blank class
    with logit `blank_logprob` in front, nn.CTCLoss(zero_infinity=True) semantics, times `weight`."""

    def __init__(self, blank_logprob: float = -1, weight: float = 1.0, skip_steps: int = 0):
        super().__init__()
        self.blank_logprob, self.weight, self.skip_steps = blank_logprob, weight, skip_steps

    def forward(self, attn_logits: Tensor, text_lengths: Tensor, mel_lengths: Tensor, step=None):
        if step is not None and step < self.skip_steps:
            return 0.
        return _weighted(self.weight, _CTCLossFunction.apply(attn_logits, text_lengths, mel_lengths, float(self.blank_logprob)))


_DEFAULT = object()   # "argument not given" (None means: criterion disabled, as in the reference)


class AcousticModelLoss(torch.nn.Module):
    """`AcousticModelLoss` of models/acoustic/loss.py:122-182: mel loss + the adaptor's own losses (the flow loss) + attention CTC
    loss + attention binarisation loss, each a kernel; returns (loss, {"model/mel_loss", "adaptor/...", "aligner/attention_loss",
    "aligner/kl_loss"}) like the reference.  `inputs` needs `.mel`, `.mel_len`, `.text_len` (a dict or any object with those
    attributes: the collator's batch through `AcousticModel.prepare_inputs`); `outputs` is the model's `AcousticModelOutput`."""

    def __init__(self, mel_loss=_DEFAULT, attention_loss=_DEFAULT, attention_kl_loss=_DEFAULT):
        """Each argument is that criterion's config (a dict of its constructor arguments); as in the reference (loss.py:140-150)
        `attention_loss=None` / `attention_kl_loss=None` DISABLE the term, leaving the argument out takes the defaults."""
        super().__init__()
        cfg = lambda c: {} if c is _DEFAULT or c is None else dict(c)   # noqa: E731
        self.mel_criterion = MelLoss(**cfg(mel_loss))
        self.attention_criterion = None if attention_loss is None else AttentionCTCLoss(**cfg(attention_loss))
        self.attention_kl_criterion = None if attention_kl_loss is None else AttentionBinarizationLoss(**cfg(attention_kl_loss))

    def forward(self, inputs, outputs, step=None):
        get = (lambda k: inputs[k]) if isinstance(inputs, dict) else (lambda k: getattr(inputs, k))
        terms, losses = [], {}
        mel_loss = self.mel_criterion(outputs.mel, get("mel"), get("mel_len"), step=step)
        losses["model/mel_loss"] = mel_loss
        terms.append(mel_loss)
        if outputs.adaptor_output.losses is not None:
            for key, loss_i in outputs.adaptor_output.losses.items():
                losses[f"adaptor/{key}"] = loss_i
                terms.append(loss_i)
        if self.attention_criterion is not None:
            attn_loss = self.attention_criterion(outputs.aligner_output.attn_logits, get("text_len"), get("mel_len"), step=step)
            losses["aligner/attention_loss"] = attn_loss
            terms.append(attn_loss)
        if self.attention_kl_criterion is not None:
            kl = self.attention_kl_criterion(outputs.aligner_output.attn_soft, outputs.aligner_output.attn_hard, step=step)
            losses["aligner/kl_loss"] = kl
            terms.append(kl)
        return sum_losses(terms), losses
