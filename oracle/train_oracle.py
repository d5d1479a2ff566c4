"""ORACLE (test infrastructure, not product): the reference's training-step pieces, restated on CPU with plain PyTorch.

  * `reference_optimizer` / `reference_step`: experiments/optimizers.py:15-20, :34-40 (weight-decay grouping), :72-74
    (torch.optim.AdamW), :230-244 (clip_grad_norm_ on param_groups[0] only, step, zero_grad).
  * `mel_loss`: models/acoustic/loss.py:22-35 with utils/functions.py:44-58 (`masked_mean`).
  * `adamw_flat` / `sqnorm_flat`: the same AdamW arithmetic on flat arenas - the stand-in the CPU (gloo) test plugs into
    `FlatAdamW` so that the gradient exchange runs without a GPU.
Gradients of the transformer stacks come from autograd over `acoustic_oracle.transformer` (no separate restatement).
  * `acoustic_losses`: the reference's total objective (model.py:116-174 + loss.py:140-182) over the oracle's forward.
Pinned against the REAL reference: oracle/make_goldens.py runs the reference's `AcousticModel.forward`, `AcousticModelLoss`,
`.backward()`, `group_weight_decayable_params` and one clipped AdamW step and writes tests/golden/train.npz (losses, every
parameter's gradient norm + a strided sample, the decay grouping, the update of one step); tests/test_oracle_goldens.py holds
this module to it on CPU, tests/test_gpu_train.py the HIP step on the GPU.

Only tests/ may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def group_weight_decayable_params(params):
    """optimizers.py:15-20."""
    wd, no_wd = [], []
    for p in params:
        (no_wd if p.squeeze().ndim < 2 else wd).append(p)
    return wd, no_wd


def reference_optimizer(params, lr=2e-4, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8, group_wd_params=True):
    """optimizers.py:29-43 + :72-74: two param_groups when weight_decay > 0 (decay group first), else one."""
    params = list(params)
    if weight_decay > 0. and group_wd_params:
        wd, no_wd = group_weight_decayable_params(params)
        groups = [{"params": wd}, {"params": no_wd, "weight_decay": 0.}]
    else:
        groups = params
    return torch.optim.AdamW(groups, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)


def reference_step(opt: torch.optim.Optimizer, grad_clip=1.0):
    """optimizers.py:233-244 after backward: clip group 0, step, zero_grad -> grad norm (None if not finite)."""
    norm = None
    if grad_clip is not None:
        norm = torch.nn.utils.clip_grad_norm_(opt.param_groups[0]["params"], grad_clip)
        if torch.isnan(norm) or torch.isinf(norm):
            norm = None
    opt.step()
    opt.zero_grad()
    return norm


def mel_loss(mel_out: Tensor, mel_target: Tensor, mel_len: Tensor) -> Tensor:
    """loss.py:28-35: MSE (reduction none), mask frames >= mel_len, per-utterance sum / count, mean over the batch."""
    loss = F.mse_loss(mel_out, mel_target, reduction="none")
    mask = (torch.arange(mel_out.shape[-1])[None, :] < mel_len[:, None])[:, None].expand_as(mel_out)
    loss = loss.masked_fill(~mask, 0.)
    num = loss.sum(dim=-1).sum(dim=-1)
    den = mask.sum(dim=-1).sum(dim=-1)
    return (num / den.clamp(min=1e-5)).mean()


def attention_ctc_loss(attn_logits: Tensor, text_len: Tensor, mel_len: Tensor, blank_logprob: float = -1) -> Tensor:
    """loss.py:56-77: pad a blank class in front, log-softmax over classes, [T, B, C] layout, targets 1 .. len padded with
    0, nn.CTCLoss(zero_infinity=True) with its default "mean" reduction."""
    padded = F.pad(attn_logits, pad=(1, 0), value=blank_logprob)
    logprob = F.log_softmax(padded, dim=2).transpose(0, 1)
    ids = torch.arange(1, int(text_len.max()) + 1)[None].expand(text_len.numel(), -1).clone()
    ids[ids > text_len.unsqueeze(1)] = 0
    return torch.nn.CTCLoss(zero_infinity=True)(log_probs=logprob, targets=ids, input_lengths=mel_len, target_lengths=text_len)


def attention_binarization_loss(soft: Tensor, hard: Tensor, eps: float = 1e-6) -> Tensor:
    """loss.py:100-107."""
    log_sum = torch.log(torch.clamp(soft[hard == 1], min=eps)).sum()
    return -log_sum / hard.sum()


def acoustic_losses(sd: dict, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Tensor, energy: Tensor,
                    flow_x0: Tensor, flow_t: Tensor):
    """The reference's training objective on the oracle's forward, differentiable w.r.t. the tensors of `sd` that require
    grad: `AcousticModel.forward` (model.py:116-174) composed with its detach points - the aligner sees the DETACHED encoder
    output (:139), the flow predictor's targets are detached (temporal_adaptor.py:112), and the pitch / energy averages
    enter the embedding stack DETACHED (:284, :292), so the mel loss reaches attn_soft only through the length regulator
    (:300) - under `AcousticModelLoss.forward` (loss.py:140-182).  -> (total, {the reference's four loss keys}).
    Pinned against the reference's own forward + loss + backward by tests/golden/train.npz (oracle/make_goldens.py)."""
    from . import acoustic_oracle as orc
    from . import mas_oracle
    emb = F.embedding(text, sd["text_embedding.weight"], padding_idx=0)
    enc_mask = torch.arange(text.shape[1])[None, :] < text_len[:, None]
    m3 = enc_mask[..., None]
    enc_out = orc.transformer(sd, "encoder", emb, enc_mask)
    soft, logits = orc.conv_attention(sd, mel, enc_out.detach().transpose(1, 2), mel_len, text_len)
    hard = torch.from_numpy(mas_oracle.b_mas(logits.detach().numpy(), text_len.numpy(), mel_len.numpy()))
    dur = hard.sum(dim=1)
    if not torch.all(dur.sum(dim=1) == mel_len):               # alignment.py:278-282
        dur[:, 0] += mel_len - dur.sum(dim=1)
    pt = orc.soft_average(pitch[:, None], soft).transpose(1, 2) * m3
    et = orc.soft_average(energy[:, None], soft).transpose(1, 2) * m3
    targets = torch.cat([torch.log1p(dur.float())[..., None], pt, et], dim=-1)
    _, flow_loss = orc.predictor_forward(sd, enc_out, targets.detach(), enc_mask, flow_x0, flow_t)
    x = enc_out + orc.embedding_module(sd, torch.cat([pt.detach(), et.detach()], dim=-1), enc_mask)
    dec_lens = torch.clamp_max((dur.sum(dim=1).float() + 0.5).long(), mel.shape[2])
    dec_in = soft @ x
    dec_mask = torch.arange(mel.shape[2])[None, :] < dec_lens[:, None]
    dec = orc.transformer(sd, "decoder", dec_in, dec_mask)
    mel_out = F.linear(dec, sd["to_mel.weight"], sd["to_mel.bias"]).transpose(1, 2) * dec_mask[:, None]
    terms = {"model/mel_loss": mel_loss(mel_out, mel, mel_len), "adaptor/flow_loss": flow_loss,
             "aligner/attention_loss": attention_ctc_loss(logits, text_len, mel_len),
             "aligner/kl_loss": attention_binarization_loss(soft, hard)}
    return sum(terms.values()), terms


def sqnorm_flat(g: Tensor, out: Tensor) -> Tensor:
    out.copy_((g.double() ** 2).sum().float().reshape(1))
    return out


def adamw_flat(p, g, m, v, n_decay, lr, betas, eps, weight_decay, step, grad_sqnorm=None, max_norm=1.0, grad_scale=1.0):
    """torch.optim.AdamW's single-tensor update (torch/optim/adamw.py) on flat arenas, decay + clip on [0, n_decay)."""
    b1, b2 = betas
    g = g * grad_scale
    if grad_sqnorm is not None:
        coef = min(max_norm / (float(grad_sqnorm.sqrt()) * grad_scale + 1e-6), 1.0)
        g = torch.cat([g[:n_decay] * coef, g[n_decay:]])
    p[:n_decay].mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
