"""Training step for the forward path (BASELINE config 5).

  optim.py   FlatParameters (one fp32 arena for parameters / gradients, weight-decay group first) and FlatAdamW: the
             reference's Optimizer.step (experiments/optimizers.py:230-244) as three launches - squared norm of the decay
             group, AdamW with the clip folded in - and, across ranks, a reduce-scatter of the gradient arena, the update of
             the rank's own slice and an all-gather of the parameters (RCCL over xGMI; optimizer state sharded).
  stack.py   TransformerStackFunction: a `Transformer` stack (plain LayerNorm, fp32, dropout by in-kernel masks) as ONE
             autograd node whose backward is the kernels of csrc/backward.hip.
  stack.py   also ToMelFunction (to_mel's Linear + transpose + mask, model.py:167-168) and `mel_decoder_train_forward`.
  predictor.py  the flow predictor (time embedding, AdaptiveLayerNorm projections, split input projection, adaptive-norm stack,
             output Linear, flow loss) as autograd nodes: `flow_predictor_loss`.
  model.py   `acoustic_train_forward`: the whole teacher-forced forward under the reference's total loss (mel + flow + CTC +
             binarisation, loss.py:140-182); gradients for every parameter tensor of the model.
  aligner.py the aligner front-end (conv blocks, masked instance norms, scores) and the soft averages as autograd nodes
             (csrc/aligner_bwd.hip).
  loss.py    MelLoss (models/acoustic/loss.py:22-35), AttentionCTCLoss (:39-77) and AttentionBinarizationLoss (:80-107), value
             and gradient by kernels.

  graph.py   GraphedTrainStep: the whole step as ONE HIP graph (dropout seed word and AdamW factors read from device memory).

`AcousticModel.forward` itself returns `acoustic_train_outputs` when gradients are enabled, so the reference's loop body
(`outputs = model(**inputs)`; `loss, losses = criterion(inputs=, outputs=, step=)`; `optimizer.step(loss)`,
experiments/trainer.py:543-549) drives these kernels unchanged.  Under bf16 AMP (`amp=True` / torch.autocast) the Linear,
convolution and attention products - forward, dX and weight gradients - run on bf16 MFMAs with LDS-staged attention kernels
(csrc/attention_train.hip) and LDS-DMA weight-gradient GEMMs; DESIGN.md section 4.10 has the step's timeline (13.5 ms per
64 x 512-frame step as a graph).
"""
from .loss import AcousticModelLoss, AttentionBinarizationLoss, AttentionCTCLoss, MelLoss
from .model import acoustic_train_forward, acoustic_train_outputs
from .predictor import flow_predictor_loss
from .optim import FlatAdamW, FlatParameters, group_weight_decayable_params
from .stack import (EmbedTokensFunction, LengthRegulateFunction, MaskedLinearResidualFunction, ToMelFunction, TransformerStackFunction, acoustic_mel_train_forward, mel_decoder_train_forward, transformer_train_forward)

from .graph import GraphedTrainStep

__all__ = ["AcousticModelLoss", "GraphedTrainStep", "EmbedTokensFunction", "MaskedLinearResidualFunction", "acoustic_mel_train_forward", "acoustic_train_forward", "acoustic_train_outputs", "flow_predictor_loss", "AttentionBinarizationLoss", "AttentionCTCLoss", "FlatAdamW", "FlatParameters", "LengthRegulateFunction", "MelLoss", "ToMelFunction", "TransformerStackFunction",
           "group_weight_decayable_params", "mel_decoder_train_forward", "transformer_train_forward"]
