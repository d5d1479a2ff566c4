"""Flow-matching temporal adaptor (tts/models/acoustic/modules/temporal_adaptor.py of the reference).

The transformer stacks (predictor: dim 256 x 3 with AdaptiveLayerNorm; embedding: dim 256 x 1) run on the HIP
kernels through `Transformer`.  The small tensor algebra around them (soft averaging, soft length regulation,
soft-path generation: SURVEY row f3 "next") is stock PyTorch-ROCm with the reference's order of operations.

The flow noise is an explicit, optional input (`noise`, `time_steps`) so that parity runs can feed host-generated
draws; when omitted it is drawn like the reference (randn_like then rand, temporal_adaptor.py:113-115; randn :148).
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import runtime
from ..staging import StagedWeights
from ..modules.constructor import Constructor
from ..modules.transformer import Transformer, TimePositionalEmbedding
from ..utils import get_float_mask_from_lengths, get_mask_3d, masked_mean


class TransformerTemporalModule(nn.Module, Constructor):
    """temporal_adaptor.py:26-59: Transformer(emb_dim=input_dim) -> Linear(with bias) -> * mask."""

    def __init__(self, input_dim: int = 256, output_dim: int = 256, transformer=None, detach_inputs: bool = False):
        super().__init__()
        self.transformer = Transformer.init(transformer, emb_dim=input_dim)
        self.linear_layer = nn.Linear(self.transformer.dim, output_dim, bias=True)
        self.detach_inputs = detach_inputs
        self._cache = StagedWeights()

    def _weight(self, dtype: torch.dtype) -> Tensor:
        w = self.linear_layer.weight
        if dtype == torch.float16:   # split fp16 planes
            return self._cache.get(dtype, (w,), lambda: runtime.split_f16(w.detach().float().contiguous()))
        return self._cache.get(dtype, (w,), lambda: w.detach().to(dtype).contiguous())

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, *, key_len: Optional[Tensor] = None,
                residual: Optional[Tensor] = None) -> Tensor:
        """`key_len` (= mask.sum(1), when the caller already has the lengths) saves the reduction launch; `residual`
        (fp32 [B, L, output_dim]) is added AFTER the mask in the output GEMM's epilogue - the caller's
        `enc_out + embedding(...)` (temporal_adaptor.py:297) without a separate add."""
        m2 = mask[..., 0] if mask is not None else None
        cdt = self.transformer.layers[0].attention.compute_dtype   # bf16 path: the output Linear is an MFMA GEMM too
        out = self.transformer(x, mask=m2, out_dtype=cdt, key_len=key_len if m2 is not None else None).out
        flags = runtime.EP_MASK_ACC if m2 is not None else 0       # (acc + bias) * mask, then + residual
        if cdt == torch.float16:
            return runtime.gemm_split(out, self._weight(cdt), bias=self.linear_layer.bias, mask=m2, flags=flags, resid=residual)
        return runtime.gemm(out, self._weight(cdt), bias=self.linear_layer.bias, mask=m2, flags=flags, resid=residual,
                            out_dtype=torch.float32)

    def infer(self, x: Tensor, mask: Optional[Tensor] = None, steps: int = 4) -> Tensor:
        return self.forward(x, mask)


class FlowTransformerTemporalModule(nn.Module, Constructor):
    """temporal_adaptor.py:72-170.  The predictor input is cat([x_t (3), cond (384)]) -> project_emb (387 -> 256).
    That projection is split: the 384 condition channels go through ONE MFMA GEMM per call (they do not change between
    Euler steps), the 3 flow channels are a K=3 update (`ispk_linear_small_f32` with the GEMM result as residual)."""

    def __init__(self, input_dim: int = 256, output_dim: int = 256, transformer=None,
                 time_embedding_dim: Optional[int] = None, sigma: float = 1e-5, detach_inputs: bool = False):
        super().__init__()
        time_embedding_dim = time_embedding_dim or input_dim
        self.time_embedding = TimePositionalEmbedding(freq_dim=64, emb_dim=time_embedding_dim, with_steps=True)
        self.transformer = Transformer.init(transformer, emb_dim=output_dim + input_dim, adaptive_norm=True,
                                            condition_dim=time_embedding_dim)
        self.linear_layer = nn.Linear(self.transformer.dim, output_dim, bias=True)
        self.output_dim, self.sigma, self.detach_inputs = output_dim, sigma, detach_inputs
        self.fused_head = True          # final norm + linear_layer + flow algebra as `runtime.flow_head` (False: three launches; tests)
        self._cache = StagedWeights()
        self._grids: dict = {}

    def _cond_weight(self, dtype: torch.dtype) -> Tensor:
        w = self.transformer.project_emb.weight
        if dtype == torch.float16:   # split fp16 planes
            return self._cache.get(dtype, (w,), lambda: runtime.split_f16(w.detach()[:, self.output_dim:].float().contiguous()))
        return self._cache.get(dtype, (w,), lambda: w.detach()[:, self.output_dim:].to(dtype).contiguous())

    def _project(self, x_t: Tensor, cond_proj: Tensor) -> Tensor:
        w = self.transformer.project_emb.weight
        return runtime.linear_small(x_t.contiguous(), w[:, :self.output_dim], None, resid=cond_proj)

    def _cond_projection(self, cond: Tensor) -> Tensor:
        cdt = self.transformer.layers[0].attention.compute_dtype   # bf16 path: bf16 operands, fp32 result (residual stream)
        c = cond.float().contiguous()
        if cdt == torch.float16:
            return runtime.gemm_split(runtime.split_f16(c), self._cond_weight(cdt), bias=self.transformer.project_emb.bias)
        if cdt == torch.bfloat16:
            c = runtime.cast_bf16(c)
        return runtime.gemm(c, self._cond_weight(cdt), bias=self.transformer.project_emb.bias, out_dtype=torch.float32)

    def forward(self, x: Tensor, targets: Tensor, mask: Optional[Tensor] = None, *, noise: Optional[Tensor] = None,
                time_steps: Optional[Tensor] = None, key_len: Optional[Tensor] = None):
        cond = x
        have_mask = mask is not None
        if mask is None:
            mask = torch.ones(x.shape[:2], dtype=torch.bool, device=x.device)
        elif mask.ndim == 3:
            mask = mask[..., 0]
        x1 = targets.detach()
        x0 = torch.randn_like(x1) if noise is None else noise.to(x1)
        t = torch.rand((x1.shape[0],), dtype=x1.dtype, device=x1.device) if time_steps is None else time_steps.to(x1)
        time_emb = self.time_embedding(t)
        x_t, flow = runtime.flow_mix(x0, x1, t, self.sigma)                     # :123-126, one launch
        proj = self._project(x_t, self._cond_projection(cond))
        tr = self.transformer
        head = (self.fused_head and tr.dim == 256 and self.output_dim == 3 and tr.norm.weight is not None and tr.norm.bias is not None
                and self.linear_layer.bias is not None)
        out = tr(None, mask=mask, adaptive_condition=time_emb, projected=proj, key_len=key_len if have_mask else None,
                 final_norm=not head).out
        if head:
            # the stack's final LayerNorm, linear_layer and the flow-matching algebra in ONE pass per row (two launches, the
            # second a one-workgroup sum): these are the last kernels of the forward's side branch - they run after the decoder
            pred, self._duration_estimate, _, loss = runtime.flow_head(out, tr.norm.weight, tr.norm.bias, tr.norm.eps,
                                                                       self.linear_layer.weight, self.linear_layer.bias, flow, x0, mask)
            return pred, {"flow_loss": loss}
        raw = runtime.linear_small(out, self.linear_layer.weight, self.linear_layer.bias)
        # pred_flow = raw * m3 ; loss = masked_mean(mse(pred_flow, flow), m3) ; pred = (x0 + pred_flow) * m3 ;
        # duration estimate = clamp(exp(pred[..., 0]) - 1, 0): one launch + the mean over the batch
        pred, self._duration_estimate, _, loss = runtime.flow_finish(raw, flow, x0, mask)
        return pred, {"flow_loss": loss}

    def euler_grid(self, steps: int, step_factor: float, device=None) -> Tensor:
        """temporal_adaptor.py:150-156 (steps=4, factor .75 -> [0, .3657, .6400, .8457, 1]): a function of (steps, step_factor)
        alone, evaluated ONCE on the host with the reference's fp32 expression and cached - no device launches."""
        assert step_factor <= 1.
        key = (int(steps), float(step_factor))
        grid = self._grids.get(key)
        if grid is None:
            if step_factor == 1.:
                grid = torch.linspace(0, 1, steps + 1)
            else:
                ts = -torch.diff(torch.logspace(0, steps, steps + 1, base=step_factor))
                ts = torch.cat([torch.zeros(1), ts])
                grid = torch.cumsum(ts / ts.sum(), dim=0)
            self._grids[key] = grid
        return grid if device is None else grid.to(device)

    def infer(self, x: Tensor, mask: Optional[Tensor] = None, steps: int = 4, step_factor: float = 0.75, *,
              noise: Optional[Tensor] = None, key_len: Optional[Tensor] = None) -> Tensor:
        """temporal_adaptor.py:140-170.  Every launch is a libispk kernel: the time grid lives on the host (its `steps`
        embeddings come from ONE `ispk_time_embedding_f32` launch), an Euler step is `ispk_flow_euler_f32`; `key_len`
        (= mask.sum(1), when the caller has the lengths) saves the reduction."""
        dev = x.device
        if mask is not None and mask.ndim == 3:
            mask = mask.squeeze(-1)
        x_t = (torch.randn(x.shape[0], x.shape[1], self.output_dim, device=dev) if noise is None
               else noise.to(device=dev, dtype=torch.float32))
        dts = self._grids.get(("dt", int(steps), float(step_factor)))
        if dts is None:                                                # fp32 differences, as the reference forms them (cached)
            ts = self.euler_grid(steps, step_factor)
            dts = self._grids[("dt", int(steps), float(step_factor))] = (ts[1:] - ts[:-1]).tolist()
        temb = self.time_embedding(self._grid_on(dev, steps, step_factor)[:steps])     # [steps, emb]: one launch
        cond_proj = self._cond_projection(x)
        if mask is not None and key_len is None:
            key_len = mask.sum(dim=1)
        for i in range(steps):
            out = self.transformer(None, mask=mask, adaptive_condition=temb[i].view(1, 1, -1),
                                   projected=self._project(x_t, cond_proj), key_len=key_len).out
            vel = runtime.linear_small(out, self.linear_layer.weight, self.linear_layer.bias)
            x_t = runtime.flow_euler(x_t, vel, dts[i], mask if i == steps - 1 else None)
        return x_t

    def _grid_on(self, device, steps: int, step_factor: float) -> Tensor:
        key = (str(device), int(steps), float(step_factor))
        g = self._grids.get(key)
        if g is None:
            g = self._grids[key] = self.euler_grid(steps, step_factor).to(device)
        return g


class TemporalAdaptorOutput(NamedTuple):
    enc_out: Tensor
    log_duration: Optional[Tensor]
    duration: Tensor
    dec_lengths: Tensor
    pitch: Optional[Tensor]
    energy: Optional[Tensor]
    pitch_target: Optional[Tensor]
    energy_target: Optional[Tensor]
    losses: Optional[dict] = None
    dec_mask: Optional[Tensor] = None   # arange(frames) < dec_lengths (not in the reference's tuple; saves the caller a launch)


class LengthRegulator(nn.Module):
    """temporal_adaptor.py:411-436, soft branch (the recipes' soft_duration): out = alignment @ x,
    dec_lens = (sum(dur) + .5).long(), both cut to max_len.  One kernel (`ispk_length_regulate_f32`: exact-fp32 MFMA
    products, decoder lengths and the decoder mask from the same launch; the mask of the last call is kept in
    `self.dec_mask` for the caller that needs it next).  `alignment=None` with fp32 durations: the soft path of
    `generate_soft_path` is generated inside the kernel (`infer`).  `split_bf16` (set by `AcousticModel.set_compute_dtype` on the
    bf16 path): `ispk_length_regulate_split_bf16`, each product as three bf16 MFMAs on hi / lo splits of the fp32 operands."""
    split_bf16 = False

    def forward(self, x: Tensor, durations: Tensor, max_len: Optional[int] = None, alignment: Optional[Tensor] = None, *,
                enc_len: Optional[Tensor] = None, frames: Optional[int] = None):
        if alignment is None and (frames is None or durations.dtype == torch.int64):
            raise NotImplementedError("hard (repeat) length regulation is unused by the recipes (soft_duration: true)")
        rows = alignment.shape[1] if alignment is not None else frames
        if max_len is not None and alignment is not None:
            rows = min(rows, max_len)
            alignment = alignment[:, :rows]
        out, dec_lens, self.dec_mask = runtime.length_regulate(x.float(), durations, alignment, rows,
                                                               max_len=-1 if max_len is None else max_len, enc_len=enc_len,
                                                               split_bf16=self.split_bf16)
        return out, dec_lens


class TemporalAverager(nn.Module):
    """temporal_adaptor.py:439-449, soft branch: x[B,1,M] @ A[B,M,L] / (colsum(A) + 1e-5)."""

    def forward(self, x: Tensor, durations: Tensor, alignment: Optional[Tensor] = None) -> Tensor:
        if alignment is None:
            raise NotImplementedError("hard averaging is unused by the recipes (soft_duration: true)")
        return x @ alignment / (alignment.sum(dim=1, keepdim=True) + 1e-5)


def generate_soft_path(duration: Tensor, mask: Tensor) -> Tensor:
    """temporal_adaptor.py:468-478."""
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1).view(b * t_x)
    path = get_float_mask_from_lengths(cum, t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, [0, 0, 1, 0, 0, 0])[:, :-1]
    return path * mask


class FlowTemporalAdaptor(nn.Module, Constructor):
    # launch order with a predictor stream: the predictor's launches issued BEFORE the embedding stack's (True) or after (False)
    predictor_first = False

    def __init__(self, encoder_dim: int = 384, predictor=None, embedding=None, pitch: bool = True, energy: bool = True,
                 soft_duration: bool = False):
        super().__init__()
        if not (pitch and energy and soft_duration):
            raise NotImplementedError("built for the recipes' adaptor: pitch, energy and soft_duration all on")
        self.length_regulator = LengthRegulator()
        self.averager = TemporalAverager()
        self.encoder_dim = encoder_dim
        self.feature_dim = 3
        self.pitch, self.energy, self.soft_duration = pitch, energy, soft_duration
        self.pitch_idx, self.energy_idx = 1, 2
        self.predictor = FlowTransformerTemporalModule.init(predictor, input_dim=encoder_dim, output_dim=self.feature_dim)
        self.embedding = TransformerTemporalModule.init(embedding, input_dim=self.feature_dim - 1, output_dim=encoder_dim)

    def _process_target(self, dense: Tensor, duration_target: Tensor, alignment: Tensor, enc_mask: Tensor) -> Tensor:
        if dense.ndim == 2:
            dense = dense[:, None]
        return self.averager(dense, duration_target, alignment).transpose(1, 2) * enc_mask

    def forward(self, enc_out: Tensor, enc_mask: Tensor, max_dec_len: int, duration_target: Optional[Tensor] = None,
                alignment: Optional[Tensor] = None, pitch_target_dense: Optional[Tensor] = None,
                energy_target_dense: Optional[Tensor] = None, *, noise: Optional[Tensor] = None,
                time_steps: Optional[Tensor] = None, enc_len: Optional[Tensor] = None,
                predictor_stream=None, duration_sum: Optional[Tensor] = None) -> TemporalAdaptorOutput:
        """temporal_adaptor.py:238-312 (teacher-forced: the decoder input uses the TARGET pitch/energy, :284,:292).

        What the DECODER waits for is short: pitch / energy targets (soft averages over attn_soft), the embedding stack,
        the length regulator.  Two things it does not wait for:
          * the flow predictor - its outputs (predicted duration / pitch / energy, flow loss) feed nothing downstream in
            the teacher-forced forward; with `predictor_stream` its ~40 small launches go to that stream, issued AFTER the
            decoder-critical ones (in a captured graph launch order follows creation order);
          * the hard alignment - `duration_target` enters the decoder path only as dec_lens = sum of durations, and
            `duration_sum` (int64 [B]) says what that sum is (AcousticModel.forward passes mel_len: the MAS durations add
            up to it by construction, alignment.py:278-282).  `duration_target` itself may then still be in flight on
            `predictor_stream` (MAS runs there); it is only read on that stream (log1p duration target of the predictor).
        The CALLER joins `predictor_stream` before using the predictor's outputs (`AcousticModel.forward` does)."""
        assert alignment is not None and duration_target is not None
        assert pitch_target_dense is not None and energy_target_dense is not None
        m3 = enc_mask[..., None]
        if enc_len is None:
            enc_len = enc_mask.sum(dim=1)
        side = predictor_stream if (predictor_stream is not None and enc_out.is_cuda) else None
        cond = enc_out
        if side is None or duration_sum is None:
            # one stream, or no shortcut for the lengths: the three flow targets (log1p duration, soft-averaged pitch and
            # energy, :257-269) from ONE kernel, read by both branches
            targets = runtime.soft_average(alignment, pitch_target_dense, energy_target_dense, duration_target, enc_len)
            feats = targets
            len_src = duration_target
        else:
            feats = runtime.soft_average(alignment, pitch_target_dense, energy_target_dense, None, enc_len)   # no durations
            targets = None
            len_src = duration_sum.view(-1, 1)
        if side is not None:   # the predictor's inputs are complete here; its launches are issued further down
            ready = torch.cuda.Event()
            ready.record()
        pitch_target, energy_target = feats[..., 1:2], feats[..., 2:3]
        features = feats[..., 1:3]                   # = cat([pitch_target, energy_target], -1): a view, no copy

        def decoder_input(enc_out_):
            enc_out_ = self.embedding(features, mask=m3, key_len=enc_len, residual=enc_out_)   # enc_out + embedding(...)
            return self.length_regulator(enc_out_, len_src, max_len=max_dec_len, alignment=alignment)

        if not (side is not None and self.predictor_first):
            enc_out, dec_lens = decoder_input(enc_out)

        def predict():
            tg = targets
            if tg is None:     # on the predictor's stream, behind MAS: the targets again, with the log1p durations this time
                tg = runtime.soft_average(alignment, pitch_target_dense, energy_target_dense, duration_target, enc_len)
            pred_, losses_ = self.predictor(cond, tg, m3, noise=noise, time_steps=time_steps, key_len=enc_len)
            return pred_, losses_, self.predictor._duration_estimate   # = clamp(exp(pred[..., 0]) - 1, min=0)

        if side is not None:
            side.wait_event(ready)
            with torch.cuda.stream(side):
                pred, losses, duration_pred = predict()
            for t in (cond, feats, alignment, pitch_target_dense, energy_target_dense):
                t.record_stream(side)
            if self.predictor_first:
                enc_out, dec_lens = decoder_input(enc_out)
        else:
            pred, losses, duration_pred = predict()
        log_duration_pred = pred[..., 0]
        return TemporalAdaptorOutput(enc_out=enc_out, log_duration=log_duration_pred, duration=duration_pred,
                                     dec_lengths=dec_lens, pitch=pred[..., 1], energy=pred[..., 2],
                                     pitch_target=pitch_target.squeeze(-1), energy_target=energy_target.squeeze(-1),
                                     losses=losses, dec_mask=self.length_regulator.dec_mask)

    def infer(self, enc_out: Tensor, enc_mask: Optional[Tensor] = None, duration_target: Optional[Tensor] = None,
              duration_factor: float = 1.0, pitch_target: Optional[Tensor] = None, pitch_factor: float = 1.0,
              pitch_delta: float = 0., energy_target: Optional[Tensor] = None, energy_factor: float = 1.0,
              energy_delta: float = 0., steps: int = 4, *, noise: Optional[Tensor] = None,
              max_dec_len: Optional[int] = None, enc_len: Optional[Tensor] = None) -> TemporalAdaptorOutput:
        """temporal_adaptor.py:331-408.  Durations stay fractional (soft_duration, :355-356); the embedding transformer
        gets NO mask even when batched (:384).  `max_dec_len` (optional) fixes the decoder length without reading
        `dec_lens.max()` back to the host."""
        m3 = enc_mask[..., None] if enc_mask is not None else None
        pred = self.predictor.infer(enc_out, mask=m3, steps=steps, noise=noise, key_len=enc_len)
        # :351-381 in ONE kernel: duration = clamp(duration_factor * (exp(pred[..., 0]) - 1), 0) with the given targets
        # taking the place of predictions where they are >= 0 (the reference tests `(duration_target < 0).any()` on the host
        # and fills only the negative entries; per element on the device that is the same values without the round trip,
        # which keeps the call capturable in a HIP graph), and the embedding stack's [pitch, energy] input.
        duration_pred, feats = runtime.infer_features(pred, duration_target, pitch_target, energy_target, duration_factor,
                                                      pitch_factor, pitch_delta, energy_factor, energy_delta)
        pitch, energy = feats[..., 0:1], feats[..., 1:2]
        enc_out = self.embedding(feats, residual=enc_out)                                   # no mask, even batched (:384)
        enc_lens = enc_len
        if enc_lens is None and enc_mask is not None:
            enc_lens = enc_mask.sum(dim=1)
        if max_dec_len is None:   # the output shape is data: like the reference, read the longest decoder length back
            max_dec_len = int((duration_pred.sum(dim=1) + 0.5).long().max().item())
        # :388-397: soft path (generate_soft_path) and length regulation in one kernel
        enc_out, dec_lens = self.length_regulator(enc_out, duration_pred, alignment=None, enc_len=enc_lens,
                                                  frames=max_dec_len)
        return TemporalAdaptorOutput(enc_out=enc_out, log_duration=None, duration=duration_pred, dec_lengths=dec_lens,
                                     pitch=pitch.squeeze(-1), energy=energy.squeeze(-1), pitch_target=pitch_target,
                                     energy_target=energy_target, dec_mask=self.length_regulator.dec_mask)
