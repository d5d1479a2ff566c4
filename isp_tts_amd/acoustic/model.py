"""`AcousticModel`: forward / infer orchestration (tts/models/acoustic/model.py:60-238 of the reference).

Same constructor config, parameter names (`state_dict` keys) and call signatures.  What runs where:
  * encoder / decoder / adaptor transformer stacks, `to_mel`, MAS  -> hand-written gfx950 kernels (libispk.so);
  * token embedding lookup, the aligner's conv front-end, the adaptor's small tensor algebra -> PyTorch-ROCm ops
    ("next");
all on the current stream, with no host synchronisation inside `forward` when `max_*` lengths are given by shapes.
"""
from __future__ import annotations

import warnings
from typing import NamedTuple, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import runtime
from ..staging import StagedWeights
from ..modules.constructor import Constructor
from ..modules.transformer import Transformer
from ..utils import get_mask_from_lengths
from .alignment import Aligner, AlignerOutput
from .temporal_adaptor import FlowTemporalAdaptor, TemporalAdaptorOutput


class AcousticModelOutput(NamedTuple):
    mel: Tensor
    adaptor_output: TemporalAdaptorOutput
    aligner_output: AlignerOutput
    loss: Optional[Tensor] = None
    losses: Optional[dict] = None


class AcousticModel(nn.Module, Constructor):
    def __init__(self, encoding_map: dict, mel_dim: int, text_dim: int = 384, encoder=None, decoder=None,
                 temporal_adaptor=None, aligner=None, num_speakers: Optional[int] = 0, pitch_mean=None, pitch_std=None):
        super().__init__()
        self.encoding_map = dict(encoding_map)
        self.mel_dim, self.text_dim = mel_dim, text_dim
        self.text_embedding = nn.Embedding(len(encoding_map), text_dim, padding_idx=0)
        self.encoder = Transformer.init(encoder, emb_dim=text_dim)
        enc_dim = self.encoder.dim
        self.aligner = Aligner.init(aligner, mel_dim=mel_dim, text_dim=enc_dim)
        # model.py:93-97: multi-speaker models (the authors' published 1,307-speaker checkpoint, notebooks/inference.ipynb)
        self.speaker_embedding = None
        if (num_speakers or 0) > 0:
            self.speaker_embedding = nn.Embedding(num_speakers, enc_dim)
            nn.init.xavier_uniform_(self.speaker_embedding.weight)
        self.train_amp: Optional[bool] = None     # forward under grad: bf16 AMP (None = follow torch.autocast's state)
        self.temporal_adaptor = FlowTemporalAdaptor.init(temporal_adaptor, encoder_dim=enc_dim)
        self.decoder = Transformer.init(decoder, emb_dim=enc_dim)
        self.to_mel = nn.Linear(self.decoder.dim, mel_dim)
        self.overlap_streams = True          # run the aligner's mel-side branch beside the text encoder (forward())
        self._side_streams: dict = {}
        self.register_buffer("pitch_mean", torch.tensor(float(pitch_mean or 0.)))
        self.register_buffer("pitch_std", torch.tensor(float(pitch_std or 1.)))
        self.compute_dtype = torch.float32
        self._cache = StagedWeights()

    def set_compute_dtype(self, dtype: torch.dtype, *, alignment_dtype: Optional[torch.dtype] = None):
        """fp32 (parity path: exact-fp32 MFMA) or bf16 (throughput path: bf16 operands, fp32 accumulation, fp32 residual
        stream / LayerNorm / softmax statistics) for the decoder and the temporal adaptor's stacks - where the FLOPs are.

        `alignment_dtype` is the precision of everything UPSTREAM OF MAS: text encoder -> aligner key convolutions, mel ->
        aligner query convolutions, scores / log-softmax / prior.  MAS turns those logits into DISCRETE outputs (the hard
        alignment, the durations), and a near-tie between two paths flips on any change of the logits' low bits, so only
        the same fp32 arithmetic reproduces the fp32 path's alignments.  Measured at B = 64 x 100 x 512 (MI355X, round 2):
          * None (default: the chain follows `dtype`): all-bf16, 38/64 alignments identical to the fp32 path's;
          * aligner front-end alone in fp32 (bf16 text encoder): 44/64 - the logits still carry the encoder's bf16
            rounding - for +0.45 ms per step; not offered as a mode;
          * torch.float32: text encoder AND aligner front-end run exactly the fp32 path's kernels - logits, MAS paths
            and durations are bit-identical to the fp32 path's on all 64 utterances (and equal the oracle's on all 64),
            at 4.9 instead of 2.5 ms per step (the 6,400-row fp32 encoder costs 16 x the bf16 MFMA time).
        mel does not depend on the hard alignment (the decoder input is built from attn_soft and the dense targets,
        temporal_adaptor.py:284-300); the durations and the flow-matching duration target do."""
        chain = alignment_dtype if alignment_dtype is not None else dtype
        assert chain in (torch.float32, torch.bfloat16, torch.float16)
        self.encoder.set_compute_dtype(chain)
        self.decoder.set_compute_dtype(dtype)
        self.temporal_adaptor.predictor.transformer.set_compute_dtype(dtype)
        self.temporal_adaptor.embedding.transformer.set_compute_dtype(dtype)
        self.aligner.attention.compute_dtype = chain
        # the regulator's products: exact fp32 MFMAs (fp32 path), three bf16 MFMAs (bf16 path: 2^-16), three fp16 MFMAs over
        # 22-bit operands (split-fp16 path: fp32-grade)
        self.temporal_adaptor.length_regulator.split_bf16 = (True if dtype == torch.bfloat16 else
                                                             "f16" if dtype == torch.float16 else False)
        self.compute_dtype = dtype
        return self

    def _to_mel(self, dec_out: Tensor, dec_mask: Optional[Tensor]) -> Tensor:
        w = self.to_mel.weight
        if dec_out.dtype == torch.float16:   # split fp16 planes [2, B, T, D] from the decoder's final norm
            ws = self._cache.get(torch.float16, (w,), lambda: runtime.split_f16(w.detach().float().contiguous()))
            return runtime.to_mel_split(dec_out, ws, self.to_mel.bias, dec_mask)
        if dec_out.dtype == torch.bfloat16:
            w = self._cache.get(torch.bfloat16, (w,), lambda: w.detach().to(torch.bfloat16).contiguous())
        return runtime.to_mel(dec_out, w, self.to_mel.bias, dec_mask)

    def forward(self, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Optional[Tensor] = None,
                energy: Optional[Tensor] = None, speaker: Optional[Tensor] = None, sigma: float = 0., steps: int = 1, *,
                flow_noise: Optional[Tensor] = None, flow_time: Optional[Tensor] = None) -> AcousticModelOutput:
        """model.py:116-174.  text int64 [B,L], mel fp32 [B,80,M], pitch/energy fp32 [B,M], lengths int64 [B]
        (collator.py:36-55).  Padded shapes define the masks' widths (max length = L / M, as collated batches have).

        With gradients enabled and a trainable parameter - the reference's training loop, `outputs = model(**inputs)`
        (experiments/trainer.py:544) - the outputs carry autograd nodes whose backward is HIP kernels
        (`train.acoustic_train_outputs`): `criterion(inputs=, outputs=)` then `optimizer.step(loss)` train the model as they
        train the reference.  Under `torch.no_grad()` / with every parameter frozen: the inference kernels (no tape)."""
        if self.speaker_embedding is not None:
            # model.py:145-146: the reference's forward reads `self.speaker_encoder`, which no AcousticModel has - a
            # multi-speaker model cannot be run through `forward` there either (only through `infer`); same error here
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'speaker_encoder'")
        if torch.is_grad_enabled() and text.is_cuda and any(p.requires_grad for p in self.parameters()):
            from ..train.model import acoustic_train_outputs
            amp = self.train_amp if self.train_amp is not None else torch.is_autocast_enabled("cuda")
            train_aligner = any(p.requires_grad for p in self.aligner.parameters())
            with torch.autocast("cuda", enabled=False):      # (the kernels choose their own operand types: `amp`)
                return acoustic_train_outputs(self, text, text_len, mel, mel_len, pitch, energy, flow_noise, flow_time, amp=amp,
                                              train_aligner=train_aligner)
        with torch.no_grad():
            return self._forward_values(text, text_len, mel, mel_len, pitch, energy, flow_noise=flow_noise, flow_time=flow_time)

    def _forward_values(self, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Optional[Tensor] = None,
                        energy: Optional[Tensor] = None, *, flow_noise: Optional[Tensor] = None,
                        flow_time: Optional[Tensor] = None) -> AcousticModelOutput:
        """The forward without a tape (inference kernels, `set_compute_dtype`'s precision)."""
        # The aligner's mel-side projections (33,000 frames: large, HBM-bound launches) do not depend on the text encoder
        # (6,400 tokens: small, latency-bound launches), so they run beside it on a second stream; under HIP-graph
        # capture the fork / join become graph edges.
        q_proj = None
        if mel.is_cuda and self.overlap_streams:
            main = torch.cuda.current_stream()
            # (a high-priority side stream changes nothing for one graph - HIP graph replay does not seem to honour stream
            # priorities - and it stops two graph instances in flight from overlapping at all: measured 4.99 vs 2.37 ms)
            side = self._side_streams.setdefault(mel.device, torch.cuda.Stream(device=mel.device))
            side.wait_stream(main)
            with torch.cuda.stream(side):
                q_proj = self.aligner.attention.project_queries(mel, mel_len)
        token_emb, enc_mask = runtime.embed_tokens(text, self.text_embedding.weight, text_len)   # model.py:131-134
        enc_out = self.encoder(token_emb, mask=enc_mask, key_len=text_len).out
        if q_proj is not None:
            main.wait_stream(side)
            q_proj.record_stream(main)
        # MAS and the flow predictor (which wants the MAS durations as a target) leave the main stream here: the decoder
        # path needs neither - it is built from attn_soft and the dense targets, and its lengths are sum(durations) =
        # mel_len.  They run on the side stream beside the embedding stack, the length regulator and the decoder.
        branch = side if q_proj is not None else None
        aligner_output = self.aligner(mel=mel, enc_text=enc_out.transpose(1, 2).detach(), mel_len=mel_len,
                                      text_len=text_len, q_proj=q_proj, mas_stream=branch)
        adaptor_output = self.temporal_adaptor(
            enc_out=enc_out, enc_mask=enc_mask, max_dec_len=mel.size(2),
            duration_target=aligner_output.attn_hard_duration, alignment=aligner_output.attn_soft,
            pitch_target_dense=pitch, energy_target_dense=energy, noise=flow_noise, time_steps=flow_time,
            enc_len=text_len, predictor_stream=branch, duration_sum=mel_len if branch is not None else None)
        dec_len = adaptor_output.dec_lengths
        dec_mask = adaptor_output.dec_mask              # arange(frames) < dec_len, from the length-regulation kernel
        dec_out = self.decoder(adaptor_output.enc_out, mask=dec_mask, key_len=dec_len, out_dtype=self.compute_dtype).out
        mel_out = self._to_mel(dec_out, dec_mask)
        if q_proj is not None:
            # join the MAS / flow-predictor branch (it ran beside the embedding stack and the decoder) before handing out its tensors
            main.wait_stream(side)
            for t in (adaptor_output.log_duration, adaptor_output.duration, adaptor_output.pitch, adaptor_output.energy,
                      aligner_output.attn_hard, aligner_output.attn_hard_duration, *(adaptor_output.losses or {}).values()):
                if isinstance(t, Tensor):
                    t.record_stream(main)
        return AcousticModelOutput(mel=mel_out, adaptor_output=adaptor_output, aligner_output=aligner_output)

    # ------------------------------------------------------------------------------------------------------------------
    # The same forward in three pieces for `graph.SegmentedForward` (one HIP graph each).  Inside ONE captured graph the
    # side branch (MAS + flow predictor) is replayed late - its last kernels run alone after to_mel and lengthen the step
    # by ~0.12 ms; launched as its own graph on its own stream right after the front piece it runs under the decoder.
    @torch.no_grad()
    def forward_front(self, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Tensor, energy: Tensor):
        """Token embedding, text encoder, aligner front-end (model.py:131-141 up to the soft attention) and the dense
        targets' soft averages -> state for the other two pieces."""
        main = torch.cuda.current_stream()
        side = self._side_streams.setdefault(mel.device, torch.cuda.Stream(device=mel.device))
        side.wait_stream(main)
        with torch.cuda.stream(side):
            q_proj = self.aligner.attention.project_queries(mel, mel_len)
        token_emb, enc_mask = runtime.embed_tokens(text, self.text_embedding.weight, text_len)
        enc_out = self.encoder(token_emb, mask=enc_mask, key_len=text_len).out
        main.wait_stream(side)
        q_proj.record_stream(main)
        attn_soft, attn_logits = self.aligner.attention(mel, enc_out.transpose(1, 2).detach(), mel_len, text_len,
                                                        q_proj=q_proj)
        feats = runtime.soft_average(attn_soft, pitch, energy, None, text_len)      # pitch / energy targets (no durations)
        return {"enc_out": enc_out, "enc_mask": enc_mask, "attn_soft": attn_soft, "attn_logits": attn_logits, "feats": feats}

    @torch.no_grad()
    def forward_side(self, st: dict, text_len: Tensor, mel_len: Tensor, pitch: Tensor, energy: Tensor,
                     flow_noise: Optional[Tensor], flow_time: Optional[Tensor]):
        """MAS and the flow predictor (model.py:142, temporal_adaptor.py:257-282): nothing the decoder needs."""
        attn_hard, dur = self.aligner.binarize_attention_parallel(st["attn_logits"], text_len, mel_len, return_duration=True)
        ad = self.temporal_adaptor
        targets = runtime.soft_average(st["attn_soft"], pitch, energy, dur, text_len)
        pred, losses = ad.predictor(st["enc_out"], targets, st["enc_mask"][..., None], noise=flow_noise, time_steps=flow_time,
                                    key_len=text_len)
        return {"attn_hard": attn_hard, "dur": dur, "pred": pred, "losses": losses,
                "duration_pred": ad.predictor._duration_estimate}

    @torch.no_grad()
    def forward_back(self, st: dict, text_len: Tensor, mel_len: Tensor, max_dec_len: int):
        """Embedding stack, length regulator, decoder, to_mel (temporal_adaptor.py:284-300, model.py:160-168)."""
        ad = self.temporal_adaptor
        feats = st["feats"]
        x = ad.embedding(feats[..., 1:3], mask=st["enc_mask"][..., None], key_len=text_len, residual=st["enc_out"])
        dec_in, dec_len = ad.length_regulator(x, mel_len.view(-1, 1), max_len=max_dec_len, alignment=st["attn_soft"])
        dec_mask = ad.length_regulator.dec_mask
        dec_out = self.decoder(dec_in, mask=dec_mask, key_len=dec_len, out_dtype=self.compute_dtype).out
        return {"mel": self._to_mel(dec_out, dec_mask), "dec_in": dec_in, "dec_len": dec_len, "dec_mask": dec_mask}

    @staticmethod
    def assemble_output(front: dict, side: dict, back: dict) -> AcousticModelOutput:
        feats, pred = front["feats"], side["pred"]
        adaptor = TemporalAdaptorOutput(enc_out=back["dec_in"], log_duration=pred[..., 0], duration=side["duration_pred"],
                                        dec_lengths=back["dec_len"], pitch=pred[..., 1], energy=pred[..., 2],
                                        pitch_target=feats[..., 1], energy_target=feats[..., 2], losses=side["losses"],
                                        dec_mask=back["dec_mask"])
        aligner = AlignerOutput(attn_soft=front["attn_soft"], attn_logits=front["attn_logits"], attn_hard=side["attn_hard"],
                                attn_hard_duration=side["dur"])
        return AcousticModelOutput(mel=back["mel"], adaptor_output=adaptor, aligner_output=aligner)

    @torch.no_grad()
    def infer(self, input_sequence: Tensor, text_lengths: Optional[Tensor] = None,
              duration_target: Optional[Tensor] = None, duration_factor: float = 1.0,
              pitch_target: Optional[Tensor] = None, pitch_factor: float = 1.0, pitch_delta: float = 0.,
              pitch_normalize: bool = False, energy_target: Optional[Tensor] = None, steps: int = 4,
              speaker: Optional[Tensor] = None, *, flow_noise: Optional[Tensor] = None,
              max_dec_len: Optional[int] = None):
        """model.py:177-238: masks only for batch > 1 (:191-201, :228)."""
        batch_infer = input_sequence.shape[0] > 1
        token_emb, enc_mask = runtime.embed_tokens(input_sequence, self.text_embedding.weight,
                                                   text_lengths if batch_infer else None, want_mask=batch_infer)
        enc_out = self.encoder(token_emb, mask=enc_mask, key_len=text_lengths if batch_infer else None).out
        if self.speaker_embedding is not None and speaker is not None:      # model.py:205-207
            enc_out = runtime.add_speaker_(enc_out, self.speaker_embedding.weight, speaker)
        if pitch_normalize:
            if pitch_target is not None:
                pitch_target = (pitch_target - self.pitch_mean) / self.pitch_std
            pitch_delta = pitch_delta / self.pitch_std
        adaptor_output = self.temporal_adaptor.infer(
            enc_out=enc_out, enc_mask=enc_mask, duration_target=duration_target, pitch_target=pitch_target,
            energy_target=energy_target, duration_factor=duration_factor, pitch_factor=pitch_factor,
            pitch_delta=pitch_delta, steps=steps, noise=flow_noise, max_dec_len=max_dec_len,
            enc_len=text_lengths if batch_infer else None)
        dec_mask = adaptor_output.dec_mask if batch_infer else None
        dec_out = self.decoder(adaptor_output.enc_out, mask=dec_mask, key_len=adaptor_output.dec_lengths if batch_infer else None,
                               out_dtype=self.compute_dtype).out
        return self._to_mel(dec_out, dec_mask), adaptor_output

    # ---- checkpoint drop-in (tts/models/base.py:39-108 of the reference; trainer.py:361-372 writes the file) ----
    @classmethod
    def from_pretrained(cls, checkpoint_path: str, strict: bool = True) -> "AcousticModel":
        """Builds the model from a checkpoint written by the reference's Trainer: a `torch.save`d dict whose
        `["model"]["config"]` is the plain-container model config and `["model"]["state_dict"]` the weights
        (base.py:39-56).  Keys the checkpoint lacks keep their initial values, as in the reference (:50-52)."""
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        model = cls.init(checkpoint["model"]["config"])
        state_dict = dict(checkpoint["model"]["state_dict"])
        for key, weight in model.state_dict().items():
            if key not in state_dict:
                state_dict[key] = weight
        model.load_state_dict(state_dict, strict=strict)
        return model

    def load(self, state_dict: dict, ignore_layers: Optional[list] = None, ignore_mismatched_keys: bool = False):
        """base.py:58-64 / `load_state_dict` :73-108: drops checkpoint keys the model lacks, keys matching
        `ignore_layers` (substring) and - with `ignore_mismatched_keys` - keys whose shape differs; the rest is loaded
        over the current weights."""
        own = self.state_dict()
        extra = [k for k in state_dict if k not in own]
        if extra:
            warnings.warn(f"checkpoint keys absent from the model are ignored: {extra}")
        kept = {k: v for k, v in state_dict.items() if k in own}
        ignored = []
        if ignore_mismatched_keys:
            ignored += [k for k, v in kept.items() if v.shape != own[k].shape]
        if ignore_layers:
            ignored += [k for k in kept if any(layer in k for layer in ignore_layers)]
        kept = {k: v for k, v in kept.items() if k not in ignored}
        own.update(kept)
        self.load_state_dict(own)
        return self

    def freeze(self, exception_list: Optional[list] = None):
        """base.py:66-73: requires_grad only for parameters whose name starts with an entry of `exception_list`."""
        exception_list = exception_list or []
        for name, param in self.named_parameters():
            param.requires_grad = any(name.startswith(layer) for layer in exception_list)
        return self

    def prepare_inputs(self, inputs: dict) -> dict:
        """model.py:244-259 (collator field names -> forward kwargs)."""
        return {"text": inputs["text_vector"], "text_len": inputs["text_vector_len"], "mel": inputs["mel"],
                "mel_len": inputs["mel_len"], "pitch": inputs["pitch"], "energy": inputs["energy"],
                "speaker": inputs.get("speaker")}
