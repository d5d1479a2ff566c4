"""Deterministic synthetic weights and inputs.

There is no network on the build or GPU boxes, so checkpoints are unavailable;
every test, golden fixture and benchmark uses weights produced here.  Each tensor is drawn from
its own counter-based stream keyed by the parameter NAME, so the values are identical in this
container, on the GPU box and inside the golden-fixture generator without committing any weights.

The parameter names/shapes are the reference's `state_dict` layout (verified against the
reference's own key list in tests/golden/state_dict_keys.json).
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch

from .config import AcousticDims

SEED = 23  # recipe seed (recipes/acoustic/core.yaml:5)


def _rng(name: str, seed: int = SEED) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()), seed]))


def _normal(name: str, shape, scale: float = 1.0, shift: float = 0.0, seed: int = SEED) -> torch.Tensor:
    x = _rng(name, seed).standard_normal(size=tuple(shape)) * scale + shift
    return torch.from_numpy(x.astype(np.float32))


def alibi_default_slopes(heads: int) -> list[float]:
    """ALiBi geometric slopes (embeddings.py:38-49 of the reference): for a power-of-two head count
    2^(-8/n * (i+1)); otherwise the closest lower power of two plus every other slope of the next one."""
    def pow2(n):
        start = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if math.log2(heads).is_integer():
        return pow2(heads)
    n = 2 ** math.floor(math.log2(heads))
    return pow2(n) + pow2(2 * n)[0::2][: heads - n]


def _transformer_spec(prefix: str, dim: int, depth: int, heads: int, ffn: int, emb_dim: int,
                      adaptive: bool, cond_dim: int) -> list[tuple]:
    spec = []
    for i in range(depth):
        p = f"{prefix}.layers.{i}"
        for norm in ("attention_norm", "feed_forward_norm"):
            if adaptive:
                spec += [(f"{p}.{norm}.weight.weight", (dim, cond_dim), "ada_w"),
                         (f"{p}.{norm}.weight.bias", (dim,), "gamma"),
                         (f"{p}.{norm}.bias.weight", (dim, cond_dim), "ada_w"),
                         (f"{p}.{norm}.bias.bias", (dim,), "beta")]
            else:
                spec += [(f"{p}.{norm}.weight", (dim,), "gamma"), (f"{p}.{norm}.bias", (dim,), "beta")]
            if norm == "attention_norm":
                spec += [(f"{p}.attention.to_q.weight", (heads * 64, dim), "linear"),
                         (f"{p}.attention.to_kv.weight", (128, dim), "linear"),
                         (f"{p}.attention.rel_pos.learned_logslopes", (heads, 1, 1), "logslopes"),
                         (f"{p}.attention.to_out.weight", (dim, heads * 64), "linear")]
        spec += [(f"{p}.feed_forward.net.0.weight", (ffn, dim), "linear"),
                 (f"{p}.feed_forward.net.3.weight", (dim, ffn), "linear")]
    if emb_dim != dim:
        spec += [(f"{prefix}.project_emb.weight", (dim, emb_dim), "linear"),
                 (f"{prefix}.project_emb.bias", (dim,), "beta")]
    spec += [(f"{prefix}.norm.weight", (dim,), "gamma"), (f"{prefix}.norm.bias", (dim,), "beta")]
    return spec


def model_spec(dims: AcousticDims = AcousticDims()) -> list[tuple]:
    """(name, shape, kind) for every entry of the reference state_dict, in its order."""
    d, a = dims.text_dim, dims.ada_dim
    spec = [("pitch_mean", (), "zero"), ("pitch_std", (), "one"),
            ("text_embedding.weight", (dims.vocab, d), "embedding")]
    spec += _transformer_spec("encoder", d, dims.enc_depth, dims.heads, dims.ffn, d, False, 0)
    kk, (q0, q1) = dims.key_kernel, dims.query_kernels
    al = "aligner.attention"
    spec += [(f"{al}.key_proj.0.conv.weight", (2 * d, d, kk), "conv"),
             (f"{al}.key_proj.0.norm.weight", (2 * d,), "gamma"), (f"{al}.key_proj.0.norm.bias", (2 * d,), "beta"),
             (f"{al}.key_proj.1.conv.weight", (dims.attn_dim, 2 * d, 1), "conv"),
             (f"{al}.query_proj.0.conv.weight", (2 * dims.mel_dim, dims.mel_dim, q0), "conv"),
             (f"{al}.query_proj.0.norm.weight", (2 * dims.mel_dim,), "gamma"),
             (f"{al}.query_proj.0.norm.bias", (2 * dims.mel_dim,), "beta"),
             (f"{al}.query_proj.1.conv.weight", (dims.mel_dim, 2 * dims.mel_dim, q1), "conv"),
             (f"{al}.query_proj.1.norm.weight", (dims.mel_dim,), "gamma"),
             (f"{al}.query_proj.1.norm.bias", (dims.mel_dim,), "beta"),
             (f"{al}.query_proj.2.conv.weight", (dims.attn_dim, dims.mel_dim, 1), "conv")]
    tp = "temporal_adaptor.predictor"
    spec += [(f"{tp}.time_embedding.freq_emb.freq_scale", (1,), "freq_scale"),
             (f"{tp}.time_embedding.mlp.0.weight", (dims.time_dim, 65), "linear"),
             (f"{tp}.time_embedding.mlp.0.bias", (dims.time_dim,), "beta"),
             (f"{tp}.time_embedding.mlp.2.weight", (dims.time_dim, dims.time_dim), "linear"),
             (f"{tp}.time_embedding.mlp.2.bias", (dims.time_dim,), "beta")]
    spec += _transformer_spec(f"{tp}.transformer", a, dims.ada_depth, dims.ada_heads, dims.ada_ffn, d + 3, True,
                              dims.time_dim)
    spec += [(f"{tp}.linear_layer.weight", (3, a), "linear"), (f"{tp}.linear_layer.bias", (3,), "beta")]
    te = "temporal_adaptor.embedding"
    spec += _transformer_spec(f"{te}.transformer", a, dims.emb_depth, dims.ada_heads, dims.ada_ffn, 2, False, 0)
    spec += [(f"{te}.linear_layer.weight", (d, a), "linear"), (f"{te}.linear_layer.bias", (d,), "beta")]
    spec += _transformer_spec("decoder", d, dims.dec_depth, dims.heads, dims.ffn, d, False, 0)
    spec += [("to_mel.weight", (dims.mel_dim, d), "linear"), ("to_mel.bias", (dims.mel_dim,), "beta")]
    return spec


def make_state_dict(dims: AcousticDims = AcousticDims(), seed: int = SEED) -> dict[str, torch.Tensor]:
    """fp32 CPU tensors for every state_dict entry.  Scales are chosen so activations stay O(1) through
    the 12 pre-norm layers (so that an absolute 1e-4 mel tolerance is a meaningful bar)."""
    sd = {}
    for name, shape, kind in model_spec(dims):
        if kind == "zero":
            t = torch.tensor(0.0)
        elif kind == "one":
            t = torch.tensor(1.0)
        elif kind == "freq_scale":
            t = torch.full((1,), 1000.0)
        elif kind == "embedding":
            t = _normal(name, shape, seed=seed)
            t[0].zero_()  # padding_idx = 0
        elif kind == "linear":
            t = _normal(name, shape, 1.0 / math.sqrt(shape[1]), seed=seed)
        elif kind == "conv":
            t = _normal(name, shape, 1.0 / math.sqrt(shape[1] * shape[2]), seed=seed)
        elif kind == "gamma":
            t = _normal(name, shape, 0.1, 1.0, seed=seed)
        elif kind == "beta":
            t = _normal(name, shape, 0.1, seed=seed)
        elif kind == "ada_w":
            t = _normal(name, shape, 0.1 / math.sqrt(shape[1]), seed=seed)
        elif kind == "logslopes":
            # float64 libm then one rounding: identical on every host (fp32 vector log is not)
            base = torch.tensor([math.log(v) for v in alibi_default_slopes(shape[0])], dtype=torch.float64)
            t = (base.view(shape) + _normal(name, shape, 0.1, seed=seed).double()).float()
        else:
            raise KeyError(kind)
        sd[name] = t.contiguous()
    return sd


def make_speaker_table(num_speakers: int, dims: AcousticDims = AcousticDims(), seed: int = SEED) -> torch.Tensor:
    """`speaker_embedding.weight` [num_speakers, encoder dim] of a multi-speaker model (model.py:93-97), O(0.3) entries so that
    the speaker shift is visible against the O(1) encoder output."""
    return _normal("speaker_embedding.weight", (num_speakers, dims.text_dim), 0.3, seed=seed).contiguous()


def make_lengths(batch: int, text_max: int, mel_max: int, variable: bool, seed: int = SEED):
    """Fixed-length batches (BASELINE configs 2-3) or the variable-length rule of config 4:
    mel_len ~ U{mel_max/8 .. mel_max}, text_len = clamp(round(mel_len / 5.12), 25, text_max) <= mel_len.
    Item 0 always has the maximum lengths so the padded shapes are (text_max, mel_max)."""
    if not variable:
        return (torch.full((batch,), text_max, dtype=torch.int64), torch.full((batch,), mel_max, dtype=torch.int64))
    g = _rng(f"lengths/{batch}/{text_max}/{mel_max}", seed)
    mel_len = g.integers(max(mel_max // 8, 4), mel_max + 1, size=batch)
    mel_len[0] = mel_max
    lo = min(25, text_max)
    text_len = np.clip(np.rint(mel_len / (mel_max / text_max)).astype(np.int64), lo, text_max)
    text_len = np.minimum(text_len, mel_len)
    text_len[0] = text_max
    return torch.from_numpy(text_len.astype(np.int64)), torch.from_numpy(mel_len.astype(np.int64))


def make_inputs(batch: int, text_max: int = 100, mel_max: int = 512, variable: bool = False,
                dims: AcousticDims = AcousticDims(), seed: int = SEED) -> dict[str, torch.Tensor]:
    """Synthetic random-phoneme batch in the collator's layout (collator.py:36-55 of the reference):
    text int64 [B,L] zero-padded, mel fp32 [B,80,M], pitch/energy fp32 [B,M], lengths int64 [B]."""
    tag = f"inputs/{batch}/{text_max}/{mel_max}/{int(variable)}"
    g = _rng(tag, seed)
    text_len, mel_len = make_lengths(batch, text_max, mel_max, variable, seed)
    text = torch.from_numpy(g.integers(2, dims.vocab, size=(batch, text_max)).astype(np.int64))
    mel = torch.from_numpy((g.standard_normal((batch, dims.mel_dim, mel_max)) * 2.0 - 5.0).astype(np.float32))
    mel = mel.clamp_(min=math.log(1e-5))
    pitch = torch.from_numpy(g.standard_normal((batch, mel_max)).astype(np.float32))
    # float64 log1p rounded once to fp32 (host-independent, unlike torch's vectorised fp32 log1p)
    energy = torch.from_numpy(np.log1p(np.abs(g.standard_normal((batch, mel_max))) * 5.0).astype(np.float32))
    tmask = torch.arange(text_max)[None] < text_len[:, None]
    mmask = torch.arange(mel_max)[None] < mel_len[:, None]
    text = text * tmask
    mel = mel * mmask[:, None]
    pitch = pitch * mmask
    energy = energy * mmask
    # flow-matching noise is an INPUT here (generated on the host, SURVEY 7 "Randomness"):
    x0 = torch.from_numpy(g.standard_normal((batch, text_max, 3)).astype(np.float32))
    t = torch.from_numpy(g.random(batch).astype(np.float32))
    return {"text": text, "text_len": text_len, "mel": mel, "mel_len": mel_len, "pitch": pitch, "energy": energy,
            "flow_x0": x0, "flow_t": t}


def make_mas_logits(batch: int, mel_max: int, text_max: int, variable: bool = False, kind: str = "realistic",
                    seed: int = SEED):
    """Aligner-like MAS inputs, built from IEEE-exact operations only (+, -, *, /, max) so that every host produces
    the same bits: "realistic" = N(0,1) noise plus the log of the reference's diagonal prior
    (-(t/T - m/M)^2 / (2 * 0.1^2), floored at log(1e-6) like `log(prior + 1e-6)`, alignment.py:196; per-row
    normalisation constants are dropped because MAS is invariant to them); "ties" = small integers, which force the
    tie-breaking rule (ties -> diagonal, mas.py:17 of the reference)."""
    g = _rng(f"mas/{batch}/{mel_max}/{text_max}/{int(variable)}/{kind}", seed)
    text_len, mel_len = make_lengths(batch, text_max, mel_max, variable, seed)
    if kind == "ties":
        x = torch.from_numpy(g.integers(-3, 1, size=(batch, mel_max, text_max)).astype(np.float32))
    else:
        z = torch.from_numpy(g.standard_normal((batch, mel_max, text_max)).astype(np.float32))
        ti = torch.arange(text_max, dtype=torch.float32)[None, None, :] / text_len[:, None, None].float()
        mi = torch.arange(mel_max, dtype=torch.float32)[None, :, None] / mel_len[:, None, None].float()
        d = ti - mi
        x = z + torch.clamp(-(d * d) * 50.0, min=-13.815511)
    return x.contiguous(), text_len, mel_len
