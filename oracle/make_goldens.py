"""Generates tests/golden/*.npz by running the REAL reference (/root/reference) in the build container.

Run from the repo root (build container only; the reference does not exist on the GPU box):

    python3 oracle/make_goldens.py

The reference is imported read-only through `oracle/ref_shims` (config/logging/JIT-decorator plumbing only, see
its README); all arithmetic is the reference's own code on the installed torch/numpy.  Weights and inputs come
from `isp_tts_amd.synth` (deterministic, keyed by name), so fixtures hold only reference OUTPUTS plus small
inputs' checksums.  While generating, every output is also compared with the oracle restatement
(`oracle/acoustic_oracle.py`, `oracle/mas_oracle.c`) and the differences are printed: this is the pin of
the oracle against the reference.
"""
from __future__ import annotations

import json
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shims"))
sys.path.insert(1, "/root/reference")

from omegaconf import DictConfig  # noqa: E402  (shim)
from tts.models.acoustic.model import AcousticModel  # noqa: E402  (reference)
from tts.modules.aligner import b_mas as ref_b_mas  # noqa: E402  (reference)
from tts.modules.transformer import Attention, Transformer, TransformerLayer  # noqa: E402
from tts.modules.transformer.feedforward import FeedForward  # noqa: E402

from isp_tts_amd import synth  # noqa: E402
from isp_tts_amd.config import AcousticDims  # noqa: E402
from oracle import acoustic_oracle as orc  # noqa: E402
from oracle import mas_oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def crc(t) -> int:
    a = t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def report(name, ref, mine):
    ref, mine = torch.as_tensor(ref), torch.as_tensor(mine)
    if ref.dtype.is_floating_point:
        d = (ref.double() - mine.double()).abs().max().item()
        print(f"  oracle-vs-reference {name:28s} max|diff| = {d:.3e}   (ref max|x| = {ref.abs().max().item():.3f})")
        return d
    eq = bool((ref == mine).all())
    print(f"  oracle-vs-reference {name:28s} exact = {eq}")
    return 0.0 if eq else float("inf")


def build_reference(dims: AcousticDims):
    model = AcousticModel.init(DictConfig(dims.model_config())).eval()
    sd = synth.make_state_dict(dims)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state_dict key order differs from the reference"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=True)
    return model, sd


# --------------------------------------------------------------------------------------------- MAS
MAS_CASES = [  # (B, M, L, variable, kind)
    (4, 64, 17, False, "ties"),
    (4, 96, 23, True, "realistic"),
    (2, 3, 5, False, "realistic"),       # n < m
    (3, 40, 40, True, "ties"),
    (8, 512, 100, False, "realistic"),
    (4, 1024, 200, True, "realistic"),
]


def gen_mas():
    print("MAS")
    out = {}
    for idx, (B, M, L, var, kind) in enumerate(MAS_CASES):
        x, tl, ml = synth.make_mas_logits(B, M, L, var, kind)
        ref = ref_b_mas(x.numpy().copy(), in_lens=tl.numpy(), out_lens=ml.numpy())  # copy: the reference mutates
        mine, path = mas_oracle.b_mas(x.numpy(), tl.numpy(), ml.numpy(), return_path=True)
        report(f"mas[{B},{M},{L},{kind}]", ref, mine)
        ref_path = np.where(ref.any(axis=2), ref.argmax(axis=2), -1).astype(np.int16)
        assert (ref.sum(axis=2) <= 1).all()
        out[f"case{idx}_shape"] = np.array([B, M, L, int(var)], dtype=np.int64)
        out[f"case{idx}_kind"] = np.array(kind)
        out[f"case{idx}_logits_crc"] = np.array(crc(x), dtype=np.int64)
        out[f"case{idx}_path"] = ref_path
        out[f"case{idx}_dur"] = ref.sum(axis=1).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "mas.npz"), **out)


# --------------------------------------------------------------------------------------------- per-op
def gen_ops(model, sd):
    print("per-op")
    out = {}
    B = 2
    for tag, n, lens in (("enc", 100, [100, 73]), ("dec", 512, [512, 390])):
        x = synth._normal(f"golden/op/{tag}/x", (B, n, 384))
        mask = torch.arange(n)[None] < torch.tensor(lens)[:, None]
        stack = model.encoder if tag == "enc" else model.decoder
        pre = "encoder" if tag == "enc" else "decoder"
        layer = stack.layers[0]
        # Attention (attention.py:88-176)
        a_ref = layer.attention(x, mask=mask)[0]
        a_mine = orc.attention(sd, f"{pre}.layers.0.attention", x, mask, orc.alibi_int_bias(n, n))
        report(f"{tag}.attention", a_ref, a_mine)
        a_ref_nomask = layer.attention(x)[0]
        report(f"{tag}.attention(nomask)", a_ref_nomask,
               orc.attention(sd, f"{pre}.layers.0.attention", x, None, orc.alibi_int_bias(n, n)))
        # FeedForward (feedforward.py:39-40)
        f_ref = layer.feed_forward(x.clone())
        report(f"{tag}.feed_forward", f_ref, orc.feed_forward(sd, f"{pre}.layers.0.feed_forward", x))
        # TransformerLayer (transformer.py:62-118)
        l_ref = layer(x, mask=mask).out
        report(f"{tag}.layer", l_ref,
               orc.transformer_layer(sd, f"{pre}.layers.0", x, mask, None, orc.alibi_int_bias(n, n)))
        # Transformer (transformer.py:174-211)
        t_ref = stack(x, mask=mask).out
        report(f"{tag}.transformer", t_ref, orc.transformer(sd, pre, x, mask))
        t_ref_nomask = stack(x).out
        report(f"{tag}.transformer(nomask)", t_ref_nomask, orc.transformer(sd, pre, x, None))
        step = 1 if n <= 128 else 8   # keep fixtures small: every 8th row of the 512-frame outputs
        out.update({f"{tag}_lens": np.array(lens), f"{tag}_x_crc": np.array(crc(x), dtype=np.int64),
                    f"{tag}_row_step": np.array(step),
                    f"{tag}_attention": a_ref[:, ::step].numpy(),
                    f"{tag}_attention_nomask": a_ref_nomask[:, ::step].numpy(),
                    f"{tag}_feed_forward": f_ref[:, ::step].numpy(), f"{tag}_layer": l_ref[:, ::step].numpy(),
                    f"{tag}_transformer": t_ref[:, ::step].numpy(),
                    f"{tag}_transformer_nomask": t_ref_nomask[:, ::step].numpy()})
    # adaptive-norm stack (adaptor predictor: dim 256, 4 heads, AdaLN, project_emb 387 -> 256)
    n, lens = 100, [100, 73]
    x = synth._normal("golden/op/ada/x", (B, n, 387))
    cond = synth._normal("golden/op/ada/cond", (B, 32))
    mask = torch.arange(n)[None] < torch.tensor(lens)[:, None]
    tr = model.temporal_adaptor.predictor.transformer
    pre = "temporal_adaptor.predictor.transformer"
    t_ref = tr(x, mask=mask, adaptive_condition=cond).out
    report("ada.transformer", t_ref, orc.transformer(sd, pre, x, mask, cond))
    x256 = synth._normal("golden/op/ada/x256", (B, n, 256))
    l_ref = tr.layers[0](x256, mask=mask, adaptive_condition=cond).out
    report("ada.layer", l_ref, orc.transformer_layer(sd, f"{pre}.layers.0", x256, mask, cond, orc.alibi_int_bias(n, n)))
    cond3 = synth._normal("golden/op/ada/cond3", (1, 1, 32))
    t_ref3 = tr(x, mask=mask, adaptive_condition=cond3).out
    report("ada.transformer(cond[1,1,C])", t_ref3, orc.transformer(sd, pre, x, mask, cond3))
    out.update({"ada_lens": np.array(lens), "ada_x_crc": np.array(crc(x), dtype=np.int64),
                "ada_transformer": t_ref.numpy(), "ada_layer": l_ref.numpy(), "ada_transformer_cond3": t_ref3.numpy()})
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)


# --------------------------------------------------------------------------------------------- end-to-end
class _Noise:
    """Feeds the flow-matching draws of the reference from given tensors (draw order: temporal_adaptor.py:113-115
    `randn_like` then `rand`; :148 `randn`)."""

    def __init__(self, x0=None, t=None):
        self.x0, self.t = x0, t

    def __enter__(self):
        self.saved = (torch.randn_like, torch.rand, torch.randn)
        if self.x0 is not None:
            torch.randn_like = lambda x, *a, **k: self.x0.clone()
            torch.randn = lambda *a, **k: self.x0.clone()
        if self.t is not None:
            torch.rand = lambda *a, **k: self.t.clone()
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.rand, torch.randn = self.saved


def gen_forward(model, sd):
    print("forward (B=2, L=(100,73), M=(512,390))")
    inp = synth.make_inputs(2, 100, 512)
    text_len = torch.tensor([100, 73])
    mel_len = torch.tensor([512, 390])
    tm = torch.arange(100)[None] < text_len[:, None]
    mm = torch.arange(512)[None] < mel_len[:, None]
    text, mel = inp["text"] * tm, inp["mel"] * mm[:, None]
    pitch, energy = inp["pitch"] * mm, inp["energy"] * mm
    captured = {}
    hook = model.aligner.attention.register_forward_hook(
        lambda mod, args, res: captured.update(logits=res[1].clone(), soft=res[0].clone()))
    with _Noise(inp["flow_x0"], inp["flow_t"]):
        ref = model(text, text_len, mel, mel_len, pitch=pitch, energy=energy)
    hook.remove()
    mine = orc.acoustic_forward(sd, text, text_len, mel, mel_len, pitch, energy, inp["flow_x0"], inp["flow_t"])
    report("forward.mel", ref.mel, mine.mel)
    report("forward.attn_logits(pre-MAS)", captured["logits"], mine.aligner.attn_logits)
    report("forward.attn_soft", ref.aligner_output.attn_soft, mine.aligner.attn_soft)
    report("forward.attn_hard", ref.aligner_output.attn_hard, mine.aligner.attn_hard)
    report("forward.duration_target", ref.aligner_output.attn_hard_duration, mine.aligner.attn_hard_duration)
    ao = ref.adaptor_output
    report("forward.dec_lengths", ao.dec_lengths, mine.adaptor.dec_lengths)
    report("forward.log_duration", ao.log_duration, mine.adaptor.log_duration)
    report("forward.pitch", ao.pitch, mine.adaptor.pitch)
    report("forward.energy", ao.energy, mine.adaptor.energy)
    report("forward.pitch_target", ao.pitch_target, mine.adaptor.pitch_target)
    report("forward.flow_loss", ao.losses["flow_loss"], mine.adaptor.flow_loss)
    hard = ref.aligner_output.attn_hard.numpy()
    np.savez_compressed(
        os.path.join(OUT, "forward.npz"),
        text_len=text_len.numpy(), mel_len=mel_len.numpy(),
        inputs_crc=np.array([crc(text), crc(mel), crc(pitch), crc(energy)], dtype=np.int64),
        mel=ref.mel.numpy(), attn_logits=captured["logits"].numpy(),
        attn_soft_rows=ref.aligner_output.attn_soft[:, ::8].numpy(),
        path=np.where(hard.any(axis=2), hard.argmax(axis=2), -1).astype(np.int16),
        duration_target=ref.aligner_output.attn_hard_duration.numpy(),
        dec_lengths=ao.dec_lengths.numpy(), log_duration=ao.log_duration.numpy(), duration=ao.duration.numpy(),
        pitch=ao.pitch.numpy(), energy=ao.energy.numpy(), pitch_target=ao.pitch_target.numpy(),
        energy_target=ao.energy_target.numpy(), flow_loss=ao.losses["flow_loss"].numpy(),
        adaptor_enc_out_rows=ao.enc_out[:, ::16].numpy())


def gen_infer(model, sd):
    print("infer(steps=4)")
    out = {}
    inp = synth.make_inputs(2, 100, 512)
    text_len = torch.tensor([100, 73])
    text = inp["text"] * (torch.arange(100)[None] < text_len[:, None])
    # integer durations summing to 512 / 390 over the valid tokens
    dur = torch.zeros(2, 100, dtype=torch.int64)
    for b, (l, m) in enumerate(((100, 512), (73, 390))):
        base = m // l
        dur[b, :l] = base
        dur[b, : m - base * l] += 1
    x_t = inp["flow_x0"]
    with _Noise(x_t):
        mel_ref, ao = model.infer(text, text_lengths=text_len, duration_target=dur.clone(), steps=4)
    mel_mine, ad = orc.acoustic_infer(sd, text, text_len, dur, x_t, 4)
    report("infer.mel (B=2, dur target)", mel_ref, mel_mine)
    report("infer.pitch", ao.pitch, ad.pitch)
    report("infer.energy", ao.energy, ad.energy)
    report("infer.dec_lengths", ao.dec_lengths, ad.dec_lengths)
    out.update(b2_text_len=text_len.numpy(), b2_dur=dur.numpy(), b2_mel=mel_ref.numpy(), b2_pitch=ao.pitch.numpy(),
               b2_energy=ao.energy.numpy(), b2_dec_lengths=ao.dec_lengths.numpy())
    # predicted (fractional) durations, batch of 2
    with _Noise(x_t):
        mel_ref, ao = model.infer(text, text_lengths=text_len, steps=4)
    mel_mine, ad = orc.acoustic_infer(sd, text, text_len, None, x_t, 4)
    if mel_ref.shape == mel_mine.shape:
        report("infer.mel (B=2, predicted dur)", mel_ref, mel_mine)
    else:
        print("  shapes differ", mel_ref.shape, mel_mine.shape)
    report("infer.duration (predicted)", ao.duration, ad.duration)
    out.update(b2p_mel=mel_ref.numpy(), b2p_duration=ao.duration.numpy(), b2p_dec_lengths=ao.dec_lengths.numpy())
    # single utterance: no masks anywhere (model.py:191-201,228)
    x1 = x_t[:1]
    with _Noise(x1):
        mel_ref, ao = model.infer(text[:1], duration_target=dur[:1].clone(), steps=4)
    mel_mine, ad = orc.acoustic_infer(sd, text[:1], None, dur[:1], x1, 4)
    report("infer.mel (B=1)", mel_ref, mel_mine)
    out.update(b1_mel=mel_ref.numpy(), b1_pitch=ao.pitch.numpy(), b1_energy=ao.energy.numpy())
    np.savez_compressed(os.path.join(OUT, "infer.npz"), **out)


def gen_infer_speakers(sd):
    """A 4-speaker model (model.py:93-97): the reference's `infer` with `speaker` ids (:205-207) - a batch of 2 with the
    collator's [B, 1] field, and one utterance with the notebook's `torch.tensor([id])`.  (The reference's `forward` cannot
    run such a model: it reads a `speaker_encoder` attribute that does not exist, :145-146.)"""
    print("infer(steps=4), 4 speakers")
    dims = AcousticDims()
    model = AcousticModel.init(DictConfig(dict(dims.model_config(), num_speakers=4))).eval()
    sd4 = dict(sd)
    sd4["speaker_embedding.weight"] = synth.make_speaker_table(4, dims)
    assert set(model.state_dict()) == set(sd4)
    model.load_state_dict(sd4, strict=True)
    inp = synth.make_inputs(2, 100, 512)
    text_len = torch.tensor([100, 73])
    text = inp["text"] * (torch.arange(100)[None] < text_len[:, None])
    dur = torch.zeros(2, 100, dtype=torch.int64)
    for b, (l, m) in enumerate(((100, 512), (73, 390))):
        base = m // l
        dur[b, :l] = base
        dur[b, : m - base * l] += 1
    x_t = inp["flow_x0"]
    out = {}
    speaker = torch.tensor([[3], [1]])
    with _Noise(x_t):
        mel_ref, ao = model.infer(text, text_lengths=text_len, duration_target=dur.clone(), steps=4, speaker=speaker)
    mel_mine, ad = orc.acoustic_infer(sd4, text, text_len, dur, x_t, 4, speaker=speaker)
    report("infer.mel (B=2, speakers 3, 1)", mel_ref, mel_mine)
    report("infer.pitch", ao.pitch, ad.pitch)
    with _Noise(x_t):
        mel_plain, _ = model.infer(text, text_lengths=text_len, duration_target=dur.clone(), steps=4)
    print(f"  the speaker rows move the mel by {float((mel_ref - mel_plain).abs().max()):.3f} (max)")
    out.update(b2_speaker=speaker.numpy(), b2_text_len=text_len.numpy(), b2_dur=dur.numpy(), b2_mel=mel_ref.numpy(),
               b2_pitch=ao.pitch.numpy(), b2_energy=ao.energy.numpy(), b2_dec_lengths=ao.dec_lengths.numpy())
    one = torch.tensor([2])
    with _Noise(x_t[:1]):
        mel_ref, ao = model.infer(text[:1], duration_target=dur[:1].clone(), steps=4, speaker=one)
    mel_mine, ad = orc.acoustic_infer(sd4, text[:1], None, dur[:1], x_t[:1], 4, speaker=one)
    report("infer.mel (B=1, speaker 2)", mel_ref, mel_mine)
    out.update(b1_speaker=one.numpy(), b1_mel=mel_ref.numpy(), b1_pitch=ao.pitch.numpy())
    np.savez_compressed(os.path.join(OUT, "infer_speakers.npz"), **out)


# --------------------------------------------------------------------------------------------- training step (row f2)
def _sample(t: torch.Tensor, n: int = 192) -> np.ndarray:
    """A strided sample of a tensor's flattened values: first element, then every ceil(numel / n)-th."""
    f = t.detach().reshape(-1)
    step = max(1, -(-f.numel() // n))
    return f[::step].numpy().copy()


def gen_train(sd):
    """The reference's training step on the B=2 golden inputs: `AcousticModel.forward` under autograd (eval mode: no
    dropout draws), `AcousticModelLoss` (models/acoustic/loss.py:122-182), `.backward()`, the weight-decay grouping of
    experiments/optimizers.py:15-20 and one torch.optim.AdamW step with `clip_grad_norm_` on param_groups[0] only
    (optimizers.py:230-244; recipes/default.yaml: lr 2e-4, weight_decay 1e-2, grad_clip 1.0).  -> tests/golden/train.npz:
    loss values, per-parameter gradient norms + strided samples, the grouping, parameters after the step."""
    print("training step (B=2, L=(100,73), M=(512,390))")
    from tts.experiments.optimizers import group_weight_decayable_params  # noqa: E402  (reference)
    from tts.models.acoustic.loss import AcousticModelLoss  # noqa: E402  (reference)
    from oracle import train_oracle as torc
    model = AcousticModel.init(DictConfig(AcousticDims().model_config())).eval()
    model.load_state_dict(sd, strict=True)
    # On CPU the reference's MAS branch receives `attn_logits.detach().cpu().numpy()` - a VIEW of the autograd tensor's own
    # memory - and overwrites it in place (alignment.py:308, mas.py:12-16), so its CTC loss would read MAS's running sums
    # (with -inf entries: NaN gradients).  On the reference's training device (CUDA) `.cpu()` copies and the CUDA branch
    # clones (:320): the losses see the logits as the attention produced them.  The fixture holds THAT semantics: MAS gets a
    # copy.  (gen_forward captures the pre-MAS logits through a hook for the same reason.)
    cpu_mas = type(model.aligner).cpu_binarize_attention_parallel
    model.aligner.cpu_binarize_attention_parallel = lambda logits, tl, ml: cpu_mas(logits.clone(), tl, ml)
    criterion = AcousticModelLoss()
    inp = synth.make_inputs(2, 100, 512)
    text_len, mel_len = torch.tensor([100, 73]), torch.tensor([512, 390])
    tm = torch.arange(100)[None] < text_len[:, None]
    mm = torch.arange(512)[None] < mel_len[:, None]
    text, mel = inp["text"] * tm, inp["mel"] * mm[:, None]
    pitch, energy = inp["pitch"] * mm, inp["energy"] * mm
    names = [n for n, _ in model.named_parameters()]
    with torch.enable_grad():
        with _Noise(inp["flow_x0"], inp["flow_t"]):
            outputs = model(text, text_len, mel, mel_len, pitch=pitch, energy=energy)
        loss, losses = criterion({"text": text, "text_len": text_len, "mel": mel, "mel_len": mel_len, "pitch": pitch,
                                  "energy": energy}, outputs)
        loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads.values())
    # the oracle's restatement of the same step (autograd over the oracle's forward, composed as the reference composes it)
    osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith("freq_scale") else v.clone())
           for k, v in sd.items()}
    with torch.enable_grad():
        ototal, oterms = torc.acoustic_losses(osd, text, text_len, mel, mel_len, pitch, energy, inp["flow_x0"], inp["flow_t"])
        ototal.backward()
    report("train.total_loss", loss.detach(), ototal.detach())
    for k in losses:
        report(f"train.{k}", losses[k].detach(), oterms[k].detach())
    worst = max(((grads[n] - osd[n].grad).abs().max().item() / max(grads[n].abs().max().item(), 1e-12), n) for n in names)
    print(f"  oracle-vs-reference gradients: worst relative max|diff| over {len(names)} tensors = {worst[0]:.3e} ({worst[1]})")
    # optimizer: the reference's grouping, clip of group 0, one AdamW step
    params = list(model.parameters())
    wd, no_wd = group_weight_decayable_params(params)
    wd_ids = {id(p) for p in wd}
    opt = torch.optim.AdamW([{"params": wd}, {"params": no_wd, "weight_decay": 0.}], lr=2e-4, weight_decay=1e-2)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    grad_norm = torch.nn.utils.clip_grad_norm_(opt.param_groups[0]["params"], 1.0)
    opt.step()
    out = {"text_len": text_len.numpy(), "mel_len": mel_len.numpy(),
           "inputs_crc": np.array([crc(text), crc(mel), crc(pitch), crc(energy)], dtype=np.int64),
           "loss_total": loss.detach().numpy(), "grad_norm_group0": grad_norm.numpy(),
           "names": np.array(names), "in_decay_group": np.array([id(p) in wd_ids for p in params]),
           "lr": np.array(2e-4), "weight_decay": np.array(1e-2), "grad_clip": np.array(1.0)}
    for k, v in losses.items():
        out["loss_" + k.replace("/", "_")] = v.detach().numpy()
    out["grad_norm"] = np.array([grads[n].double().norm().item() for n in names])
    out["grad_absmax"] = np.array([grads[n].abs().max().item() for n in names])
    out["update_norm"] = np.array([(p.detach() - before[n]).double().norm().item() for n, p in model.named_parameters()])
    for i, (n, p) in enumerate(model.named_parameters()):
        out[f"g{i}"] = _sample(grads[n])
        out[f"u{i}"] = _sample(p.detach() - before[n])          # the update of one optimizer step, same stride
    np.savez_compressed(os.path.join(OUT, "train.npz"), **out)
    print(f"  losses: total {loss.item():.6f} " + " ".join(f"{k}={v.item():.6f}" for k, v in losses.items())
          + f"; clip norm of group 0 = {grad_norm.item():.4f}; {int(out['in_decay_group'].sum())} of {len(names)} tensors decay")


def gen_known_answers():
    print("known answers")
    dims = AcousticDims(vocab=77)
    model = AcousticModel.init(DictConfig(dims.model_config()))
    counts = {name: sum(p.numel() for p in getattr(model, name).parameters())
              for name in ("text_embedding", "encoder", "aligner", "temporal_adaptor", "decoder", "to_mel")}
    counts["total"] = sum(p.numel() for p in model.parameters())
    print("  ", counts)  # notebooks/inference.ipynb:313-665, training.ipynb:23180
    model149 = AcousticModel.init(DictConfig(AcousticDims().model_config()))
    keys = {k: list(v.shape) for k, v in model149.state_dict().items()}
    att = model149.encoder.layers[0].attention.rel_pos
    att4 = model149.temporal_adaptor.predictor.transformer.layers[0].attention.rel_pos
    ka = {
        "param_counts_vocab77": counts,
        "alibi_slopes_h6": att.slopes.flatten().tolist(),
        "alibi_slopes_h4": att4.slopes.flatten().tolist(),
        "euler_grid_steps4": orc.euler_grid(4).tolist(),
        "torch_version": torch.__version__, "numpy_version": np.__version__,
    }
    # Euler grid straight from the reference expression (temporal_adaptor.py:152-156)
    ts = -torch.diff(torch.logspace(0, 4, 5, base=0.75))
    ts = torch.cat([torch.tensor([0.]), ts])
    ka["euler_grid_steps4_ref"] = torch.cumsum(ts / ts.sum(), dim=0).tolist()
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_known_answers()
    gen_mas()
    model, sd = build_reference(AcousticDims())
    gen_ops(model, sd)
    gen_forward(model, sd)
    gen_infer(model, sd)
    gen_infer_speakers(sd)
    gen_train(sd)
    for f in sorted(os.listdir(OUT)):
        print(f"{f:28s} {os.path.getsize(os.path.join(OUT, f)) / 1e6:.2f} MB")
