"""GPU parity of the module mirror (Attention, FeedForward, TransformerLayer, Transformer, AcousticModel.forward/infer)
against the golden fixtures generated from the real reference, and against the oracle on other seeded inputs.

Tolerances: BASELINE.json's bar is mel L-inf < 1e-4 (fp32); per-op outputs are held to 5e-5."""
import numpy as np
import pytest
import torch

from conftest import crc, golden

pytestmark = pytest.mark.gpu

from isp_tts_amd import synth  # noqa: E402
from isp_tts_amd.modules.transformer import FeedForward, Transformer, TransformerLayer  # noqa: E402
from oracle import acoustic_oracle as orc  # noqa: E402

DEV = "cuda"
OP_TOL = 5e-5
MEL_TOL = 1e-4


def _maxdiff(a, b):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = b.detach().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a.double() - b.double()).abs().max().item()


@pytest.mark.parametrize("tag,n", [("enc", 100), ("dec", 512)])
def test_ops_against_reference_goldens(gpu_model, tag, n):
    g = golden("ops.npz")
    step = int(g[f"{tag}_row_step"])
    lens = torch.tensor(g[f"{tag}_lens"])
    x = synth._normal(f"golden/op/{tag}/x", (2, n, 384))
    assert crc(x) == int(g[f"{tag}_x_crc"])
    mask = (torch.arange(n)[None] < lens[:, None]).to(DEV)
    xd = x.to(DEV)
    stack = gpu_model.encoder if tag == "enc" else gpu_model.decoder
    layer = stack.layers[0]
    out, inter, shared = layer.attention(xd, mask=mask)
    assert _maxdiff(out[:, ::step], g[f"{tag}_attention"]) < OP_TOL
    assert inter.queries.shape == (2, 6, n, 64) and inter.keys.shape == (2, n, 64) and shared.rel_pos_bias is None
    assert _maxdiff(layer.attention(xd)[0][:, ::step], g[f"{tag}_attention_nomask"]) < OP_TOL
    assert _maxdiff(layer.feed_forward(xd)[:, ::step], g[f"{tag}_feed_forward"]) < OP_TOL
    assert _maxdiff(layer(xd, mask=mask).out[:, ::step], g[f"{tag}_layer"]) < OP_TOL
    assert _maxdiff(stack(xd, mask=mask).out[:, ::step], g[f"{tag}_transformer"]) < OP_TOL
    assert _maxdiff(stack(xd).out[:, ::step], g[f"{tag}_transformer_nomask"]) < OP_TOL


def test_adaptive_norm_stack_against_reference_goldens(gpu_model):
    g = golden("ops.npz")
    n, lens = 100, torch.tensor(g["ada_lens"])
    x = synth._normal("golden/op/ada/x", (2, n, 387))
    assert crc(x) == int(g["ada_x_crc"])
    cond = synth._normal("golden/op/ada/cond", (2, 32)).to(DEV)
    mask = (torch.arange(n)[None] < lens[:, None]).to(DEV)
    tr = gpu_model.temporal_adaptor.predictor.transformer
    assert _maxdiff(tr(x.to(DEV), mask=mask, adaptive_condition=cond).out, g["ada_transformer"]) < OP_TOL
    x256 = synth._normal("golden/op/ada/x256", (2, n, 256)).to(DEV)
    assert _maxdiff(tr.layers[0](x256, mask=mask, adaptive_condition=cond).out, g["ada_layer"]) < OP_TOL
    cond3 = synth._normal("golden/op/ada/cond3", (1, 1, 32)).to(DEV)
    assert _maxdiff(tr(x.to(DEV), mask=mask, adaptive_condition=cond3).out, g["ada_transformer_cond3"]) < OP_TOL


def _forward_inputs():
    inp = synth.make_inputs(2, 100, 512)
    text_len, mel_len = torch.tensor([100, 73]), torch.tensor([512, 390])
    tm = torch.arange(100)[None] < text_len[:, None]
    mm = torch.arange(512)[None] < mel_len[:, None]
    return dict(text=inp["text"] * tm, text_len=text_len, mel=inp["mel"] * mm[:, None], mel_len=mel_len,
                pitch=inp["pitch"] * mm, energy=inp["energy"] * mm, flow_noise=inp["flow_x0"], flow_time=inp["flow_t"])


def test_forward_against_reference_golden(gpu_model):
    g = golden("forward.npz")
    inp = _forward_inputs()
    assert [crc(inp[k]) for k in ("text", "mel", "pitch", "energy")] == [int(v) for v in g["inputs_crc"]]
    out = gpu_model(**{k: v.to(DEV) for k, v in inp.items()})
    torch.cuda.synchronize()
    assert out.mel.shape == (2, 80, 512)
    d = _maxdiff(out.mel, g["mel"])
    print(f"forward mel L-inf vs reference = {d:.3e}")
    assert d < MEL_TOL
    assert np.array_equal(out.adaptor_output.dec_lengths.cpu().numpy(), g["dec_lengths"])
    assert _maxdiff(out.aligner_output.attn_logits, g["attn_logits"]) < 1e-3      # PyTorch-ROCm conv front-end
    assert _maxdiff(out.aligner_output.attn_soft[:, ::8], g["attn_soft_rows"]) < 1e-4
    assert _maxdiff(out.adaptor_output.log_duration, g["log_duration"]) < 1e-3
    assert _maxdiff(out.adaptor_output.pitch, g["pitch"]) < 1e-3
    assert _maxdiff(out.adaptor_output.pitch_target, g["pitch_target"]) < 1e-4
    assert abs(float(out.adaptor_output.losses["flow_loss"]) - float(g["flow_loss"])) < 1e-3
    # MAS on the reference's own pre-MAS logits is bit-exact (the end-to-end path may differ at near-ties because the
    # front-end logits differ in the last bits, SURVEY 7 "MAS bit-exactness")
    hard = gpu_model.aligner.binarize_attention_parallel(torch.from_numpy(g["attn_logits"]).to(DEV),
                                                         inp["text_len"].to(DEV), inp["mel_len"].to(DEV))
    path = g["path"]
    ref_hard = np.zeros(hard.shape, np.int16)
    bi, mi = np.nonzero(path >= 0)
    ref_hard[bi, mi, path[bi, mi]] = 1
    assert hard.dtype == torch.int16 and np.array_equal(hard.cpu().numpy(), ref_hard)
    assert np.array_equal(out.aligner_output.attn_hard_duration.sum(1).cpu().numpy(), g["mel_len"])


def test_infer_against_reference_golden(gpu_model):
    g = golden("infer.npz")
    inp = synth.make_inputs(2, 100, 512)
    text_len = torch.tensor(g["b2_text_len"])
    text = (inp["text"] * (torch.arange(100)[None] < text_len[:, None])).to(DEV)
    dur = torch.from_numpy(g["b2_dur"]).to(DEV)
    x_t = inp["flow_x0"].to(DEV)
    mel, ao = gpu_model.infer(text, text_lengths=text_len.to(DEV), duration_target=dur, steps=4, flow_noise=x_t)
    assert _maxdiff(mel, g["b2_mel"]) < MEL_TOL
    assert _maxdiff(ao.pitch, g["b2_pitch"]) < MEL_TOL and _maxdiff(ao.energy, g["b2_energy"]) < MEL_TOL
    assert np.array_equal(ao.dec_lengths.cpu().numpy(), g["b2_dec_lengths"])
    # single utterance: no masks anywhere
    mel1, ao1 = gpu_model.infer(text[:1], duration_target=dur[:1], steps=4, flow_noise=x_t[:1])
    assert _maxdiff(mel1, g["b1_mel"]) < MEL_TOL
    # predicted (fractional) durations
    melp, aop = gpu_model.infer(text, text_lengths=text_len.to(DEV), steps=4, flow_noise=x_t)
    assert _maxdiff(aop.duration, g["b2p_duration"]) < 1e-3
    if np.array_equal(aop.dec_lengths.cpu().numpy(), g["b2p_dec_lengths"]):
        assert _maxdiff(melp, g["b2p_mel"]) < 5e-4   # soft path is continuous in the (fp32-noisy) predicted durations


def test_forward_matches_oracle_on_other_inputs(gpu_model, state_dict):
    """Different seed, B=3 variable lengths (not in the fixtures): HIP path vs the oracle."""
    inp = synth.make_inputs(3, 60, 200, variable=True, seed=7)
    args = (inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"], inp["energy"])
    ref = orc.acoustic_forward(state_dict, *args, inp["flow_x0"], inp["flow_t"])
    out = gpu_model(*[a.to(DEV) for a in args], flow_noise=inp["flow_x0"].to(DEV), flow_time=inp["flow_t"].to(DEV))
    assert _maxdiff(out.mel, ref.mel) < MEL_TOL
    assert np.array_equal(out.adaptor_output.dec_lengths.cpu().numpy(), ref.adaptor.dec_lengths.numpy())


def test_full_size_forward_properties(gpu_model):
    """BASELINE headline size (B=64, L=100, M=512): padding invariance and batch independence."""
    inp = synth.make_inputs(64, 100, 512, variable=True)
    dev = {k: v.to(DEV) for k, v in inp.items()}
    kw = dict(flow_noise=dev["flow_x0"], flow_time=dev["flow_t"])
    out = gpu_model(dev["text"], dev["text_len"], dev["mel"], dev["mel_len"], dev["pitch"], dev["energy"], **kw)
    assert out.mel.shape == (64, 80, 512) and torch.isfinite(out.mel).all()
    mm = torch.arange(512, device=DEV)[None] < dev["mel_len"][:, None]
    assert (out.mel * ~mm[:, None]).abs().max() == 0, "padded frames must be exactly zero"
    assert torch.equal(out.aligner_output.attn_hard_duration.sum(1), dev["mel_len"])
    # an utterance's mel does not depend on its batch neighbours
    sub = slice(5, 9)
    out4 = gpu_model(dev["text"][sub], dev["text_len"][sub], dev["mel"][sub], dev["mel_len"][sub], dev["pitch"][sub],
                     dev["energy"][sub], flow_noise=dev["flow_x0"][sub], flow_time=dev["flow_t"][sub])
    assert _maxdiff(out4.mel, out.mel[sub]) < 2e-5


# ------------------------------------------------------------------------------------------------ bf16 throughput path
BF16_MEL_TOL = 6e-2   # bf16 operands (8 mantissa bits) through 12 layers, fp32 residual stream; stated, not the 1e-4 bar


def test_bf16_forward_error_vs_reference(gpu_model):
    """The bf16 path (BASELINE config 3 dtype) against the fp32 reference golden: reported and bounded, NOT held to 1e-4."""
    g = golden("forward.npz")
    inp = _forward_inputs()
    try:
        gpu_model.set_compute_dtype(torch.bfloat16)
        out = gpu_model(**{k: v.to(DEV) for k, v in inp.items()})
        torch.cuda.synchronize()
    finally:
        gpu_model.set_compute_dtype(torch.float32)
    ref = torch.from_numpy(g["mel"])
    err = (out.mel.cpu() - ref).abs()
    rel_rms = (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"bf16 forward: mel L-inf = {err.max().item():.3e}, relative RMS = {rel_rms:.3e}")
    assert err.max().item() < BF16_MEL_TOL and rel_rms < 1e-2
    assert np.array_equal(out.adaptor_output.dec_lengths.cpu().numpy(), g["dec_lengths"])
    # back on the fp32 path the 1e-4 bar holds again (staged weights are rebuilt per dtype)
    out = gpu_model(**{k: v.to(DEV) for k, v in inp.items()})
    assert _maxdiff(out.mel, g["mel"]) < MEL_TOL


def test_bf16_full_size_forward_vs_fp32_path(gpu_model):
    """The benchmark configuration itself (B=64 x 512 frames, variable lengths: every decoder-sized fused path is on -
    fused feed-forward with its LayerNorm prologue and statistics epilogue, LayerNorm inside the q/kv GEMM) against the
    fp32 path of the same build (which holds the 1e-4 bar against the oracle): same bound as the small golden case."""
    inp = synth.make_inputs(64, 100, 512, variable=True)
    dev = {k: v.to(DEV) for k, v in inp.items()}
    args = (dev["text"], dev["text_len"], dev["mel"], dev["mel_len"], dev["pitch"], dev["energy"])
    kw = dict(flow_noise=dev["flow_x0"], flow_time=dev["flow_t"])
    ref = gpu_model(*args, **kw)
    try:
        gpu_model.set_compute_dtype(torch.bfloat16)
        out = gpu_model(*args, **kw)
        torch.cuda.synchronize()
    finally:
        gpu_model.set_compute_dtype(torch.float32)
    # the teacher-forced decoder input depends on the hard alignment only through durations of the SAME path; compare
    # the utterances whose MAS paths agree between the two precisions (the bf16 aligner may move a boundary by a frame)
    same = (out.aligner_output.attn_hard == ref.aligner_output.attn_hard).flatten(1).all(1)
    assert same.float().mean().item() >= 0.5, "most alignments must agree between the bf16 and the fp32 aligner"
    err = (out.mel[same] - ref.mel[same]).abs()
    rel_rms = (err.pow(2).mean().sqrt() / ref.mel[same].pow(2).mean().sqrt()).item()
    print(f"bf16 full size: {int(same.sum())}/64 identical alignments, mel L-inf = {err.max().item():.3e}, relative RMS = {rel_rms:.3e}")
    assert err.max().item() < BF16_MEL_TOL and rel_rms < 1e-2
    mm = torch.arange(512, device=DEV)[None] < dev["mel_len"][:, None]
    assert (out.mel * ~mm[:, None]).abs().max() == 0 and torch.isfinite(out.mel).all()
    assert torch.equal(out.adaptor_output.dec_lengths, ref.adaptor_output.dec_lengths)


def test_graph_lanes_reproduce_the_eager_forward(gpu_model):
    """HIP-graph replay on two alternating lanes (isp_tts_amd/graph.py:GraphedForwardLanes, what bench.py times) gives
    bit for bit the eager forward - including the side-stream branches, which become graph edges - and still does after
    new inputs are copied into a lane's static buffers; MAS durations equal the reference golden."""
    from isp_tts_amd.graph import GraphedForwardLanes
    g = golden("forward.npz")
    inp = {k: v.to(DEV) for k, v in _forward_inputs().items()}
    eager = gpu_model(**inp)
    torch.cuda.synchronize()
    lanes = GraphedForwardLanes(gpu_model, inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"],
                                inp["energy"], inp["flow_noise"], inp["flow_time"], lanes=2)
    outs = [lanes.replay() for _ in range(4)]
    torch.cuda.synchronize()
    for out in outs[-2:]:
        assert torch.equal(out.mel, eager.mel)
        assert torch.equal(out.adaptor_output.duration, eager.adaptor_output.duration)
        assert torch.equal(out.adaptor_output.losses["flow_loss"], eager.adaptor_output.losses["flow_loss"])
        assert np.array_equal(out.adaptor_output.dec_lengths.cpu().numpy(), g["dec_lengths"])
    # new inputs into lane 0's static buffers: utterance order swapped -> outputs swapped
    g0, s0 = lanes.lanes[0]
    with torch.cuda.stream(s0):
        swapped = g0(**{k: inp[k].flip(0) for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy", "flow_noise",
                                                     "flow_time")})
    torch.cuda.synchronize()
    assert _maxdiff(swapped.mel.flip(0), eager.mel) < 1e-5


def test_bf16_fused_ffn_matches_two_gemm_path_at_full_size(gpu_model):
    """Decoder stack at the benchmark shape (B=64 x 512 frames -> 32,768 rows, where FeedForward takes the fused
    ispk_ffn_bf16 kernel) against the same stack forced onto the two-GEMM path."""
    x = synth._normal("t/ffnfull/x", (64, 512, 384)).to(DEV)
    lens = torch.full((64,), 512, device=DEV)
    lens[1::3] = 300
    mask = torch.arange(512, device=DEV)[None] < lens[:, None]
    dec = gpu_model.decoder
    try:
        dec.set_compute_dtype(torch.bfloat16)
        fused = dec(x, mask=mask, key_len=lens).out
        for layer in dec.layers:
            layer.feed_forward.fused_min_rows = 1 << 30
        plain = dec(x, mask=mask, key_len=lens).out
    finally:
        for layer in dec.layers:
            layer.feed_forward.fused_min_rows = 128 * 128
        dec.set_compute_dtype(torch.float32)
    assert (fused - plain).abs().max() < 4e-2 and (fused - plain).pow(2).mean().sqrt() < 3e-3
    assert (fused * ~mask[..., None]).abs().max() == 0


def test_bf16_chained_layernorm_matches_separate_layernorm_at_full_size(gpu_model):
    """Decoder stack at the benchmark shape with every LayerNorm after a feed-forward emitted by the fused FFN kernel's
    epilogue (`Transformer.chain_layernorm`, opt-in) against the same stack with separate LayerNorm launches: the
    same fp32 two-pass statistics on the same values, so only 1-ulp bf16 roundings of the normalised rows differ (an ulp
    is 2^-7 at |value| in [1, 2)); through 6 layers they stay a fraction of an ulp in RMS."""
    x = synth._normal("t/chain/x", (64, 512, 384)).to(DEV)
    lens = torch.full((64,), 512, device=DEV)
    lens[2::5] = 211
    mask = torch.arange(512, device=DEV)[None] < lens[:, None]
    dec = gpu_model.decoder
    try:
        dec.set_compute_dtype(torch.bfloat16)
        dec.stats_layernorm = False
        plain = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
        del dec.stats_layernorm
        dec.chain_layernorm = True
        chained = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
    finally:
        dec.chain_layernorm = False
        dec.set_compute_dtype(torch.float32)
    assert chained.dtype == torch.bfloat16
    diff = (chained.float() - plain.float()).abs()
    assert diff.max() < 6e-2 and diff.pow(2).mean().sqrt() < 5e-3
    assert (chained.float() * ~mask[..., None]).abs().max() == 0


def test_bf16_layernorm_statistics_handoff_matches_separate_layernorm_at_full_size(gpu_model, monkeypatch):
    """Decoder stack at the benchmark shape with layers 2..6's attention LayerNorm applied inside the q/kv GEMM from the
    previous feed-forward kernel's row statistics (`Transformer.stats_layernorm`) against separate LayerNorm launches:
    same fp32 statistics up to the reduction tree, so only 1-ulp bf16 roundings of the normalised operand differ."""
    monkeypatch.setattr(FeedForward, "prenorm_fused", False)   # isolate the hand-off (ispk_ffn_bf16_ln -> ispk_gemm_bf16_lnin)
    x = synth._normal("t/chain/x", (64, 512, 384)).to(DEV)
    lens = torch.full((64,), 512, device=DEV)
    lens[2::5] = 211
    mask = torch.arange(512, device=DEV)[None] < lens[:, None]
    dec = gpu_model.decoder
    try:
        dec.set_compute_dtype(torch.bfloat16)
        dec.stats_layernorm = False
        plain = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
        dec.stats_layernorm = True
        handed = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
    finally:
        del dec.stats_layernorm              # back to the class default
        dec.set_compute_dtype(torch.float32)
    diff = (handed.float() - plain.float()).abs()
    assert diff.max() < 6e-2 and diff.pow(2).mean().sqrt() < 5e-3
    assert (handed.float() * ~mask[..., None]).abs().max() == 0


def test_bf16_prenorm_feed_forward_matches_separate_layernorm_at_full_size(gpu_model, monkeypatch):
    """Decoder stack at the benchmark shape with feed_forward_norm computed inside the fused feed-forward kernel
    (ispk_ffn_bf16_prenorm, the default) against separate LayerNorm launches (ISPK_FFN_PRENORM=0)."""
    x = synth._normal("t/chain/x", (64, 512, 384)).to(DEV)
    lens = torch.full((64,), 512, device=DEV)
    lens[2::5] = 211
    mask = torch.arange(512, device=DEV)[None] < lens[:, None]
    dec = gpu_model.decoder
    try:
        dec.set_compute_dtype(torch.bfloat16)
        fused = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
        monkeypatch.setattr(FeedForward, "prenorm_fused", False)
        monkeypatch.setattr(Transformer, "stats_layernorm", False)
        monkeypatch.setattr(FeedForward, "lnin_self", False)
        monkeypatch.setattr(TransformerLayer, "lnin_self", False)
        plain = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
    finally:
        dec.set_compute_dtype(torch.float32)
    diff = (fused.float() - plain.float()).abs()
    assert diff.max() < 6e-2 and diff.pow(2).mean().sqrt() < 5e-3
    assert (fused.float() * ~mask[..., None]).abs().max() == 0


def test_bf16_fused_output_projection_matches_separate_launches_at_full_size(gpu_model, monkeypatch):
    """Decoder stack at the benchmark shape with to_out + residual + mask inside the feed-forward kernel
    (ispk_attn_out_ffn_bf16, opt-in) against out-projection GEMM + pre-norm feed-forward kernel (the default)."""
    x = synth._normal("t/chain/x", (64, 512, 384)).to(DEV)
    lens = torch.full((64,), 512, device=DEV)
    lens[2::5] = 211
    mask = torch.arange(512, device=DEV)[None] < lens[:, None]
    dec = gpu_model.decoder
    try:
        dec.set_compute_dtype(torch.bfloat16)
        plain = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
        monkeypatch.setattr(TransformerLayer, "fuse_out_proj", True)   # opt-in (measured slower than the two launches)
        fused = dec(x, mask=mask, key_len=lens, out_dtype=torch.bfloat16).out
    finally:
        dec.set_compute_dtype(torch.float32)
    diff = (fused.float() - plain.float()).abs()
    assert diff.max() < 6e-2 and diff.pow(2).mean().sqrt() < 5e-3
    assert (fused.float() * ~mask[..., None]).abs().max() == 0


def test_bf16_encoder_with_layernorms_inside_the_gemms_matches_separate_layernorms(gpu_model, monkeypatch):
    """Encoder-sized stack (two-GEMM feed-forward): attention_norm inside the q/kv GEMM and feed_forward_norm inside the
    first feed-forward GEMM (ispk_gemm_bf16_lnin, statistics by the GEMM's own waves) against separate LayerNorm launches."""
    x = synth._normal("t/lnself/x", (16, 100, 384)).to(DEV)
    lens = torch.full((16,), 100, device=DEV)
    lens[1::3] = 37
    mask = torch.arange(100, device=DEV)[None] < lens[:, None]
    enc = gpu_model.encoder
    saved = [(l.lnin_self_min_rows, l.feed_forward.fused_min_rows) for l in enc.layers]
    try:
        enc.set_compute_dtype(torch.bfloat16)
        for l in enc.layers:                 # (by default only decoder-sized batches take these paths)
            l.lnin_self_min_rows = 0
            l.feed_forward.fused_min_rows = 0
            l.feed_forward.prenorm_fused = False
        fused = enc(x, mask=mask, key_len=lens).out
        monkeypatch.setattr(FeedForward, "lnin_self", False)
        monkeypatch.setattr(TransformerLayer, "lnin_self", False)
        plain = enc(x, mask=mask, key_len=lens).out
    finally:
        for l, (a, b) in zip(enc.layers, saved):
            l.lnin_self_min_rows, l.feed_forward.fused_min_rows, l.feed_forward.prenorm_fused = a, b, True
        enc.set_compute_dtype(torch.float32)
    diff = (fused - plain).abs()
    assert diff.max() < 6e-2 and diff.pow(2).mean().sqrt() < 5e-3
    assert (fused * ~mask[..., None]).abs().max() == 0


def test_config2_encoder_decoder_scope_fp32(gpu_model, state_dict):
    """BASELINE config 2 scope (TextEncoder + MelDecoder + to_mel on given activations, fp32) against the oracle."""
    tok = synth._normal("t/c2/tok", (4, 100, 384))
    dec_in = synth._normal("t/c2/dec", (4, 512, 384))
    tl, ml = torch.tensor([100, 80, 100, 33]), torch.tensor([512, 512, 301, 77])
    em = torch.arange(100)[None] < tl[:, None]
    dm = torch.arange(512)[None] < ml[:, None]
    enc_ref, mel_ref = orc.encoder_decoder(state_dict, tok, em, dec_in, dm)
    enc = gpu_model.encoder(tok.to(DEV), mask=em.to(DEV)).out
    dec = gpu_model.decoder(dec_in.to(DEV), mask=dm.to(DEV)).out
    mel = gpu_model._to_mel(dec, dm.to(DEV))
    assert _maxdiff(enc, enc_ref) < OP_TOL and _maxdiff(mel, mel_ref) < MEL_TOL


def test_rccl_all_gather_of_mel_single_rank():
    """The one collective of the path through RCCL itself (backend "nccl" on ROCm), world_size 1 on this one-GPU box;
    the world_size-2 logic is covered on CPU over gloo (tests/test_host_logic.py)."""
    import os
    import torch.distributed as dist
    from isp_tts_amd import dist as idist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 1000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        mel = synth._normal("t/rccl/mel", (5, 80, 64)).to(DEV)
        lens = torch.tensor([64, 10, 33, 64, 1], device=DEV)
        g, l = idist.all_gather_mel(mel, lens)
        assert g.shape == (1, 5, 80, 64) and torch.equal(g[0], mel) and torch.equal(l[0], lens)
        g2, l2 = idist.all_gather_mel(mel, lens, max_frames=64, max_batch=5)
        assert torch.equal(g2[0], mel)
        full, dec = idist.unshard(g, l, [[3, 0, 4, 1, 2]])
        assert torch.equal(full[3], mel[0]) and int(dec[2]) == 1
        # the overlapped pipeline of bench.py, as a gather to rank 0 (default) and as an all-gather
        for root in (0, None):
            pipe = idist.MelGatherPipeline(5, 80, 64, DEV, root=root)
            src, src_len = torch.empty_like(mel), torch.empty_like(lens)
            for step in range(3):
                src.copy_(mel + step)
                src_len.copy_(lens + step)
                pipe.submit(src, src_len)
                src.fill_(-1.0)
            g3, l3 = pipe.wait()
            assert torch.equal(g3[0], mel + 2) and torch.equal(l3[0], lens + 2)
    finally:
        dist.destroy_process_group()
