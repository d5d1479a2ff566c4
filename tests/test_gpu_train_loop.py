"""Row f2 behind the reference's own training interface, and the benchmarked (bf16 AMP) step pinned against the oracle.

  * the three lines of the reference's loop body (experiments/trainer.py:543-549) - `outputs = model(**inputs)`;
    `loss, losses = criterion(inputs=inputs, outputs=outputs, step=...)`; `optimizer.step(loss, step_optimizer=...)` - drive the
    HIP backward unchanged and give bit for bit the gradients of `train.acoustic_train_forward`;
  * dropout masks of forward and backward agree when the backward runs where autograd runs it (its worker thread) under a seed
    source - the captured step's situation;
  * a tensor frozen after the optimizer was built does not move;
  * the AMP step (what `bench.py`'s `train_step` times) against the reference's fixture (B = 2) and, at the bench shape
    (B = 64 x 100 x 512, ragged), against autograd over the oracle on the host - with the bf16 bounds stated here.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import crc, golden  # noqa: E402
from isp_tts_amd import runtime, synth, train
from oracle import train_oracle as torc

pytestmark = pytest.mark.gpu
DEV = "cuda"

from amp_bounds import (AMP_GRAD_NORM_RTOL, AMP_GRAD_SAMPLE, AMP_LOSS_RTOL, AMP_SLOPE_FAMILY,     # noqa: E402  (the stated bounds)
                        check_full_gradients, tensor_class)


def _model(state_dict, train_mode=False):
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    m = AcousticModel.init(AcousticDims().model_config())
    m.load_state_dict(state_dict, strict=True)
    m = m.to(DEV)
    return m.train() if train_mode else m.eval()


def _collated(inp: dict) -> dict:
    """The collator's field names (data/collator.py:27-95) for a synthetic batch on the device."""
    return {"text_vector": inp["text"].to(DEV), "text_vector_len": inp["text_len"].to(DEV), "mel": inp["mel"].to(DEV),
            "mel_len": inp["mel_len"].to(DEV), "pitch": inp["pitch"].to(DEV), "energy": inp["energy"].to(DEV), "speaker": None}


@pytest.mark.parametrize("amp", [False, True])
def test_reference_loop_body_drives_the_hip_backward(state_dict, amp):
    """experiments/trainer.py:543-549 verbatim against `train.acoustic_train_forward` on a twin model: same gradient arena, same
    parameters after the step, bit for bit (train mode: the recipes' dropout; the seeds follow torch.manual_seed)."""
    batch = _collated(synth.make_inputs(3, 52, 160, variable=True, seed=9))
    model, twin = _model(state_dict, True), _model(state_dict, True)
    criterion = train.AcousticModelLoss()
    optimizer = train.FlatAdamW(model.parameters(), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
    opt_twin = train.FlatAdamW(twin.parameters(), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
    model.train_amp = amp
    global_step = 0
    for step_end in (False, True):        # one accumulation sub-step, then the step that updates
        torch.manual_seed(1234 + int(step_end))
        # ---- the reference's loop body -------------------------------------------------------------------------------------
        inputs = model.prepare_inputs(batch)
        outputs = model(**inputs)
        loss, losses = criterion(inputs=inputs, outputs=outputs, step=global_step)
        grad_norm = optimizer.step(loss, step_optimizer=step_end)
        # ----------------------------------------------------------------------------------------------------------------------
        assert loss.requires_grad is False or loss.grad_fn is not None
        torch.manual_seed(1234 + int(step_end))
        _, total, terms = train.acoustic_train_forward(twin, batch["text_vector"], batch["text_vector_len"], batch["mel"], batch["mel_len"],
                                                       batch["pitch"], batch["energy"], amp=amp)
        assert torch.equal(loss.detach(), total.detach())
        assert set(losses) == set(terms) and all(torch.equal(losses[k].detach(), terms[k].detach()) for k in terms)
        if not step_end:
            total.backward()
            assert grad_norm is None
            assert torch.equal(optimizer.flat.grad, opt_twin.flat.grad) and float(optimizer.flat.grad.abs().max()) > 0
        else:
            norm_twin = opt_twin.step(total)
            assert grad_norm is not None and torch.equal(grad_norm, norm_twin)
    assert torch.equal(optimizer.flat.data, opt_twin.flat.data)
    assert not torch.equal(optimizer.flat.data, train.FlatAdamW(_model(state_dict).parameters()).flat.data)    # (it did move)
    # the outputs are the reference's AcousticModelOutput, predictions included (values, no tape)
    ao = outputs.adaptor_output
    assert ao.log_duration.shape == ao.pitch.shape == ao.energy.shape == batch["text_vector"].shape
    assert outputs.mel.grad_fn is not None and outputs.aligner_output.attn_logits.grad_fn is not None


def test_autocast_selects_the_amp_step_and_no_grad_the_inference_kernels(state_dict):
    """`model(**inputs)` under `torch.autocast` (how the reference trains, recipes/default.yaml:56 through accelerate) is the AMP
    step; under `torch.no_grad()` (the reference's evaluation loop, trainer.py:534) it is the tape-free inference path."""
    batch = _collated(synth.make_inputs(2, 40, 120, variable=True, seed=3))
    model = _model(state_dict)
    inputs = model.prepare_inputs(batch)
    torch.manual_seed(5)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_ac = model(**inputs)
    model.train_amp = True
    torch.manual_seed(5)
    out_amp = model(**inputs)
    model.train_amp = False
    torch.manual_seed(5)
    out_32 = model(**inputs)
    assert torch.equal(out_ac.mel, out_amp.mel) and not torch.equal(out_amp.mel, out_32.mel)
    with torch.no_grad():
        out_ng = model(**inputs)
    assert out_ng.mel.grad_fn is None and not out_ng.mel.requires_grad
    assert (out_ng.mel - out_32.mel).abs().max().item() < 1e-4
    model.requires_grad_(False)           # every parameter frozen: nothing to differentiate, inference kernels
    assert model(**inputs).mel.grad_fn is None


def test_dropout_masks_of_forward_and_backward_agree_under_a_seed_source(state_dict):
    """The captured training step's masks come from (launch seed, device seed word).  autograd launches the backward kernels
    from ITS worker thread: they must fold in the same word as the forward kernels the calling thread launched (the source
    used to be thread-local - the backward then drew other masks, silently).  Check: with the word set, the directional
    derivative of a stack in train mode (attention and feed-forward dropout on) equals <gradient, direction> - it cannot if
    forward and backward disagree on which elements are dropped."""
    from isp_tts_amd.train.stack import transformer_train_forward
    model = _model(state_dict, True)
    tr = model.encoder
    B, L, D = 2, 48, 384
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(B, L, D, generator=g).to(DEV)
    v = torch.randn(B, L, D, generator=g).to(DEV)
    w = torch.randn(B, L, D, generator=g).to(DEV)
    lens = torch.tensor([48, 31], device=DEV)
    mask = torch.arange(L, device=DEV)[None] < lens[:, None]
    word = torch.tensor([0x5DEECE66D], dtype=torch.int64, device=DEV)

    def run(x):
        torch.manual_seed(11)           # the launch seeds are host draws that follow torch's seed
        return transformer_train_forward(tr, x, mask, False, key_len=lens)
    runtime.set_seed_source(word)
    try:
        x = x0.clone().requires_grad_()
        out = run(x)
        (out * w).sum().backward()      # backward kernels are launched by autograd's device thread
        analytic = float((x.grad.double() * v.double()).sum())
        eps = 1e-2
        with torch.no_grad():
            numeric = float(((run(x0 + eps * v) - run(x0 - eps * v)).double() * w.double()).sum() / (2 * eps))
            base = run(x0)
        word.fill_(0x1234567)           # another word: other masks from the same launch arguments
        with torch.no_grad():
            other = run(x0)
    finally:
        runtime.set_seed_source(None)
    with torch.no_grad():
        plain = run(x0)                 # no source: the launch seeds alone
    assert not torch.equal(base, other) and not torch.equal(base, plain)
    assert abs(analytic - numeric) <= 2e-2 * abs(numeric), (analytic, numeric)


def test_graphed_training_step_in_train_mode_trains(state_dict):
    """`train.GraphedTrainStep` with dropout ON: replays on one batch must reduce the loss like eager steps do (they did not
    have to while the backward's masks were not the forward's), the optimizer it was handed stays an eager optimizer, and a
    capture without warm-up steps still re-stages the weights inside the graph."""
    d = {k: v.to(DEV) for k, v in synth.make_inputs(3, 52, 160, variable=True, seed=9).items()}
    batch = {k: d[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy", "flow_x0", "flow_t")}

    def losses_of(graphed: bool, n=12):
        torch.manual_seed(21)
        m = _model(state_dict, True)
        o = train.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, grad_clip=1.0)
        out = []
        if graphed:
            # an eager forward with no optimizer step leaves CURRENT weight images in the cache; with no warm-up step either, the
            # capture would hit that cache and record no staging launch (replays would then train on capture-time weights) if
            # GraphedTrainStep did not invalidate the images first
            train.acoustic_train_forward(m, *[d[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy")],
                                         flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
            step = train.GraphedTrainStep(m, o, batch, amp=True, warmup=0)
            for _ in range(n):
                total, _, _ = step()
                out.append(float(total))
            assert o.check_finite is True and not hasattr(o, "args_dev")          # the optimizer object was not switched to graph mode
            step.close()
            _, total, _ = train.acoustic_train_forward(m, *[d[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy")],
                                                       flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
            assert o.step(total) is not None                                     # an eager step after the graph: host-side factors again
        else:
            for _ in range(n):
                _, total, _ = train.acoustic_train_forward(m, *[d[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy")],
                                                           flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
                o.step(total)
                out.append(float(total))
        return out
    eager, graph = losses_of(False), losses_of(True)
    drop_e, drop_g = eager[0] - min(eager[-3:]), graph[0] - min(graph[-3:])
    print(f"loss over 12 steps with dropout: eager {eager[0]:.3f} -> {eager[-1]:.3f}, graph replays {graph[0]:.3f} -> {graph[-1]:.3f}")
    assert drop_e > 0 and drop_g > 0.6 * drop_e, (eager, graph)


def test_a_parameter_frozen_after_the_optimizer_was_built_does_not_move(state_dict):
    """`model.freeze()` (base.py:66-73) AFTER `FlatAdamW` re-homed the parameters: autograd would drop the frozen tensors'
    gradients and torch.optim.AdamW would skip them (grad None: no decay, no moment update).  Same here: nothing is delivered
    into the arena for them, the update leaves value and moments alone; thawed again they train on."""
    d = {k: v.to(DEV) for k, v in synth.make_inputs(2, 40, 120, variable=True, seed=3).items()}
    model = _model(state_dict)
    opt = train.FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-1, grad_clip=1.0)

    def step():
        _, total, _ = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                                   flow_noise=d["flow_x0"], flow_time=d["flow_t"])
        return opt.step(total)
    step()
    frozen_names = ("decoder.layers.3.feed_forward.net.0.weight", "decoder.layers.3.attention_norm.bias", "to_mel.weight",
                    "aligner.attention.key_proj.0.conv.weight", "encoder.layers.0.attention.rel_pos.learned_logslopes")
    params = dict(model.named_parameters())
    for n in frozen_names:
        params[n].requires_grad_(False)
    index = {id(p): i for i, p in enumerate(opt.flat.params)}
    before = {n: params[n].detach().clone() for n in params}
    moments = {n: opt.exp_avg[opt.flat.offsets[index[id(params[n])]]:][:params[n].numel()].clone() for n in frozen_names}
    step()
    step()
    for n, p in params.items():
        if n in frozen_names:
            assert torch.equal(p.detach(), before[n]), n
            assert torch.equal(opt.exp_avg[opt.flat.offsets[index[id(p)]]:][:p.numel()], moments[n]), n
        else:
            assert not torch.equal(p.detach(), before[n]), n
    assert float(opt.flat.grad.abs().max()) == 0.0
    for n in frozen_names:
        params[n].requires_grad_(True)
    step()
    assert all(not torch.equal(params[n].detach(), before[n]) for n in frozen_names)


def _fixture_batch(g):
    inp = synth.make_inputs(2, 100, 512)
    text_len, mel_len = torch.tensor(g["text_len"]), torch.tensor(g["mel_len"])
    tm = torch.arange(100)[None] < text_len[:, None]
    mm = torch.arange(512)[None] < mel_len[:, None]
    text, mel = inp["text"] * tm, inp["mel"] * mm[:, None]
    pitch, energy = inp["pitch"] * mm, inp["energy"] * mm
    assert [crc(text), crc(mel), crc(pitch), crc(energy)] == [int(v) for v in g["inputs_crc"]]
    return [t.to(DEV) for t in (text, text_len, mel, mel_len, pitch, energy)], inp["flow_x0"].to(DEV), inp["flow_t"].to(DEV)


def test_amp_training_step_against_the_reference_fixture(state_dict):
    """the bf16-AMP step (dropout off) on the inputs of tests/golden/train.npz - the REFERENCE's own
    forward / loss / backward (fp32) - within the stated bf16 bounds: four losses and the total, and for each of the 206
    gradients its norm and its sampled entries."""
    g = golden("train.npz")
    args, x0, t = _fixture_batch(g)
    model = _model(state_dict)
    names = [str(n) for n in g["names"]]
    params = dict(model.named_parameters())
    assert list(params) == names
    _, total, losses = train.acoustic_train_forward(model, *args, flow_noise=x0, flow_time=t, amp=True, train_aligner=True)
    worst_loss = 0.0
    for k, v in list(losses.items()) + [("total", total)]:
        ref = float(g["loss_" + k.replace("/", "_")])
        worst_loss = max(worst_loss, abs(v.item() - ref) / max(abs(ref), 1e-3))
        assert abs(v.item() - ref) <= AMP_LOSS_RTOL * max(abs(ref), 1e-3), (k, v.item(), ref)
    total.backward()

    def sample(t_, n=192):
        f = t_.detach().reshape(-1)
        return f[::max(1, -(-f.numel() // n))].cpu()
    slope_scale = max(float(g["grad_norm"][i]) for i, n in enumerate(names) if tensor_class(n) == "slope")
    worst_s, worst_n, worst_sl = (0.0, ""), (0.0, ""), (0.0, "")
    for i, n in enumerate(names):
        gr, scale, ref_norm = params[n].grad, float(g["grad_absmax"][i]), float(g["grad_norm"][i])
        assert gr is not None, n
        ref_s = torch.from_numpy(g[f"g{i}"])
        if tensor_class(n) == "slope":       # (the fixture's sample of a 6-element tensor is the whole tensor)
            assert ref_s.numel() == gr.numel()
            e = float((gr.detach().reshape(-1).cpu().double() - ref_s.double()).norm()) / slope_scale
            worst_sl = max(worst_sl, (e, n))
            assert e <= AMP_SLOPE_FAMILY, (n, e)
            continue
        e_n = abs(gr.double().norm().item() - ref_norm) / max(ref_norm, 1e-12)
        e_s = (sample(gr) - ref_s).abs().max().item() / max(scale, 1e-12)
        worst_s, worst_n = max(worst_s, (e_s, n)), max(worst_n, (e_n, n))
        assert e_n <= AMP_GRAD_NORM_RTOL and e_s <= AMP_GRAD_SAMPLE, (n, e_n, e_s)
    print(f"AMP step vs the reference's fp32 step (B=2): losses {worst_loss:.2e} relative; worst gradient norm error {worst_n[0]:.2e} "
          f"({worst_n[1]}); worst sampled entry {worst_s[0]:.2e} of the tensor's scale ({worst_s[1]}); slopes {worst_sl[0]:.2e} of the "
          f"largest slope gradient ({worst_sl[1]})")


def _bench_shape_reference(state_dict, B=64):
    """Autograd over the oracle's forward at the bench shape on the host (fp32): (inputs, loss terms, total, gradients)."""
    inp = synth.make_inputs(B, 100, 512, variable=True)
    text, text_len, mel, mel_len, pitch, energy = (inp[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy"))
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith("freq_scale") else v.clone())
          for k, v in state_dict.items()}
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    total_ref, terms = torc.acoustic_losses(sd, text, text_len, mel, mel_len, pitch, energy, inp["flow_x0"], inp["flow_t"])
    total_ref.backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad}
    return inp, {k: float(v) for k, v in terms.items()}, float(total_ref), grads


@pytest.fixture(scope="module")
def bench_shape_reference(state_dict):
    return _bench_shape_reference(state_dict)


def _check_against(model, losses, total, ref_terms, ref_total, ref_grads, what):
    worst_l = 0.0
    for k, v in ref_terms.items():
        e = abs(float(losses[k]) - v) / max(abs(v), 1e-3)
        worst_l = max(worst_l, e)
        assert e <= AMP_LOSS_RTOL, (what, k, float(losses[k]), v)
    assert abs(float(total) - ref_total) <= AMP_LOSS_RTOL * abs(ref_total)
    grads = {n: p.grad for n, p in model.named_parameters()}
    assert len(grads) == 206 and all(v is not None for v in grads.values())
    print(f"{what}: losses within {worst_l:.2e} relative of the oracle's")
    print(check_full_gradients(grads, {n: ref_grads[n] for n in grads}, what))


def test_amp_training_step_at_the_bench_shape_against_the_oracle(state_dict, bench_shape_reference):
    """ONE B = 64 x 100 x 512 ragged step - the shape `bench.py`'s `train_step` times - bf16 AMP, dropout
    off, against torch autograd over `oracle.train_oracle.acoustic_losses` on the host: four losses to AMP_LOSS_RTOL, every
    one of the 206 gradients within the per-class bounds of tests/amp_bounds.py."""
    inp, ref_terms, ref_total, ref_grads = bench_shape_reference
    d = {k: v.to(DEV) for k, v in inp.items()}
    model = _model(state_dict)
    _, total, losses = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                                    flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
    total.backward()
    _check_against(model, losses, total, ref_terms, ref_total, ref_grads, "eager AMP step, B=64 x 512")


def test_graphed_amp_training_step_at_the_bench_shape_against_the_oracle(state_dict, bench_shape_reference):
    """the same step through `train.GraphedTrainStep` (what the bench's 13 ms figure runs).  A replay consumes
    its gradients (the arena is zeroed inside the graph), so the pin is transitive: the losses a replay returns are checked
    against the oracle's, and two replays must leave bit for bit the parameters and moments of two EAGER steps - whose
    gradients the test above holds against the oracle."""
    inp, ref_terms, ref_total, _ = bench_shape_reference
    d = {k: v.to(DEV) for k, v in inp.items()}
    batch = {k: d[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy", "flow_x0", "flow_t")}
    m_e, m_g = _model(state_dict), _model(state_dict)
    o_e = train.FlatAdamW(m_e.parameters(), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
    o_g = train.FlatAdamW(m_g.parameters(), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
    norms_e = []
    for _ in range(2):
        _, total, _ = train.acoustic_train_forward(m_e, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                                   flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
        norms_e.append(float(o_e.step(total)))
    step = train.GraphedTrainStep(m_g, o_g, batch, amp=True, warmup=0)
    assert torch.equal(o_g.flat.data, train.FlatAdamW(_model(state_dict).parameters()).flat.data)     # (the capture ran nothing)
    total, losses, norm = step()
    first = (float(total), {k: float(v) for k, v in losses.items()}, float(norm))
    step()
    torch.cuda.synchronize()
    for k, v in ref_terms.items():
        assert abs(first[1][k] - v) <= AMP_LOSS_RTOL * max(abs(v), 1e-3), (k, first[1][k], v)
    assert abs(first[0] - ref_total) <= AMP_LOSS_RTOL * abs(ref_total)
    assert first[2] == norms_e[0]
    assert torch.equal(o_g.flat.data, o_e.flat.data) and torch.equal(o_g.exp_avg, o_e.exp_avg) and torch.equal(o_g.exp_avg_sq, o_e.exp_avg_sq)
    assert o_g.step_count == o_e.step_count == 2
    step.close()
