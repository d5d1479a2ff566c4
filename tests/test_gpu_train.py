"""Training-step kernels (first cut) against the oracle: torch autograd over the oracle's functions for the
gradients, torch.optim.AdamW + clip_grad_norm_ in the reference's grouping for the optimizer.  Tolerances are written per
test (floating point: fp32 kernels against an fp32 / fp64 CPU computation)."""
import pytest
import torch
import torch.nn.functional as F

from isp_tts_amd import runtime, synth, train
from isp_tts_amd.modules.transformer import Transformer
from oracle import acoustic_oracle as orc
from oracle import train_oracle as torc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _close(got, want, tol, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = max(want.abs().max().item(), 1e-30)
    assert err <= tol * ref, f"{what}: max |diff| {err:.3e} vs scale {ref:.3e} (tol {tol:g} relative)"


@pytest.mark.parametrize("M,N1,N2,masked", [(1000, 80, 384, False), (4099, 384, 1536, True), (64, 512, 384, True),
                                             (6400, 1536, 384, False)])
def test_gemm_tn_matches_float64(M, N1, N2, masked):
    """dW = dY^T X: odd row counts (tails of the 16-row steps and of the row ranges), N not a multiple of the 128-wide
    tile, row mask, accumulation; fixed summation order -> two runs give identical bits."""
    a, b = _rand((M, N1), 1), _rand((M, N2), 2)
    mask = (torch.arange(M) % 7 != 3) if masked else None
    want = ((a * mask[:, None]) if masked else a).double().T @ b.double()
    ad, bd, md = a.to(DEV), b.to(DEV), (mask.to(DEV) if masked else None)
    got = runtime.gemm_tn(ad, bd, row_mask=md)
    _close(got, want, 2e-6, "gemm_tn")
    assert torch.equal(got, runtime.gemm_tn(ad, bd, row_mask=md))
    acc = got.clone()
    runtime.gemm_tn(ad, bd, row_mask=md, out=acc, accumulate=True)
    _close(acc, 2 * want, 2e-6, "gemm_tn accumulate")
    # strided operands (column slices of wider matrices)
    wide = torch.cat([a, a], dim=1).to(DEV)
    _close(runtime.gemm_tn(wide[:, N1:], bd, row_mask=md), want, 2e-6, "gemm_tn strided")


def test_transpose():
    x = _rand((384, 1536), 3).to(DEV)
    assert torch.equal(runtime.transpose(x), x.T.contiguous())
    y = _rand((77, 130), 4).to(DEV)
    assert torch.equal(runtime.transpose(y), y.T.contiguous())


@pytest.mark.parametrize("dim,rows,masked", [(384, 1000, True), (256, 333, False), (384, 64 * 37 + 5, True)])
def test_layernorm_backward(dim, rows, masked):
    x = _rand((rows, dim), 5, 2.0).requires_grad_()
    gamma, beta = (1 + 0.1 * _rand((dim,), 6)).requires_grad_(), _rand((dim,), 7, 0.1).requires_grad_()
    dy = _rand((rows, dim), 8)
    mask = (torch.arange(rows) % 5 != 0) if masked else None
    y = F.layer_norm(x.double(), (dim,), gamma.double(), beta.double(), 1e-5)
    if masked:
        y = y * mask[:, None]
    y.backward(dy.double())
    dx, dg, db = runtime.layernorm_bwd(x.detach().to(DEV), dy.to(DEV), gamma.detach().to(DEV),
                                       row_mask=mask.to(DEV) if masked else None)
    _close(dx, x.grad, 2e-5, "dx")
    _close(dg, gamma.grad, 2e-5, "dgamma")
    _close(db, beta.grad, 2e-5, "dbeta")
    # accumulate into an existing gradient (the residual branch), no parameter gradients
    base = _rand((rows, dim), 9).to(DEV)
    dx2, dg2, _ = runtime.layernorm_bwd(x.detach().to(DEV), dy.to(DEV), gamma.detach().to(DEV),
                                        row_mask=mask.to(DEV) if masked else None, dx=base.clone(), add_to_dx=True,
                                        want_param_grads=False)
    assert dg2 is None
    _close(dx2, x.grad + base.cpu().double(), 2e-5, "dx accumulated")


def test_gelu_forward_backward():
    u = _rand((4096, 16), 10, 3.0).requires_grad_()
    da = _rand((4096, 16), 11)
    a = F.gelu(u.double())
    a.backward(da.double())
    _close(runtime.gelu(u.detach().to(DEV)), a, 1e-6, "gelu")
    _close(runtime.gelu_bwd(da.to(DEV), u.detach().to(DEV)), u.grad, 2e-6, "gelu'")


@pytest.mark.parametrize("B,N,H,lens", [(3, 100, 6, (100, 73, 33)), (2, 512, 6, (512, 390)), (2, 45, 4, None),
                                         (1, 1000, 6, (777,))])
def test_attention_backward_matches_autograd(B, N, H, lens):
    """dQ / dK / dV / d log-slope of ALiBi-MQA against float64 autograd over the reference's formulation (materialised
    bias, masked fill, softmax): ragged key lengths, N not a multiple of the 32-row tiles, 4 and 6 heads."""
    W = H * 64 + 128
    qkv = _rand((B, N, W), 12, 0.7)
    d_o = _rand((B, N, H * 64), 13)
    logs = torch.log(torch.tensor([2.0 ** (-(i + 1) * 8.0 / H) for i in range(H)])) + 0.1 * _rand((H,), 14)
    key_len = torch.tensor(lens) if lens is not None else None
    q64 = qkv.double().requires_grad_()
    l64 = logs.double().requires_grad_()
    q = q64[..., :H * 64].view(B, N, H, 64).transpose(1, 2)
    k, v = q64[..., H * 64:H * 64 + 64], q64[..., H * 64 + 64:]
    idx = torch.arange(N)
    bias = -(idx[None, :] - idx[:, None]).abs().double()[None] * l64.exp()[:, None, None]          # [H, N, N]
    s = torch.einsum("bhid,bjd->bhij", q, k) / 8.0 + bias[None]
    if key_len is not None:
        s = s.masked_fill(~(idx[None, :] < key_len[:, None])[:, None, None, :], float("-inf"))
    o = torch.einsum("bhij,bjd->bhid", s.softmax(-1), v).transpose(1, 2).reshape(B, N, H * 64)
    if key_len is not None:   # query rows beyond an utterance's length: the forward leaves them undefined, training masks d_o
        d_o = d_o * (idx[None, :] < key_len[:, None])[..., None]
    o.backward(d_o.double())
    qd, sd = qkv.to(DEV), logs.exp().to(DEV)
    kd = key_len.to(DEV) if key_len is not None else None
    o_gpu = runtime.alibi_mqa_attention(qd, H, sd, kd)
    if key_len is not None:
        valid = (idx[None, :] < key_len[:, None])[..., None]
        _close(o_gpu.cpu() * valid, o.detach() * valid, 5e-6, "forward o")
    else:
        _close(o_gpu, o, 5e-6, "forward o")
    dqkv, dls = runtime.alibi_mqa_attention_bwd(qd, o_gpu, d_o.to(DEV), H, sd, kd)
    _close(dqkv[..., :H * 64], q64.grad[..., :H * 64], 2e-5, "dQ")
    _close(dqkv[..., H * 64:H * 64 + 64], q64.grad[..., H * 64:H * 64 + 64], 2e-5, "dK")
    _close(dqkv[..., H * 64 + 64:], q64.grad[..., H * 64 + 64:], 2e-5, "dV")
    _close(dls, l64.grad, 5e-5, "dlogslopes")
    dqkv2, dls2 = runtime.alibi_mqa_attention_bwd(qd, o_gpu, d_o.to(DEV), H, sd, kd)
    assert torch.equal(dqkv, dqkv2) and torch.equal(dls, dls2), "attention backward is not reproducible"


def _decoder_stack(state_dict, depth):
    """The first `depth` layers + final norm of the MelDecoder stack (recipe configuration, this repo's synthetic weights),
    dropout set to 0 (the backward's current scope)."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    tr = model.decoder
    tr.layers = torch.nn.ModuleList(list(tr.layers)[:depth])
    for layer in tr.layers:
        layer.attention.attend.dropout = 0.
        layer.feed_forward.dropout_p = 0.
    return tr.to(DEV).train()


def _oracle_stack_sd(state_dict, depth):
    keep = {}
    for k, v in state_dict.items():
        parts = k.split(".")
        if parts[0] == "decoder" and (parts[1] != "layers" or int(parts[2]) < depth):
            keep[k] = v.clone().double().requires_grad_()
    return keep


def test_transformer_stack_gradients_match_autograd(state_dict):
    """TransformerStackFunction (2 decoder layers + final norm, ragged mask) against torch autograd over the oracle's
    `transformer`: output, d input and the gradient of every parameter."""
    tr = _decoder_stack(state_dict, 2)
    B, N, D = 3, 70, 384
    x = _rand((B, N, D), 20)
    lens = torch.tensor([70, 41, 64])
    mask = torch.arange(N)[None, :] < lens[:, None]
    dout = _rand((B, N, D), 21)
    sd = _oracle_stack_sd(state_dict, 2)
    xo = x.double().requires_grad_()
    out_ref = orc.transformer(sd, "decoder", xo, mask)
    out_ref.backward(dout.double())
    xg = x.to(DEV).requires_grad_()
    out = train.transformer_train_forward(tr, xg, mask.to(DEV))
    _close(out, out_ref, 2e-5, "stack output")
    out.backward(dout.to(DEV))
    _close(xg.grad, xo.grad, 1e-4, "d input")
    for name, p in tr.named_parameters():
        _close(p.grad, sd[f"decoder.{name}"].grad, 2e-4, f"d {name}")


def test_mel_loss_value_and_gradient():
    B, C, T = 5, 80, 300
    out, tgt = _rand((B, C, T), 30).requires_grad_(), _rand((B, C, T), 31)
    lens = torch.tensor([300, 17, 255, 1, 128])
    ref = 0.7 * torc.mel_loss(out.double(), tgt.double(), lens)
    ref.backward()
    og = out.detach().to(DEV).requires_grad_()
    loss = train.MelLoss(weight=0.7)(og, tgt.to(DEV), lens.to(DEV))
    _close(loss, ref, 2e-6, "mel loss")
    loss.backward()
    _close(og.grad, out.grad, 2e-6, "d mel")
    assert float(og.grad[1, :, 17:].abs().max()) == 0.0


def test_flat_adamw_matches_torch_adamw():
    """FlatAdamW (sqnorm + fused clip + AdamW over the arena) against torch.optim.AdamW with the reference's grouping and
    clip_grad_norm_ on group 0 only (optimizers.py:15-20, :34-40, :236-237): 6 steps with gradients large enough that the
    clip is active on some steps and idle on others; then the checkpoint layout round trip."""
    shapes = [(384, 384), (7,), (1536, 384), (384,), (1, 80, 1), (6, 1, 1), (130, 3)]
    ref_p = [torch.nn.Parameter(_rand(s, 40 + i, 0.3)) for i, s in enumerate(shapes)]
    gpu_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    ref = torc.reference_optimizer(ref_p, lr=2e-3, weight_decay=1e-2)
    opt = train.FlatAdamW(gpu_p, lr=2e-3, weight_decay=1e-2, grad_clip=1.0)
    wd_ref, _ = torc.group_weight_decayable_params(ref_p)
    assert opt.flat.n_decay_tensors == len(wd_ref) == 3
    for step in range(6):
        scale = 1e-3 if step % 2 else 3e-2          # norm of group 0 ~ 0.9 / 27: idle / active clip
        for i, (a, b) in enumerate(zip(ref_p, gpu_p)):
            g = _rand(a.shape, 100 * step + i, scale)
            a.grad = g.clone()
            b.grad.copy_(g)
        n_ref = torc.reference_step(ref, 1.0)
        n_gpu = opt.step()
        _close(n_gpu, n_ref, 1e-5, f"grad norm, step {step}")
        for i, (a, b) in enumerate(zip(ref_p, gpu_p)):
            _close(b, a, 2e-6, f"parameter {i} after step {step}")
            assert float(b.grad.abs().max()) == 0.0
    sd = opt.state_dict()
    ref_sd = ref.state_dict()
    assert [g["params"] for g in sd["optimizer"]["param_groups"]] == [g["params"] for g in ref_sd["param_groups"]]
    for i in range(len(shapes)):
        # (the clip coefficient comes from a norm summed in another order than torch's fp32 one: a few ulps of g per step,
        # squared and accumulated in exp_avg_sq)
        _close(sd["optimizer"]["state"][i]["exp_avg"], ref_sd["state"][i]["exp_avg"], 1e-5, f"exp_avg {i}")
        _close(sd["optimizer"]["state"][i]["exp_avg_sq"], ref_sd["state"][i]["exp_avg_sq"], 5e-5, f"exp_avg_sq {i}")
        assert float(ref_sd["state"][i]["step"]) == float(sd["optimizer"]["state"][i]["step"]) == 6.0
    opt2 = train.FlatAdamW([torch.nn.Parameter(p.detach().clone()) for p in gpu_p], lr=2e-3, weight_decay=1e-2)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 6 and torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    # Frozen tensors (AcousticModel.freeze, row f3) keep their place in torch.optim's numbering: the state of a torch AdamW over
    # ALL tensors (two of them frozen) loads into the right arena slots, and the saved layout numbers like torch's.
    ref_p2 = [torch.nn.Parameter(_rand(s, 70 + i, 0.3)) for i, s in enumerate(shapes)]
    for i in (0, 3):
        ref_p2[i].requires_grad_(False)
    ref2 = torc.reference_optimizer(ref_p2, lr=2e-3, weight_decay=1e-2)
    for i, a in enumerate(ref_p2):
        if a.requires_grad:
            a.grad = _rand(a.shape, 900 + i, 1e-2)
    torc.reference_step(ref2, 1.0)
    gpu_p2 = [torch.nn.Parameter(p.detach().clone().to(DEV), requires_grad=p.requires_grad) for p in ref_p2]
    opt3 = train.FlatAdamW(gpu_p2, lr=1e-3, weight_decay=1e-2)
    ref2_sd = ref2.state_dict()
    ref2_sd["param_groups"][0]["initial_lr"] = 5e-3
    opt3.load_state_dict({"optimizer": ref2_sd, "lr_scheduler": {"gamma": 0.995, "last_epoch": 3}})
    assert opt3.base_lr == 5e-3 and opt3.last_epoch == 3
    sd3 = opt3.state_dict()["optimizer"]
    assert [g["params"] for g in sd3["param_groups"]] == [g["params"] for g in ref2_sd["param_groups"]]
    assert sorted(sd3["state"]) == sorted(ref2_sd["state"])
    for k in ref2_sd["state"]:
        assert torch.equal(sd3["state"][k]["exp_avg"].cpu(), ref2_sd["state"][k]["exp_avg"])


def test_stack_training_steps_reduce_the_loss(state_dict):
    """End to end on the GPU: decoder stack (2 layers) forward -> loss -> backward kernels -> FlatAdamW, 8 steps on one
    batch; the loss falls and the staged weight images follow the arena (a stale image would freeze the loss)."""
    tr = _decoder_stack(state_dict, 2)
    opt = train.FlatAdamW(tr, lr=1e-3, weight_decay=1e-2, grad_clip=1.0)
    B, N = 4, 96
    x = _rand((B, N, 384), 50).to(DEV)
    target = _rand((B, N, 384), 51, 0.5).to(DEV)
    mask = (torch.arange(N)[None, :] < torch.tensor([96, 80, 50, 96])[:, None]).to(DEV)
    losses = []
    for _ in range(8):
        out = train.transformer_train_forward(tr, x, mask)
        loss = (((out - target) * mask[..., None]) ** 2).mean()
        losses.append(float(loss.detach()))
        opt.step(loss)
    assert losses[-1] < 0.9 * losses[0] and all(b < a for a, b in zip(losses, losses[1:])), losses


def test_to_mel_backward_and_colsum(state_dict):
    """ToMelFunction (Linear + transpose + mask, model.py:167-168) against float64 autograd: d dec, dW, db; ragged mask."""
    B, T, D, C = 3, 77, 384, 80
    dec = _rand((B, T, D), 60).requires_grad_()
    w = state_dict["to_mel.weight"].clone().requires_grad_()
    bias = state_dict["to_mel.bias"].clone().requires_grad_()
    mask = torch.arange(T)[None, :] < torch.tensor([77, 40, 64])[:, None]
    dmel = _rand((B, C, T), 61)
    ref = (F.linear(dec.double(), w.double(), bias.double()).transpose(1, 2) * mask[:, None, :])
    ref.backward(dmel.double())
    dg, wg, bg = dec.detach().to(DEV).requires_grad_(), w.detach().to(DEV).requires_grad_(), bias.detach().to(DEV).requires_grad_()
    mel = train.ToMelFunction.apply(dg, wg, bg, mask.to(DEV))
    _close(mel, ref, 5e-6, "mel")
    mel.backward(dmel.to(DEV))
    _close(dg.grad, dec.grad, 2e-5, "d dec")
    _close(wg.grad, w.grad, 2e-5, "d to_mel.weight")
    _close(bg.grad, bias.grad, 2e-5, "d to_mel.bias")
    x = _rand((1000, 80), 62)
    _close(runtime.colsum(x.to(DEV)), x.double().sum(0), 2e-6, "colsum")


def test_decoder_to_mel_loss_training_step_matches_reference_step(state_dict):
    """One whole optimizer step of MelDecoder (2 layers) + to_mel under the mel loss: HIP forward / backward / FlatAdamW
    against the oracle's forward + torch autograd + torch.optim.AdamW in the reference's grouping with clip 1.0
    (optimizers.py:230-244).  The loss agrees to 1e-5, the clipped gradient norm to 1e-4, the parameters after the step to a
    small fraction of the learning rate (gradients themselves are compared tightly in the tests above)."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    depth = 2
    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model.decoder.layers = torch.nn.ModuleList(list(model.decoder.layers)[:depth])
    for layer in model.decoder.layers:
        layer.attention.attend.dropout = 0.
        layer.feed_forward.dropout_p = 0.
    model = model.to(DEV).train()
    B, T = 3, 90
    dec_in, target = _rand((B, T, 384), 70), _rand((B, 80, T), 71)
    lens = torch.tensor([90, 55, 72])
    mask = torch.arange(T)[None, :] < lens[:, None]
    # reference step on CPU (fp32, as the GPU path)
    sd = {k: v.clone().requires_grad_() for k, v in state_dict.items()
          if k.startswith("to_mel.") or (k.startswith("decoder.") and (k.split(".")[1] != "layers" or int(k.split(".")[2]) < depth))}
    names = list(sd)
    ref_opt = torc.reference_optimizer([sd[k] for k in names], lr=1e-3, weight_decay=1e-2)
    dec = orc.transformer({k: v for k, v in sd.items() if k.startswith("decoder.")}, "decoder", dec_in, mask)
    mel_ref = F.linear(dec, sd["to_mel.weight"], sd["to_mel.bias"]).transpose(1, 2) * mask[:, None, :]
    loss_ref = torc.mel_loss(mel_ref, target, lens)
    loss_ref.backward()
    norm_ref = torc.reference_step(ref_opt, 1.0)
    # the same step on the GPU
    params = list(model.decoder.parameters()) + list(model.to_mel.parameters())
    opt = train.FlatAdamW(params, lr=1e-3, weight_decay=1e-2, grad_clip=1.0)
    mel = train.mel_decoder_train_forward(model, dec_in.to(DEV), mask.to(DEV))
    loss = train.MelLoss()(mel, target.to(DEV), lens.to(DEV))
    _close(loss, loss_ref, 1e-5, "loss")
    norm = opt.step(loss)
    _close(norm, norm_ref, 1e-4, "gradient norm of the clipped group")
    got = {f"decoder.{n}": p for n, p in model.decoder.named_parameters()}
    got.update({f"to_mel.{n}": p for n, p in model.to_mel.named_parameters()})
    # Adam's first step moves every element by ~lr * g / (|g| + eps): where |g| is within rounding noise of 0 the quotient
    # amplifies that noise, so the bound on single elements is a fraction of lr; the mean difference stays far below it
    lr = 1e-3
    for k in names:
        d = (got[k].detach().cpu() - sd[k].detach()).abs()
        assert float(d.max()) <= 0.1 * lr and float(d.mean()) <= 2e-3 * lr, (k, float(d.max()), float(d.mean()))


# ------------------------------------------------------------------------------------------------ dropout
def test_dropout_mask_rate_and_gelu_dropout():
    """The in-kernel mask: keep rate 1 - p within 4 sigma, different seeds give different masks, the same seed the same;
    GELU + dropout forward / backward equal float64 with the exported mask."""
    n, p = 1 << 20, 0.1
    m1, m2 = runtime.dropout_mask(n, p, 1234, DEV), runtime.dropout_mask(n, p, 1235, DEV)
    assert torch.equal(m1, runtime.dropout_mask(n, p, 1234, DEV)) and not torch.equal(m1, m2)
    for m in (m1, m2):
        assert abs(float(m.float().mean()) - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5
    assert bool(runtime.dropout_mask(1000, 0.0, 7, DEV).all())
    u = _rand((2048, 16), 80, 2.0).requires_grad_()
    da = _rand((2048, 16), 81)
    keep = runtime.dropout_mask(u.numel(), 0.3, 99, DEV).cpu().view_as(u)
    ref = F.gelu(u.double()) * keep / 0.7
    ref.backward(da.double())
    _close(runtime.gelu(u.detach().to(DEV), 0.3, 99), ref, 1e-6, "gelu + dropout")
    _close(runtime.gelu_bwd(da.to(DEV), u.detach().to(DEV), dropout_p=0.3, seed=99), u.grad, 2e-6, "gelu + dropout backward")


@pytest.mark.parametrize("B,N,H,lens,p", [(2, 100, 6, (100, 61), 0.1), (1, 160, 4, None, 0.3)])
def test_attention_dropout_forward_backward(B, N, H, lens, p):
    """ispk_alibi_mqa_attn_train_f32 / _bwd_f32 with dropout on the attention probabilities against float64 autograd that
    applies the SAME mask (exported by ispk_dropout_mask_u8, index ((b H + h) N + i) N + j); the training forward with p = 0
    equals the inference kernel; the backward with the forward's log-sum-exp equals the one that recomputes it."""
    W = H * 64 + 128
    qkv = _rand((B, N, W), 90, 0.7)
    d_o = _rand((B, N, H * 64), 91)
    logs = torch.log(torch.tensor([2.0 ** (-(i + 1) * 8.0 / H) for i in range(H)]))
    key_len = torch.tensor(lens) if lens is not None else None
    idx = torch.arange(N)
    if key_len is not None:
        d_o = d_o * (idx[None, :] < key_len[:, None])[..., None]
    seed = 4242
    keep = runtime.dropout_mask(B * H * N * N, p, seed, DEV).cpu().view(B, H, N, N)
    q64 = qkv.double().requires_grad_()
    l64 = logs.double().requires_grad_()
    q = q64[..., :H * 64].view(B, N, H, 64).transpose(1, 2)
    k, v = q64[..., H * 64:H * 64 + 64], q64[..., H * 64 + 64:]
    bias = -(idx[None, :] - idx[:, None]).abs().double()[None] * l64.exp()[:, None, None]
    s = torch.einsum("bhid,bjd->bhij", q, k) / 8.0 + bias[None]
    if key_len is not None:
        s = s.masked_fill(~(idx[None, :] < key_len[:, None])[:, None, None, :], float("-inf"))
    pd = s.softmax(-1) * keep / (1 - p)
    o = torch.einsum("bhij,bjd->bhid", pd, v).transpose(1, 2).reshape(B, N, H * 64)
    o.backward(d_o.double())
    qd, sd = qkv.to(DEV), logs.exp().to(DEV)
    kd = key_len.to(DEV) if key_len is not None else None
    o_gpu, lse = runtime.alibi_mqa_attention_train(qd, H, sd, kd, p, seed)
    valid = (idx[None, :] < key_len[:, None])[..., None] if key_len is not None else torch.ones(B, N, 1, dtype=torch.bool)
    _close(o_gpu.cpu() * valid, o.detach() * valid, 5e-6, "forward with dropout")
    dqkv, dls = runtime.alibi_mqa_attention_bwd(qd, o_gpu, d_o.to(DEV), H, sd, kd, lse=lse, dropout_p=p, seed=seed)
    _close(dqkv, q64.grad, 3e-5, "dqkv with dropout")
    _close(dls, l64.grad, 1e-4, "dlogslopes with dropout")
    dqkv2, dls2 = runtime.alibi_mqa_attention_bwd(qd, o_gpu, d_o.to(DEV), H, sd, kd, dropout_p=p, seed=seed)
    _close(dqkv2, dqkv, 1e-6, "backward with recomputed statistics")
    o0, _ = runtime.alibi_mqa_attention_train(qd, H, sd, kd, 0.0, 0)
    _close(o0.cpu() * valid, runtime.alibi_mqa_attention(qd, H, sd, kd).cpu() * valid, 5e-6, "training forward, p = 0")


def test_stack_training_with_the_recipes_dropout(state_dict):
    """The decoder stack in training mode with the recipes' dropout (0.1 / 0.1): a step is reproducible under
    torch.manual_seed, differs between seeds, equals the dropout-free step when p = 0, and the loss falls over 8 steps."""
    tr = _decoder_stack(state_dict, 2)
    for layer in tr.layers:
        layer.attention.attend.dropout = 0.1
        layer.feed_forward.dropout_p = 0.1
    B, N = 3, 64
    x = _rand((B, N, 384), 95).to(DEV)
    mask = (torch.arange(N)[None, :] < torch.tensor([64, 40, 57])[:, None]).to(DEV)
    dout = _rand((B, N, 384), 96).to(DEV)

    def run(seed):
        torch.manual_seed(seed)
        xg = x.clone().requires_grad_()
        out = train.transformer_train_forward(tr, xg, mask)
        out.backward(dout)
        g = tr.layers[0].feed_forward.net[0].weight.grad.clone()
        tr.zero_grad(set_to_none=True)
        return out.detach().clone(), xg.grad.clone(), g
    a, b, c = run(5), run(5), run(6)
    assert all(torch.equal(p, q) for p, q in zip(a, b)), "same seed, different step"
    assert not torch.equal(a[0], c[0])
    tr.eval()
    assert torch.equal(run(5)[0], run(6)[0]), "eval mode must not drop anything"
    tr.train()
    opt = train.FlatAdamW(tr, lr=1e-3, weight_decay=1e-2, grad_clip=1.0)
    target = _rand((B, N, 384), 97, 0.5).to(DEV)
    torch.manual_seed(0)
    losses = []
    for _ in range(8):
        out = train.transformer_train_forward(tr, x, mask)
        loss = (((out - target) * mask[..., None]) ** 2).mean()
        losses.append(float(loss.detach()))
        opt.step(loss)
    assert losses[-1] < 0.9 * losses[0], losses


def test_stack_gradients_under_bf16_amp(state_dict):
    """`amp=True`: the Linear GEMMs (forward and dX) take bf16 operands, everything else stays fp32.  Against the fp32
    kernels on the same input: output and gradients agree to bf16 accuracy (relative RMS, stated per quantity)."""
    tr = _decoder_stack(state_dict, 2)
    B, N = 3, 70
    x = _rand((B, N, 384), 20).to(DEV)
    mask = (torch.arange(N)[None, :] < torch.tensor([70, 41, 64])[:, None]).to(DEV)
    dout = _rand((B, N, 384), 21).to(DEV)

    def run(amp):
        xg = x.clone().requires_grad_()
        out = train.transformer_train_forward(tr, xg, mask, amp=amp)
        out.backward(dout)
        g = {n: p.grad.clone() for n, p in tr.named_parameters()}
        tr.zero_grad(set_to_none=True)
        return out.detach(), xg.grad, g
    o32, dx32, g32 = run(False)
    o16, dx16, g16 = run(True)

    def rel(a, b):
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))
    assert rel(o16, o32) < 1e-2 and rel(dx16, dx32) < 3e-2, (rel(o16, o32), rel(dx16, dx32))
    for n in g32:
        assert rel(g16[n], g32[n]) < 5e-2, (n, rel(g16[n], g32[n]))
    assert not torch.equal(o16, o32)


def test_amp_weight_gradients_do_not_see_padded_rows(state_dict):
    """The AMP step computes its weight gradients WITHOUT a row mask (the LDS-DMA kernel has none): that is only right
    because the rows of padded positions are exactly zero in every gradient tensor that enters a weight gradient.  Proof by
    invariance: whatever sits in the padded rows of the stack's input and of the incoming output gradient, the gradients of
    all parameters - and the valid rows of d x - are bit-for-bit the same."""
    tr = _decoder_stack(state_dict, 2).eval()              # (no dropout: the masks would differ between the runs' seeds)
    B, N = 3, 70
    lens = torch.tensor([70, 41, 64])
    mask = (torch.arange(N)[None, :] < lens[:, None]).to(DEV)
    x, dout = _rand((B, N, 384), 20).to(DEV), _rand((B, N, 384), 21).to(DEV)

    def run(garbage):
        xx, dd = x.clone(), dout.clone()
        xx[~mask] = garbage
        dd[~mask] = -3.0 * garbage
        xg = xx.requires_grad_()
        out = train.transformer_train_forward(tr, xg, mask, amp=True, key_len=lens.to(DEV))
        out.backward(dd)
        g = {n: p.grad.clone() for n, p in tr.named_parameters()}
        tr.zero_grad(set_to_none=True)
        return out.detach(), xg.grad, g
    o_a, dx_a, g_a = run(0.0)
    o_b, dx_b, g_b = run(7.5)
    assert torch.equal(o_a, o_b) and float(o_a[~mask].abs().max()) == 0.0
    assert torch.equal(dx_a[mask], dx_b[mask])
    for n in g_a:
        assert torch.equal(g_a[n], g_b[n]), n


def test_attention_binarization_loss():
    """AttentionBinarizationLoss (loss.py:80-107) on a real MAS output: value and d / d attn_soft against the reference's
    expression under autograd (float64), including cells clamped at eps (zero gradient) and ragged utterances."""
    B, M, L = 5, 200, 60
    x, tl, ml = synth.make_mas_logits(B, M, L, True, "realistic")
    hard = runtime.mas(x.to(DEV), tl.to(DEV), ml.to(DEV))[0]
    soft = torch.softmax(x, dim=-1)
    soft[0, :20] = 1e-9                                    # below eps: clamped, no gradient
    s64 = soft.double().requires_grad_()
    ref = 0.5 * torc.attention_binarization_loss(s64, hard.cpu().long(), 1e-6)
    ref.backward()
    sg = soft.to(DEV).requires_grad_()
    loss = train.AttentionBinarizationLoss(weight=0.5)(sg, hard)
    _close(loss, ref, 2e-6, "binarisation loss")
    loss.backward()
    _close(sg.grad, s64.grad, 2e-6, "d attn_soft")
    assert int(runtime.attn_bin_loss(soft.to(DEV), hard)[0][1]) == int(ml.sum())


@pytest.mark.parametrize("B,M,L,kind", [(5, 120, 40, "ragged"), (3, 64, 100, "impossible"), (2, 300, 129, "ragged")])
def test_attention_ctc_loss(B, M, L, kind):
    """AttentionCTCLoss (loss.py:39-77) against the reference's expression (pad + log_softmax + nn.CTCLoss, float64) under
    autograd: value and d / d attn_logits; ragged lengths, frames past mel_len (zero gradient), and an utterance with fewer
    frames than tokens (infinite loss -> 0 with zero gradient: zero_infinity)."""
    logits = _rand((B, M, L), 110, 2.0)
    text_len, mel_len = synth.make_lengths(B, L, M, variable=True, seed=111)
    text_len, mel_len = text_len.clamp(min=1), mel_len.clamp(min=1)
    if kind == "impossible":
        text_len[0], mel_len[0] = 90, 50            # cannot emit 90 tokens in 50 frames
        text_len[1], mel_len[1] = L, M
    l64 = logits.double().requires_grad_()
    ref = 0.7 * torc.attention_ctc_loss(l64, text_len, mel_len, -1)
    ref.backward()
    lg = logits.to(DEV).requires_grad_()
    loss = train.AttentionCTCLoss(blank_logprob=-1, weight=0.7)(lg, text_len.to(DEV), mel_len.to(DEV))
    _close(loss, ref, 2e-5, "CTC loss")
    loss.backward()
    # (fp32 log-domain recursions: alpha + beta + nll are O(10^3) in magnitude, so their fp32 rounding (1e-4 absolute) is
    # the relative error of the posterior that the gradient subtracts)
    _close(lg.grad, l64.grad, 2e-3, "d attn_logits")
    for b in range(B):
        assert float(lg.grad[b, int(mel_len[b]):].abs().max()) == 0.0 if int(mel_len[b]) < M else True
    if kind == "impossible":
        assert float(lg.grad[0].abs().max()) == 0.0


def test_acoustic_model_loss_on_a_forward(gpu_model, state_dict):
    """`train.AcousticModelLoss` (loss.py:122-182) on the outputs of a real forward: every term and the total against the
    reference's expressions evaluated on the ORACLE's outputs for the same batch (float32 CPU)."""
    inp = synth.make_inputs(3, 60, 200, variable=True, seed=7)
    args = (inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"], inp["energy"])
    ref = orc.acoustic_forward(state_dict, *args, inp["flow_x0"], inp["flow_t"])
    out = gpu_model(*[a.to(DEV) for a in args], flow_noise=inp["flow_x0"].to(DEV), flow_time=inp["flow_t"].to(DEV))
    crit = train.AcousticModelLoss()
    total, terms = crit({"mel": inp["mel"].to(DEV), "mel_len": inp["mel_len"].to(DEV), "text_len": inp["text_len"].to(DEV)}, out)
    want = {"model/mel_loss": torc.mel_loss(ref.mel, inp["mel"], inp["mel_len"]),
            "aligner/attention_loss": torc.attention_ctc_loss(ref.aligner.attn_logits.reshape(3, 200, 60), inp["text_len"], inp["mel_len"]),
            "aligner/kl_loss": torc.attention_binarization_loss(ref.aligner.attn_soft, ref.aligner.attn_hard)}
    for k, v in want.items():
        _close(terms[k], v, 1e-4, k)
    flow = [v for k, v in terms.items() if k.startswith("adaptor/")]
    assert len(flow) == 1
    _close(flow[0], ref.adaptor.flow_loss, 1e-4, "adaptor flow loss")
    _close(total, sum(want.values()) + ref.adaptor.flow_loss, 1e-4, "total loss")


def test_length_regulator_backward_and_batched_gemm_tn():
    """LengthRegulateFunction: out[b] = A[b] x[b] and d x[b] = A[b]^T d out[b] against float64 autograd (B = 5, ragged);
    the batched transposed GEMM on its own against torch.bmm; L not a multiple of the tiles (L = 100)."""
    B, M, L, D = 5, 200, 100, 384
    a = torch.softmax(_rand((B, M, L), 120, 2.0), dim=-1)
    mel_len = torch.tensor([200, 133, 64, 200, 97])
    a = a * (torch.arange(M)[None, :] < mel_len[:, None])[..., None]
    x = _rand((B, L, D), 121).requires_grad_()
    d_out = _rand((B, M, D), 122)
    ref = torch.bmm(a.double(), x.double())
    ref.backward(d_out.double())
    xg = x.detach().to(DEV).requires_grad_()
    out, dec_len, dec_mask = train.LengthRegulateFunction.apply(xg, a.to(DEV), mel_len.view(-1, 1).to(DEV), M)
    valid = (torch.arange(M)[None, :] < mel_len[:, None])[..., None]
    _close(out.cpu() * valid, ref.detach() * valid, 5e-6, "regulated output")
    assert torch.equal(dec_len.cpu(), mel_len)
    out.backward((d_out * valid).to(DEV))
    x.grad = None
    torch.bmm(a.double(), x.double()).backward((d_out * valid).double())
    _close(xg.grad, x.grad, 2e-5, "d x")
    p, q = _rand((7, 130, 100), 123), _rand((7, 130, 36), 124)
    _close(runtime.gemm_tn_batched(p.to(DEV), q.to(DEV)), torch.bmm(p.double().transpose(1, 2), q.double()), 2e-6, "batched A^T B")


def test_mel_loss_gradients_of_the_whole_chain_match_autograd(state_dict):
    """`train.acoustic_mel_train_forward` + MelLoss: the mel loss's gradient for EVERY parameter it reaches outside the aligner
    (text embedding table, TextEncoder, the adaptor's embedding module, MelDecoder, to_mel) against torch autograd over the
    oracle's forward, composed as the reference composes it (aligner fed the detached encoder output, model.py:139; the
    alignment itself a constant here).  B = 2, ragged."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    inp = synth.make_inputs(2, 40, 150, variable=True, seed=31)
    text, text_len, mel, mel_len, pitch, energy = (inp[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy"))
    keep = ("text_embedding.", "encoder.", "temporal_adaptor.embedding.", "decoder.", "to_mel.")
    sd = {k: (v.clone().requires_grad_() if k.startswith(keep) else v.clone()) for k, v in state_dict.items()}
    emb = F.embedding(text, sd["text_embedding.weight"], padding_idx=0)
    enc_mask = torch.arange(text.shape[1])[None, :] < text_len[:, None]
    enc_out = orc.transformer(sd, "encoder", emb, enc_mask)
    with torch.no_grad():
        al = orc.aligner(sd, mel, enc_out.detach().transpose(1, 2), mel_len, text_len)
    ad = orc.adaptor_forward(sd, enc_out, enc_mask, mel.shape[2], al.attn_hard_duration, al.attn_soft, pitch, energy,
                             inp["flow_x0"], inp["flow_t"])
    dec_mask = torch.arange(mel.shape[2])[None, :] < ad.dec_lengths[:, None]
    dec = orc.transformer(sd, "decoder", ad.enc_out, dec_mask)
    mel_ref = F.linear(dec, sd["to_mel.weight"], sd["to_mel.bias"]).transpose(1, 2) * dec_mask[:, None]
    loss_ref = torc.mel_loss(mel_ref, mel, mel_len)
    loss_ref.backward()

    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).eval()                      # eval: no dropout, so the two sides compute the same function
    d = {k: v.to(DEV) for k, v in inp.items()}
    out = train.acoustic_mel_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"])
    _close(out, mel_ref, 2e-4, "mel")
    loss = train.MelLoss()(out, d["mel"], d["mel_len"])
    _close(loss, loss_ref, 1e-5, "mel loss")
    loss.backward()
    checked = 0
    for name, p in model.named_parameters():
        if not name.startswith(keep):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        _close(p.grad, sd[name].grad, 1e-3, f"d {name}")
        checked += 1
    assert checked == sum(1 for k in state_dict if k.startswith(keep))          # 143 tensors
    assert float(model.text_embedding.weight.grad[0].abs().max()) == 0.0     # padding row


def test_flow_predictor_loss_gradients_match_autograd(state_dict):
    """`train.flow_predictor_loss` (time embedding -> AdaLN projections -> split input projection -> 3 x 256 adaptive-norm
    stack -> output Linear -> flow loss): value, the gradient of every predictor parameter and of the conditioning encoder
    output against torch autograd over the oracle's `predictor_forward`."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    B, L = 3, 50
    cond = _rand((B, L, 384), 130)
    targets = _rand((B, L, 3), 131)
    x0, t = _rand((B, L, 3), 132), torch.tensor([0.13, 0.71, 0.42])
    lens = torch.tensor([50, 31, 44])
    mask = torch.arange(L)[None, :] < lens[:, None]
    pre = "temporal_adaptor.predictor."
    sd = {k: (v.clone().requires_grad_() if k.startswith(pre) else v.clone()) for k, v in state_dict.items()}
    c64 = cond.clone().requires_grad_()
    _, loss_ref = orc.predictor_forward(sd, c64, targets, mask, x0, t)
    loss_ref.backward()

    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).eval()
    pred = model.temporal_adaptor.predictor
    cg = cond.to(DEV).requires_grad_()
    loss = train.flow_predictor_loss(pred, cg, targets.to(DEV), mask.to(DEV), x0.to(DEV), t.to(DEV))
    _close(loss, loss_ref, 2e-5, "flow loss")
    loss.backward()
    _close(cg.grad, c64.grad, 1e-3, "d cond")
    n = 0
    for name, p in pred.named_parameters():
        ref = sd[pre + name].grad
        assert p.grad is not None and ref is not None, name
        _close(p.grad, ref, 1e-3, f"d {name}")
        n += 1
    assert n == sum(1 for k in state_dict if k.startswith(pre) and not k.endswith("freq_scale"))      # 52 (freq_scale: a buffer)


def test_mel_plus_flow_loss_gradients_of_the_model_match_autograd(state_dict):
    """`train.acoustic_train_forward`: mel + flow loss; the gradient of EVERY parameter outside the aligner (195 tensors: text
    embedding, TextEncoder, embedding module, flow predictor, MelDecoder, to_mel) against torch autograd over the oracle's
    forward composed as the reference composes it (aligner on the detached encoder output, alignment constant); the CTC and
    binarisation terms against the reference's expressions as values."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    inp = synth.make_inputs(2, 40, 150, variable=True, seed=33)
    text, text_len, mel, mel_len, pitch, energy = (inp[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy"))
    frozen = ("aligner.",)
    sd = {k: (v.clone() if k.startswith(frozen) or not v.is_floating_point() or k.endswith("freq_scale") else v.clone().requires_grad_())
          for k, v in state_dict.items()}
    emb = F.embedding(text, sd["text_embedding.weight"], padding_idx=0)
    enc_mask = torch.arange(text.shape[1])[None, :] < text_len[:, None]
    enc_out = orc.transformer(sd, "encoder", emb, enc_mask)
    with torch.no_grad():
        al = orc.aligner(sd, mel, enc_out.detach().transpose(1, 2), mel_len, text_len)
    ad = orc.adaptor_forward(sd, enc_out, enc_mask, mel.shape[2], al.attn_hard_duration, al.attn_soft, pitch, energy,
                             inp["flow_x0"], inp["flow_t"])
    dec_mask = torch.arange(mel.shape[2])[None, :] < ad.dec_lengths[:, None]
    dec = orc.transformer(sd, "decoder", ad.enc_out, dec_mask)
    mel_ref = F.linear(dec, sd["to_mel.weight"], sd["to_mel.bias"]).transpose(1, 2) * dec_mask[:, None]
    total_ref = torc.mel_loss(mel_ref, mel, mel_len) + ad.flow_loss
    total_ref.backward()

    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).eval()
    d = {k: v.to(DEV) for k, v in inp.items()}
    mel_out, total, losses = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"],
                                                          d["energy"], flow_noise=d["flow_x0"], flow_time=d["flow_t"],
                                                          train_aligner=False)
    _close(total, total_ref, 2e-5, "mel + flow loss")
    _close(losses["adaptor/flow_loss"], ad.flow_loss, 2e-5, "flow loss")
    _close(losses["aligner/attention_loss"], torc.attention_ctc_loss(al.attn_logits.reshape(2, 150, 40), text_len, mel_len), 1e-4, "CTC value")
    _close(losses["aligner/kl_loss"], torc.attention_binarization_loss(al.attn_soft, al.attn_hard), 1e-4, "binarisation value")
    total.backward()
    checked = 0
    for name, p in model.named_parameters():
        if name.startswith(frozen):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        _close(p.grad, sd[name].grad, 1e-3, f"d {name}")
        checked += 1
    assert checked == 195, checked


def test_full_training_loss_gradients_match_autograd(state_dict):
    """`train.acoustic_train_forward(train_aligner=True)`: the reference's TOTAL loss (mel + flow + CTC + binarisation,
    loss.py:140-182) and its gradient for EVERY parameter of the model (206 tensors, the aligner's 11 included) against torch
    autograd over the oracle's forward composed as the reference composes it (`train_oracle.acoustic_losses`: aligner on the
    detached encoder output, model.py:139; flow targets and the embedding stack's pitch / energy inputs detached,
    temporal_adaptor.py:112, :284, :292 - the composition tests/golden/train.npz pins against the reference itself)."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    from oracle import mas_oracle
    inp = synth.make_inputs(2, 40, 150, variable=True, seed=35)
    text, text_len, mel, mel_len, pitch, energy = (inp[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy"))
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith("freq_scale") else v.clone())
          for k, v in state_dict.items()}
    total_ref, terms = torc.acoustic_losses(sd, text, text_len, mel, mel_len, pitch, energy, inp["flow_x0"], inp["flow_t"])
    total_ref.backward()

    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).eval()
    d = {k: v.to(DEV) for k, v in inp.items()}
    mel_out, total, losses = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"],
                                                          d["energy"], flow_noise=d["flow_x0"], flow_time=d["flow_t"],
                                                          train_aligner=True)
    for k, v in terms.items():
        _close(losses[k], v, 1e-4, k)
    _close(total, total_ref, 1e-4, "total loss")
    total.backward()
    checked = 0
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        _close(p.grad, sd[name].grad, 2e-3, f"d {name}")
        checked += 1
    assert checked == 206, checked


def test_conv_attention_backward_matches_autograd(state_dict):
    """`train.aligner.ConvAttentionFunction`: (attn_soft, attn_logits) and the gradients of the aligner's 11 parameters under
    random cotangents for BOTH outputs, against torch autograd over the oracle's ConvAttention (alignment.py:159-208).
    Ragged lengths; M, L not multiples of 4 (the padded columns of d S)."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    from isp_tts_amd.train import aligner as tal
    B, L, M = 3, 37, 141
    mel_len, text_len = torch.tensor([141, 77, 120]), torch.tensor([37, 21, 33])
    mel, keys = _rand((B, 80, M), 41), _rand((B, 384, L), 42)
    g_soft, g_logits = _rand((B, M, L), 43), _rand((B, M, L), 44)
    sd = {k: (v.clone().requires_grad_() if k.startswith("aligner.") and v.is_floating_point() else v) for k, v in state_dict.items()}
    soft, logits = orc.conv_attention(sd, mel, keys, mel_len, text_len)
    ((soft * g_soft).sum() + (logits * g_logits).sum()).backward()
    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).eval()
    s, lg = tal.conv_attention_train(model.aligner.attention, mel.to(DEV), keys.to(DEV), mel_len.to(DEV), text_len.to(DEV))
    _close(s, soft, 1e-4, "attn_soft")
    _close(lg, logits, 1e-4, "attn_logits")
    ((s * g_soft.to(DEV)).sum() + (lg * g_logits.to(DEV)).sum()).backward()
    checked = 0
    for name, p in model.aligner.attention.named_parameters():
        _close(p.grad, sd["aligner.attention." + name].grad, 1e-3, f"d {name}")
        checked += 1
    assert checked == 11, checked
    # the step under autocast: bf16 convolution operands forward and backward (fp32 outputs, norms and scores).  What remains
    # against the fp32 oracle is bf16 rounding of the operands: soft probabilities to 3e-2 of their scale, gradients to 5e-2
    for p in model.aligner.attention.parameters():
        p.grad = None
    s16, lg16 = tal.conv_attention_train(model.aligner.attention, mel.to(DEV), keys.to(DEV), mel_len.to(DEV), text_len.to(DEV), amp=True)
    _close(s16, soft, 3e-2, "attn_soft (bf16 convolutions)")
    ((s16 * g_soft.to(DEV)).sum() + (lg16 * g_logits.to(DEV)).sum()).backward()
    for name, p in model.aligner.attention.named_parameters():
        _close(p.grad, sd["aligner.attention." + name].grad, 5e-2, f"d {name} (bf16 convolutions)")


def test_soft_average_and_length_regulator_alignment_gradients():
    """The two adaptor steps through which the mel loss reaches attn_soft: `SoftAverageFunction` (TemporalAverager,
    temporal_adaptor.py:446-449) and the alignment operand of `LengthRegulateFunction` (:419-421), against float64 autograd."""
    from isp_tts_amd.train import aligner as tal
    B, M, L, D = 3, 150, 40, 384
    mel_len, text_len = torch.tensor([150, 77, 120]), torch.tensor([40, 21, 33])
    mm, tm = torch.arange(M)[None, :] < mel_len[:, None], torch.arange(L)[None, :] < text_len[:, None]
    A = torch.softmax(_rand((B, M, L), 51, 2.0), -1) * mm[..., None] * tm[:, None, :]
    pitch, energy, gf = _rand((B, M), 52), _rand((B, M), 53), _rand((B, L, 3), 54)
    A64 = A.double().requires_grad_()
    avg = lambda v: (orc.soft_average(v.double()[:, None], A64)).transpose(1, 2) * tm[..., None]   # noqa: E731
    pt, et = avg(pitch), avg(energy)
    (torch.cat([torch.zeros_like(pt), pt, et], -1) * gf.double()).sum().backward()
    Ag = A.to(DEV).requires_grad_()
    f = tal.SoftAverageFunction.apply(Ag, pitch.to(DEV), energy.to(DEV), text_len.to(DEV))
    _close(f[..., 1:], torch.cat([pt, et], -1), 1e-5, "soft averages")
    (f * gf.to(DEV)).sum().backward()
    _close(Ag.grad, A64.grad, 1e-5, "d attn_soft (soft averages)")
    x, dout = _rand((B, L, D), 55), _rand((B, M, D), 56) * mm[..., None]
    A64, x64 = A.double().requires_grad_(), x.double().requires_grad_()
    (torch.bmm(A64, x64) * dout.double()).sum().backward()
    Ag, xg = A.to(DEV).requires_grad_(), x.to(DEV).requires_grad_()
    out, _, _ = train.LengthRegulateFunction.apply(xg, Ag, mel_len.view(-1, 1).to(DEV), M)
    (out * dout.to(DEV)).sum().backward()
    _close(Ag.grad, A64.grad, 1e-5, "d attn_soft (length regulator)")
    _close(xg.grad, x64.grad, 1e-5, "d x (length regulator)")


def test_projected_stack_passes_a_gradient_to_its_input(state_dict):
    """The adaptor's embedding stack (2 -> 256 projection in front, transformer.py:170) as a differentiable node of a
    NON-contiguous input slice (how the soft averages enter it): d input against autograd over the oracle."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    B, L = 2, 40
    text_len = torch.tensor([40, 23])
    mask = torch.arange(L)[None, :] < text_len[:, None]
    feats = _rand((B, L, 3), 61).requires_grad_()
    sd = {k[len("temporal_adaptor.embedding."):]: v for k, v in state_dict.items() if k.startswith("temporal_adaptor.embedding.")}
    want = orc.transformer(sd, "transformer", feats[..., 1:3], mask)
    gy = _rand(tuple(want.shape), 62)
    (want * gy).sum().backward()
    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).eval()
    fg = feats.detach().to(DEV).requires_grad_()
    got = train.transformer_train_forward(model.temporal_adaptor.embedding.transformer, fg[..., 1:3], mask.to(DEV))
    _close(got, want, 1e-4, "projected stack forward")
    (got * gy.to(DEV)).sum().backward()
    _close(fg.grad, feats.grad, 1e-3, "d features")
    assert float(fg.grad[..., 0].abs().max()) == 0.0


@pytest.mark.parametrize("rows,dim,vocab", [(9001, 384, 50), (6400, 384, 149), (70, 256, 10)])
def test_embedding_backward_over_segments(rows, dim, vocab):
    """d table = index_add of the token gradients (nn.Embedding backward, model.py:131) with more tokens than one segment of
    the kernel's id scan, a ragged tail, an unused vocabulary row and the padding row; fixed order -> identical bits twice."""
    g = torch.Generator().manual_seed(71)
    ids = torch.randint(0, vocab - 1, (rows,), generator=g)          # row vocab - 1 never occurs
    d_emb = _rand((rows, dim), 72)
    want = torch.zeros(vocab, dim, dtype=torch.float64).index_add_(0, ids, d_emb.double())
    want[0] = 0.0
    got = runtime.embedding_bwd(ids.to(DEV), d_emb.to(DEV), vocab, padding_idx=0)
    _close(got, want, 2e-6, "d table")
    assert float(got[0].abs().max()) == 0.0 and float(got[vocab - 1].abs().max()) == 0.0
    assert torch.equal(got, runtime.embedding_bwd(ids.to(DEV), d_emb.to(DEV), vocab, padding_idx=0))


@pytest.mark.parametrize("M,N1,N2,masked", [(1000, 80, 384, False), (4099, 384, 1536, True), (64, 512, 384, True),
                                             (32768, 1536, 384, False)])
def test_gemm_tn_bf16_operands(M, N1, N2, masked):
    """ispk_gemm_tn_bf16 (autocast's weight gradient): the fp32 operands rounded to bf16 in flight, fp32 accumulation -
    against float64 over the SAME bf16-rounded operands (2e-5 of the result's scale: accumulation order only), and within
    bf16 rounding of the fp32 product.  Row tails, column tails, row mask, accumulation; identical bits run to run."""
    a, b = _rand((M, N1), 81), _rand((M, N2), 82)
    mask = (torch.arange(M) % 7 != 3) if masked else None
    a16, b16 = a.bfloat16().double(), b.bfloat16().double()
    want = ((a16 * mask[:, None]) if masked else a16).T @ b16
    ad, bd, md = a.to(DEV), b.to(DEV), (mask.to(DEV) if masked else None)
    got = runtime.gemm_tn(ad, bd, row_mask=md, bf16=True)
    _close(got, want, 2e-5, "gemm_tn bf16")
    _close(got, ((a.double() * mask[:, None]) if masked else a.double()).T @ b.double(), 1e-2, "gemm_tn bf16 vs the fp32 product")
    assert torch.equal(got, runtime.gemm_tn(ad, bd, row_mask=md, bf16=True))
    acc = got.clone()
    runtime.gemm_tn(ad, bd, row_mask=md, out=acc, accumulate=True, bf16=True)
    _close(acc, 2 * want, 2e-5, "gemm_tn bf16 accumulate")


@pytest.mark.parametrize("B,N,H,lens,p", [(2, 100, 6, [100, 73], 0.1), (2, 140, 4, None, 0.0), (1, 37, 5, [20], 0.1),
                                          (9, 512, 6, [512, 300, 1, 64, 65, 449, 512, 33, 480], 0.1), (2, 600, 6, [600, 515], 0.1),
                                          (1, 530, 2, None, 0.0)])
def test_attention_training_pair_on_bf16_tensors(B, N, H, lens, p):
    """ispk_alibi_mqa_attn_train_bf16 / _bwd_bf16 (the step under autocast; csrc/attention_train.hip): bf16 q / k / v / dO in,
    bf16 o / dqkv out, products on bf16 MFMAs off LDS-staged tiles.  Checked against the fp32 pair run on the SAME bf16-rounded
    q / k / v / dO (what remains is the rounding of P, dS and the outputs to bf16 and the accumulation order: 1e-2 of each
    tensor's scale), with the same dropout mask (same seed).  Cases: ragged key lengths incl. 1 and tile edges, more than one
    XCD group of batch items, N > 512 (two staging rounds of the forward / dQ kernels), H = 2 .. 6."""
    qkv16 = _rand((B, N, H * 64 + 128), 91).bfloat16().to(DEV)
    do16 = _rand((B, N, H * 64), 92).bfloat16().to(DEV)
    qkv, d_o = qkv16.float(), do16.float()
    slopes = (torch.tensor(synth.alibi_default_slopes(H)) * 1.1).to(DEV)
    key_len = None if lens is None else torch.tensor(lens, device=DEV)
    o32, lse32 = runtime.alibi_mqa_attention_train(qkv, H, slopes, key_len, p, 1234)
    o16, lse16 = runtime.alibi_mqa_attention_train(qkv16, H, slopes, key_len, p, 1234)
    assert o16.dtype == torch.bfloat16
    _close(o16.float(), o32, 1e-2, "o")
    _close(lse16, lse32, 1e-5, "lse")
    # the backward's `o` is an INPUT (delta = rowsum(o dO)): both pairs get the bf16 tensor autocast's SDPA hands over - its
    # rounding moves delta, and through the distance-weighted sum the slope gradient, by more than the kernels differ
    dq32, ds32 = runtime.alibi_mqa_attention_bwd(qkv, o16.float(), d_o, H, slopes, key_len, lse=lse32, dropout_p=p, seed=1234)
    dq16, ds16 = runtime.alibi_mqa_attention_bwd(qkv16, o16, do16, H, slopes, key_len, lse=lse16, dropout_p=p, seed=1234)
    assert dq16.dtype == torch.bfloat16
    hq = H * 64
    _close(dq16[..., :hq].float(), dq32[..., :hq], 1e-2, "dq")
    _close(dq16[..., hq:hq + 64].float(), dq32[..., hq:hq + 64], 1e-2, "dk")
    _close(dq16[..., hq + 64:].float(), dq32[..., hq + 64:], 1e-2, "dv")
    _close(ds16, ds32, 2e-2, "d log-slope")
    again, ds_again = runtime.alibi_mqa_attention_bwd(qkv16, o16, do16, H, slopes, key_len, lse=lse16, dropout_p=p, seed=1234)
    assert torch.equal(again, dq16) and torch.equal(ds_again, ds16)          # fixed summation orders: bit-reproducible


def test_training_step_issues_no_aten_compute_ops(state_dict):
    """One AMP training step - forward with dropout, the four losses, backward of every parameter, clip + AdamW, zeroing of the
    gradient arena, re-staging of the updated weights - is libispk launches only: a TorchDispatchMode sees views and
    allocations on the device and nothing else (gradient delivery, weight concatenation / casts, zero fills and the scalar
    algebra on the losses are kernels of csrc/util.hip; the two tensors with two consumers are forked so that the autograd
    engine adds nothing itself).  Ops on host tensors only (the dropout seeds come from torch's CPU generator) are not GPU work
    and are ignored."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils._pytree import tree_flatten
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    harmless = ("aten.view", "aten.empty", "aten._unsafe_view", "aten.transpose", "aten.slice", "aten.select",
                "aten.unsqueeze", "aten.expand", "aten.detach", "aten.alias", "aten.t.", "aten.permute", "aten.squeeze",
                "aten.reshape", "aten.as_strided", "aten.is_", "aten.size", "aten.stride", "aten.lift_fresh",
                "aten._reshape_alias", "aten.split", "aten.unbind", "aten.sym_", "aten.empty_like", "aten.new_empty",
                "aten.record_stream", "aten.view_as")
    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            if not name.startswith(harmless):
                tensors = [t for t in tree_flatten((args, kwargs, out))[0] if isinstance(t, torch.Tensor)]
                if any(t.is_cuda for t in tensors):
                    seen.append(name)
            return out

    torch.manual_seed(3)
    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).train()
    d = {k: v.to(DEV) for k, v in synth.make_inputs(4, 60, 200, variable=True, seed=5).items()}
    opt = train.FlatAdamW(list(model.parameters()), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
    opt.check_finite = False           # (the finiteness check of the reference's step is one host read of the norm)

    def step():
        _, total, losses = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                                        flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
        return total, opt.step(total)

    for _ in range(2):
        first, _ = step()
    torch.cuda.synchronize()
    with Spy():
        total, norm = step()
    torch.cuda.synchronize()
    assert seen == [], f"PyTorch kernels inside a training step: {sorted(set(seen))}"
    assert torch.isfinite(total) and torch.isfinite(norm) and float(norm) > 0


def test_data_movement_kernels_of_the_training_step():
    """csrc/util.hip against their torch expressions, bit for bit: segments (copy / add / bf16 cast, more than 32 per call),
    the weight-image pass (plain, transposed, bf16, exp; column-offset destinations), zero fill of odd byte counts, scale by a
    device scalar, ordered scalar sum, exp with zero padding, sqrt, strided copy, the two convolution-weight permutations."""
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randn(n, generator=g).to(DEV) for n in (1, 7, 384, 5000, 147456) * 8]           # 40 segments
    base = [torch.randn(t.numel(), generator=g).to(DEV) for t in srcs]
    dst = [b.clone() for b in base]
    d16 = [torch.empty(t.numel(), dtype=torch.bfloat16, device=DEV) for t in srcs]
    runtime.segments([(t, d, runtime.SEG_ADD if k % 2 else runtime.SEG_COPY) for k, (t, d) in enumerate(zip(srcs, dst))])
    runtime.segments([(t, d, runtime.SEG_BF16) for t, d in zip(srcs, d16)])
    for k, (t, b, d, h) in enumerate(zip(srcs, base, dst, d16)):
        assert torch.equal(d, b + t if k % 2 else t) and torch.equal(h, t.to(torch.bfloat16))
    wq, wkv, w1 = torch.randn(384, 384, generator=g).to(DEV), torch.randn(128, 384, generator=g).to(DEV), torch.randn(1536, 384, generator=g).to(DEV)
    logs = torch.randn(1, 6, generator=g).to(DEV)
    cat16 = torch.empty(512, 384, dtype=torch.bfloat16, device=DEV)
    cat_t = torch.empty(384, 512, dtype=torch.float32, device=DEV)
    w1_t16 = torch.empty(384, 1536, dtype=torch.bfloat16, device=DEV)
    sl = torch.empty(1, 6, device=DEV)
    runtime.stage_weights([(wq, cat16[:384], False, False), (wkv, cat16[384:], False, False), (wq, cat_t[:, :384], True, False),
                           (wkv, cat_t[:, 384:], True, False), (w1, w1_t16, True, False), (logs, sl, False, True)])
    assert torch.equal(cat16, torch.cat([wq, wkv]).to(torch.bfloat16)) and torch.equal(cat_t, torch.cat([wq, wkv]).t())
    assert torch.equal(w1_t16, w1.t().to(torch.bfloat16))
    assert (sl - logs.exp()).abs().max() < 1e-6 * float(logs.exp().max())
    assert torch.equal(runtime.cat0([wq, wkv]), torch.cat([wq, wkv])) and torch.equal(runtime.cat0([wq, wkv], torch.bfloat16), cat16)
    z = torch.full((1003,), 7, dtype=torch.int16, device=DEV)[:1001]          # 2,002 bytes: a 2-byte tail
    runtime.zero_(z)
    assert int(z.abs().sum()) == 0
    x = srcs[3].clone()
    sc = torch.tensor([0.37], device=DEV)
    assert torch.equal(runtime.scale_(x, sc, 3.0), srcs[3] * (sc * 3.0))
    terms = [torch.randn((), generator=g).to(DEV) for _ in range(4)]
    want = ((terms[0] * 1.0 + terms[1] * 2.0) + terms[2] * 0.5) + terms[3] * 1.0
    assert torch.equal(runtime.sum_scalars(terms, [1.0, 2.0, 0.5, 1.0]), want)
    ep = runtime.exp_pad(logs, 8)
    assert ep.shape == (8,) and float(ep[6:].abs().sum()) == 0 and (ep[:6] - logs.exp().reshape(-1)).abs().max() < 1e-6 * float(ep.max())
    sq = torch.tensor([9.0, 2.0], device=DEV)
    assert torch.equal(runtime.sqrt_scale(sq, 0.5), sq.sqrt() * 0.5)
    wide = torch.randn(50, 387, generator=g).to(DEV)
    part = torch.zeros(50, 384, device=DEV)
    assert torch.equal(runtime.copy2d(wide[:, 3:], part), wide[:, 3:])
    cw = torch.randn(160, 80, 5, generator=g).to(DEV)
    assert torch.equal(runtime.permute021(cw), cw.permute(0, 2, 1).contiguous())
    assert torch.equal(runtime.conv_weight_flip(cw), cw.flip(2).permute(1, 2, 0).reshape(80, 5 * 160))
    a3, b3 = torch.randn(5, 70, 384, generator=g).to(DEV), torch.randn(5, 33, 384, generator=g).to(DEV)
    _close(runtime.gemm_batched(a3, b3), torch.einsum("bmk,bnk->bmn", a3.double().cpu(), b3.double().cpu()), 2e-6, "batched NT GEMM")


@pytest.mark.parametrize("M,N,K,masked,p", [(1000, 1536, 384, True, 0.1), (6400, 1024, 256, False, 0.3), (333, 1536, 384, True, 0.0)])
def test_feed_forward_gemms_with_the_gelu_pair_in_their_epilogues(M, N, K, masked, p):
    """ispk_gemm_bf16_gelu_train / ispk_gemm_bf16_gelu_bwd: the AMP step's first feed-forward Linear with u AND
    a = dropout(gelu(u)) from one launch, and du = (m dy W2) gelu'(u) [dropout] from the backward GEMM's epilogue - bit for bit
    what the GEMM followed by the GELU kernel gives (same expressions on the same rounded values, same mask)."""
    x = _rand((M, K), 61).bfloat16().to(DEV)
    w1 = _rand((N, K), 62, K ** -0.5).bfloat16().to(DEV)
    dy = _rand((M, K), 63).bfloat16().to(DEV)
    w2t = _rand((N, K), 64, N ** -0.5).bfloat16().to(DEV)
    mask = (torch.arange(M) % 7 != 3).to(DEV) if masked else None
    u_ref = runtime.gemm(x, w1, out_dtype=torch.bfloat16)
    a_ref = runtime.gelu(u_ref, p, 99, out_dtype=torch.bfloat16)
    u, a = runtime.gemm_gelu_train(x, w1, p, 99)
    assert torch.equal(u, u_ref) and torch.equal(a, a_ref)
    da = runtime.gemm(dy, w2t, out_dtype=torch.bfloat16, mask=mask, flags=runtime.EP_MASK_OUT if masked else 0)
    du_ref = runtime.gelu_bwd(da, u_ref, dropout_p=p, seed=99)
    du = runtime.gemm_gelu_bwd(dy, w2t, u_ref, mask, p, 99)
    assert torch.equal(du, du_ref)
    if p > 0:
        assert abs(float((a == 0).float().mean()) - p) < 0.02


def test_dropout_seed_source_changes_the_masks_at_run_time():
    """ispk_set_dropout_seed_source: the same launch (same seed argument) draws another mask when the device word changes, the
    same mask when it does not, and forward / backward stay consistent - what a captured training step relies on."""
    u = _rand((1000, 256), 31).to(DEV)
    word = torch.zeros(1, dtype=torch.int64, device=DEV)
    base = runtime.gelu(u, 0.3, 77)
    try:
        runtime.set_seed_source(word)
        word.fill_(1)
        a1, a1b = runtime.gelu(u, 0.3, 77), runtime.gelu(u, 0.3, 77)
        da = torch.ones_like(u)
        d1 = runtime.gelu_bwd(da, u, dropout_p=0.3, seed=77)
        word.fill_(2)
        a2 = runtime.gelu(u, 0.3, 77)
        qkv = _rand((2, 70, 6 * 64 + 128), 32).bfloat16().to(DEV)
        slopes = torch.tensor(synth.alibi_default_slopes(6)).to(DEV)
        o2, _ = runtime.alibi_mqa_attention_train(qkv, 6, slopes, None, 0.2, 5)
        word.fill_(3)
        o3, _ = runtime.alibi_mqa_attention_train(qkv, 6, slopes, None, 0.2, 5)
    finally:
        runtime.set_seed_source(None)
    again = runtime.gelu(u, 0.3, 77)
    assert torch.equal(a1, a1b) and torch.equal(again, base)
    assert not torch.equal(a1, a2) and not torch.equal(a1, base) and not torch.equal(o2, o3)
    assert torch.equal(d1 == 0, a1 == 0) or ((d1 == 0) ^ (a1 == 0)).float().mean() < 1e-3     # same mask forward and backward
    keep = (a2 != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.02


def test_graphed_training_step_replays_match_eager_steps(state_dict):
    """train.GraphedTrainStep: the whole step (forward, four losses, backward, clip + AdamW, arena zeroing, weight re-staging)
    as ONE HIP graph.  With dropout off (eval-mode modules) replays must give bit-for-bit the parameters of the same number of
    eager steps: the graph re-stages its weight images from the arena the optimizer updates in place and reads the AdamW
    factors of each step from the device record, so nothing captured goes stale.  In train mode the masks change from replay
    to replay (the seed word), so two replays on the same batch differ."""
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims

    def make(train_mode):
        torch.manual_seed(11)
        m = AcousticModel.init(AcousticDims().model_config())
        m.load_state_dict(state_dict, strict=True)
        m = m.to(DEV)
        m = m.train() if train_mode else m.eval()
        o = train.FlatAdamW(list(m.parameters()), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
        o.check_finite = False
        return m, o
    d = {k: v.to(DEV) for k, v in synth.make_inputs(3, 52, 160, variable=True, seed=9).items()}
    keys = ("text", "text_len", "mel", "mel_len", "pitch", "energy", "flow_x0", "flow_t")
    batch = {k: d[k] for k in keys}
    m_e, o_e = make(False)
    eager_tot = []
    for _ in range(5):
        _, total, _ = train.acoustic_train_forward(m_e, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                                   flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
        o_e.step(total)
        eager_tot.append(float(total.detach()))
    m_g, o_g = make(False)
    step = train.GraphedTrainStep(m_g, o_g, batch, amp=True, warmup=2)        # two real steps ...
    graph_tot = []
    for _ in range(3):                                                         # ... and three replays = five steps
        total, losses, norm = step(**batch)
        graph_tot.append(float(total.detach()))
    torch.cuda.synchronize()
    assert o_g.step_count == 5 and o_e.step_count == 5
    assert graph_tot == eager_tot[2:], f"losses: graph {graph_tot} vs eager {eager_tot[2:]}"
    assert torch.equal(o_g.flat.data, o_e.flat.data) and torch.equal(o_g.exp_avg_sq, o_e.exp_avg_sq)
    assert float(o_g.flat.grad.abs().max()) == 0.0 and torch.isfinite(norm)
    # train mode: fresh masks per replay
    m_t, o_t = make(True)
    o_t.lr = 0.0                                       # (weights fixed: the only thing that changes between replays is the masks)
    step_t = train.GraphedTrainStep(m_t, o_t, batch, amp=True, warmup=2)
    t1 = float(step_t(**batch)[0].detach())
    t2 = float(step_t(**batch)[0].detach())
    assert t1 != t2 and abs(t1 - t2) < 0.2 * abs(t1)


def test_training_step_against_the_reference_fixture(state_dict):
    """SURVEY row f2 against the REAL reference: `train.acoustic_train_forward` (HIP forward + backward of the whole model,
    train_aligner=True, no dropout) and one `FlatAdamW` step (clip of the decay group + fused AdamW) on the B=2 golden inputs,
    compared with tests/golden/train.npz - what the reference's own AcousticModel.forward, AcousticModelLoss, .backward(),
    group_weight_decayable_params and a clipped torch.optim.AdamW step produce on them (oracle/make_goldens.py `gen_train`):
    the four losses, all 206 gradients (norm within 2e-3, strided sample within 2e-3 of the tensor's scale), the clip norm,
    the decay grouping and the parameter update."""
    import numpy as np
    from conftest import crc, golden
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    g = golden("train.npz")
    inp = synth.make_inputs(2, 100, 512)
    text_len, mel_len = torch.tensor(g["text_len"]), torch.tensor(g["mel_len"])
    tm = torch.arange(100)[None] < text_len[:, None]
    mm = torch.arange(512)[None] < mel_len[:, None]
    text, mel = inp["text"] * tm, inp["mel"] * mm[:, None]
    pitch, energy = inp["pitch"] * mm, inp["energy"] * mm
    assert [crc(text), crc(mel), crc(pitch), crc(energy)] == [int(v) for v in g["inputs_crc"]]
    model = AcousticModel.init(AcousticDims().model_config())
    model.load_state_dict(state_dict, strict=True)
    model = model.to(DEV).eval()
    names = [str(n) for n in g["names"]]
    params = dict(model.named_parameters())
    assert list(params) == names
    opt = train.FlatAdamW([params[n] for n in names], lr=float(g["lr"]), weight_decay=float(g["weight_decay"]),
                          grad_clip=float(g["grad_clip"]))
    # the decay grouping of experiments/optimizers.py:15-20 as the reference computed it
    decay = {id(p) for p in opt.flat.params[:opt.flat.n_decay_tensors]}
    assert [id(params[n]) in decay for n in names] == [bool(v) for v in g["in_decay_group"]]
    before = {n: params[n].detach().clone() for n in names}
    _, total, losses = train.acoustic_train_forward(model, text.to(DEV), text_len.to(DEV), mel.to(DEV), mel_len.to(DEV),
                                                    pitch.to(DEV), energy.to(DEV), flow_noise=inp["flow_x0"].to(DEV),
                                                    flow_time=inp["flow_t"].to(DEV), train_aligner=True)
    for k, v in losses.items():
        ref = float(g["loss_" + k.replace("/", "_")])
        assert abs(v.item() - ref) < 2e-4 * max(abs(ref), 1.0), (k, v.item(), ref)
    assert abs(total.item() - float(g["loss_total"])) < 2e-4 * float(g["loss_total"])
    total.backward()

    def sample(t, n=192):
        f = t.detach().reshape(-1)
        return f[::max(1, -(-f.numel() // n))].cpu()
    worst = 0.0
    for i, n in enumerate(names):
        gr, scale = params[n].grad, float(g["grad_absmax"][i])
        assert gr is not None, n
        assert abs(gr.double().norm().item() - float(g["grad_norm"][i])) <= 2e-3 * float(g["grad_norm"][i]) + 1e-7, n
        err = (sample(gr) - torch.from_numpy(g[f"g{i}"])).abs().max().item() / max(scale, 1e-12)
        worst = max(worst, err)
        assert err <= 2e-3, (n, err)
    print(f"HIP backward vs the reference's gradients: worst sampled error = {worst:.2e} of the tensor's scale")
    norm = opt.step(None)          # gradients are already in place
    torch.cuda.synchronize()
    if norm is not None:
        assert abs(float(norm) - float(g["grad_norm_group0"])) < 2e-3 * float(g["grad_norm_group0"])
    for i, n in enumerate(names):
        upd = params[n].detach() - before[n].to(params[n].device)
        assert (sample(upd) - torch.from_numpy(g[f"u{i}"])).abs().max().item() <= 2e-2 * float(g["lr"]), n


@pytest.mark.parametrize("M,N1,N2,masked", [(1000, 80, 384, False), (4099, 384, 1536, True), (64, 512, 384, True)])
def test_gemm_tn_with_operands_stored_in_bf16(M, N1, N2, masked):
    """`ispk_gemm_tn_b16` (the AMP step's weight gradient over activations kept in bf16): bit-equal to `ispk_gemm_tn_bf16` on
    the fp32 images of the same bf16 values - same products, same summation order - and within bf16 grade of float64."""
    a = _rand((M, N1), 301).to(torch.bfloat16)
    b = _rand((M, N2), 302).to(torch.bfloat16)
    mask = (torch.arange(M) % 5 != 2) if masked else None
    md = None if mask is None else mask.to(DEV)
    got = runtime.gemm_tn(a.to(DEV), b.to(DEV), row_mask=md)
    same = runtime.gemm_tn(a.float().to(DEV), b.float().to(DEV), row_mask=md, bf16=True)
    assert torch.equal(got, same)
    am = a.double() if mask is None else a.double() * mask[:, None]
    ref = am.t() @ b.double()
    assert (got.double().cpu() - ref).abs().max().item() < 1e-5 * max(ref.abs().max().item(), 1.0) * (M ** 0.5)


def test_gelu_bf16_forms_match_the_fp32_forms():
    """The bf16 forms of the GELU pair (bf16 a / da / du, and the _b16 pair with a bf16 pre-activation too) against the fp32
    forms on the same values: one bf16 rounding apart (they evaluate erf by the 3e-7 polynomial of common.h and exp on
    v_exp_f32, the fp32 forms by libm), with the same dropout mask."""
    u = _rand((300, 1536), 311, 2.0).to(DEV)
    da = _rand((300, 1536), 312).to(DEV)

    def one_ulp(got16, want32, what):
        err = (got16.float() - want32).abs()
        bound = want32.abs() * 2.0 ** -8 + 1e-6
        assert bool((err <= bound).all()), f"{what}: {float((err - bound).max()):.3e} over one bf16 rounding"
        assert torch.equal(got16 == 0, want32.to(torch.bfloat16) == 0) or ((got16 == 0) ^ (want32 == 0)).float().mean() < 1e-3   # (far negative inputs: 0 vs a denormal-sized value)
    for p, seed in ((0.0, 0), (0.1, 1234)):
        a32 = runtime.gelu(u, p, seed)
        one_ulp(runtime.gelu(u, p, seed, out_dtype=torch.bfloat16), a32, "gelu -> bf16")
        da16 = da.to(torch.bfloat16)
        du32 = runtime.gelu_bwd(da16.float(), u, dropout_p=p, seed=seed)
        one_ulp(runtime.gelu_bwd(da16, u, dropout_p=p, seed=seed), du32, "gelu backward, bf16 da / du")
        u16 = u.to(torch.bfloat16)                       # the pair with the pre-activation stored in bf16
        one_ulp(runtime.gelu(u16, p, seed, out_dtype=torch.bfloat16), runtime.gelu(u16.float(), p, seed), "gelu on a bf16 pre-activation")
        one_ulp(runtime.gelu_bwd(da16, u16, dropout_p=p, seed=seed), runtime.gelu_bwd(da16.float(), u16.float(), dropout_p=p, seed=seed),
                "gelu backward on a bf16 pre-activation")
