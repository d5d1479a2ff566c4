"""GPU parity of the split-fp16 kernels (csrc/split.hip: the parity-grade fast path) against float64 and the oracle.

Bars: a split product keeps 22 significant bits per operand, so every kernel here is held to fp32-grade bounds (the same
ones the exact-fp32 kernels meet), not to bf16 ones."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from isp_tts_amd import runtime, synth  # noqa: E402
from oracle import acoustic_oracle as orc  # noqa: E402

DEV = "cuda"


def _join(p):
    """split planes [2, ...] -> float64 value hi + lo"""
    return p[0].double().cpu() + p[1].double().cpu()


def test_split_planes_carry_22_bits():
    x = synth._normal("split/x", (77, 384), 3.0)
    x[0, :8] = torch.tensor([0.0, 1e-7, -1e-7, 65504.0, -65504.0, 1e6, -1e6, 6.0e-5])
    p = runtime.split_f16(x.to(DEV))
    torch.cuda.synchronize()
    ref = x.double().clamp(-65504.0, 65504.0)
    err = (_join(p) - ref).abs()
    # relative 2^-21 of the value, or fp16's subnormal spacing (2^-24) in absolute terms
    assert (err <= ref.abs() * 2.0 ** -21 + 2.0 ** -24).all(), err.max()
    assert torch.equal(p[0].cpu(), ref.float().half())          # hi is the plain fp16 rounding


def test_fp16_subnormal_terms_survive_the_matrix_cores():
    """lo terms of small weights are fp16 subnormals: the MFMA must not flush them (else the path drops to ~2^-11)."""
    a = torch.full((64, 32), 2.0 ** -20)          # hi is a subnormal fp16, exact
    w = torch.eye(32).repeat(2, 1)[:32] * 1.0
    out = runtime.gemm_split(runtime.split_f16(a.to(DEV)), runtime.split_f16(w.to(DEV)))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), a @ w.t())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 512, 384), (1000, 384, 1536), (257, 80, 384), (6400, 1536, 384),
                                   (513, 160, 400), (64, 256, 1024), (4096, 768, 1920), (130, 136, 72)])
def test_gemm_split_against_float64(M, N, K):
    a = synth._normal(f"split/a/{M}/{K}", (M, K))
    w = synth._normal(f"split/w/{N}/{K}", (N, K), K ** -0.5)
    bias = synth._normal(f"split/b/{N}", (N,))
    resid = synth._normal(f"split/r/{M}/{N}", (M, N))
    mask = (torch.arange(M) % 7 != 3)
    ap, wp = runtime.split_f16(a.to(DEV)), runtime.split_f16(w.to(DEV))
    ref = a.double() @ w.double().t()
    scale = ref.abs().max().item()
    out = runtime.gemm_split(ap, wp)
    assert (out.double().cpu() - ref).abs().max().item() < 2e-6 * scale
    # epilogue: bias, GELU, mask before the residual, residual
    out = runtime.gemm_split(ap, wp, bias=bias.to(DEV), resid=resid.to(DEV), mask=mask.to(DEV),
                             flags=runtime.EP_GELU | runtime.EP_MASK_ACC)
    ref2 = F.gelu(ref + bias.double()) * mask[:, None] + resid.double()
    assert (out.double().cpu() - ref2).abs().max().item() < 3e-6 * max(scale, 1.0)
    out = runtime.gemm_split(ap, wp, resid=resid.to(DEV), mask=mask.to(DEV), flags=runtime.EP_MASK_OUT)
    assert (out.double().cpu() - (ref + resid.double()) * mask[:, None]).abs().max().item() < 3e-6 * max(scale, 1.0)
    if N % 8 == 0:   # split-plane output (GELU epilogue: what feeds the second feed-forward Linear)
        outp = runtime.gemm_split(ap, wp, bias=bias.to(DEV), flags=runtime.EP_GELU, out_split=True)
        torch.cuda.synchronize()
        assert (_join(outp) - F.gelu(ref + bias.double())).abs().max().item() < 3e-6 * max(scale, 1.0)


@pytest.mark.parametrize("M,N,K,tile", [(32768, 512, 96, 442), (31000, 384, 64, 342), (32768, 1536, 64, 442), (65536, 128, 64, 242)])
def test_gemm_split_256_row_blocks(M, N, K, tile):
    """Decoder-sized row counts take 256 x (64 TN) blocks (two 32-row tiles per wave, XCD-aware block order): every row and
    feature once, against float64."""
    assert runtime.lib().ispk_gemm_split_f16_tile(M, N, K) == tile
    a = synth._normal(f"split/big/a/{M}/{K}", (M, K))
    w = synth._normal(f"split/big/w/{N}/{K}", (N, K), K ** -0.5)
    resid = synth._normal(f"split/big/r/{M}/{N}", (M, N))
    mask = (torch.arange(M) % 5 != 1)
    ap, wp = runtime.split_f16(a.to(DEV)), runtime.split_f16(w.to(DEV))
    ref = a.double() @ w.double().t()
    out = runtime.gemm_split(ap, wp, resid=resid.to(DEV), mask=mask.to(DEV), flags=runtime.EP_MASK_ACC)
    assert (out.double().cpu() - (ref * mask[:, None] + resid.double())).abs().max().item() < 3e-6 * max(ref.abs().max().item(), 1.0)
    outp = runtime.gemm_split(ap, wp, flags=runtime.EP_GELU, out_split=True)
    torch.cuda.synchronize()
    assert (_join(outp) - F.gelu(ref)).abs().max().item() < 3e-6 * max(ref.abs().max().item(), 1.0)


def test_gemm_split_transposed_frames_output():
    """to_mel: Linear + transpose(1, 2) + mask (model.py:167-168) with frame-contiguous stores."""
    B, T, D, C = 3, 200, 384, 80
    x = synth._normal("split/mel/x", (B, T, D))
    w = synth._normal("split/mel/w", (C, D), D ** -0.5)
    b = synth._normal("split/mel/b", (C,))
    mask = torch.arange(T)[None] < torch.tensor([200, 120, 7])[:, None]
    out = runtime.to_mel_split(runtime.split_f16(x.to(DEV)), runtime.split_f16(w.to(DEV)), b.to(DEV), mask.to(DEV))
    ref = (x.double() @ w.double().t() + b.double()).transpose(1, 2) * mask[:, None]
    assert (out.double().cpu() - ref).abs().max().item() < 5e-6


def test_gemm_split_sliding_window_convolution():
    """Conv1d(kernel 5, padding 2) over a padded channel-last buffer as one GEMM over overlapping rows (alignment.py:69-83)."""
    B, T, C, O = 3, 50, 80, 160
    x = synth._normal("split/conv/x", (B, C, T))
    w = synth._normal("split/conv/w", (O, C, 5), (5 * C) ** -0.5)
    lens = torch.tensor([50, 31, 8])
    xpad = runtime.pad_rows(x.to(DEV), lens.to(DEV), channel_first=True)
    y = runtime.conv5_padded_split(runtime.split_f16(xpad), runtime.split_f16(w.permute(0, 2, 1).reshape(O, 5 * C).contiguous().to(DEV)))
    m = (torch.arange(T)[None] < lens[:, None])[:, None]
    ref = F.conv1d((x * m).double(), w.double(), None, padding=2)
    assert (y[:, :T].double().cpu().transpose(1, 2) - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize("D", [256, 384, 320])
def test_layernorm_split_output(D):
    x = synth._normal(f"split/ln/{D}", (200, D), 2.0, 0.3)
    g, b = synth._normal("split/ln/g", (D,), 0.2, 1.0), synth._normal("split/ln/b", (D,), 0.2)
    mask = torch.arange(200) % 5 != 0
    p = runtime.layernorm_split(x.to(DEV), g.to(DEV), b.to(DEV), row_mask=mask.to(DEV))
    y = runtime.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), row_mask=mask.to(DEV))
    torch.cuda.synchronize()
    assert (_join(p) - y.double().cpu()).abs().max().item() < 2e-6


@pytest.mark.parametrize("B,N,H,ragged", [(2, 100, 6, True), (3, 512, 6, True), (2, 77, 4, False), (1, 1000, 6, False),
                                         (2, 130, 2, True), (2, 64, 1, False), (2, 200, 8, True), (2, 1723, 6, True)])
def test_attention_split_against_the_oracle_and_float64(B, N, H, ragged):
    qkv = synth._normal(f"split/attn/{B}/{N}/{H}", (B, N, H * 64 + 128), 1.5)
    slopes = torch.tensor(synth.alibi_default_slopes(H))
    lens = torch.tensor([N, max(N // 3, 1), max(N - 5, 1)][:B]) if ragged else None
    out = runtime.alibi_mqa_attention_split(qkv.to(DEV), H, slopes.to(DEV), None if lens is None else lens.to(DEV),
                                            out_split=False)
    outp = runtime.alibi_mqa_attention_split(qkv.to(DEV), H, slopes.to(DEV), None if lens is None else lens.to(DEV))
    torch.cuda.synchronize()
    q = qkv[..., :H * 64].view(B, N, H, 64).transpose(1, 2).double()
    k, v = qkv[..., H * 64:H * 64 + 64].double(), qkv[..., H * 64 + 64:].double()
    mask = None if lens is None else (torch.arange(N)[None] < lens[:, None])[:, None, None, :]
    bias = slopes.double().view(H, 1, 1) * orc.alibi_int_bias(N, N).double()
    ref = orc.attend(q, k, v, mask, bias).transpose(1, 2).reshape(B, N, H * 64)
    valid = torch.ones(B, N, dtype=torch.bool) if lens is None else torch.arange(N)[None] < lens[:, None]
    # outputs reach |6| here: 1e-5 absolute is 2e-6 of the scale (v_exp_f32 on arguments down to -100 sets the floor; the
    # exact-fp32 kernel's own bound in test_gpu_kernels.py is 2e-5)
    err = ((out.double().cpu() - ref).abs() * valid[..., None]).max().item()
    assert err < 1e-5, err
    assert ((_join(outp) - ref).abs() * valid[..., None]).max().item() < 1e-5
