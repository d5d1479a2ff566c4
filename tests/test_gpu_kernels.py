"""GPU parity of each HIP kernel against the oracle (oracle/), called through the C ABI (runtime -> libispk.so)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import crc, golden

pytestmark = pytest.mark.gpu

from isp_tts_amd import runtime, synth  # noqa: E402
from oracle import acoustic_oracle as orc  # noqa: E402
from oracle import mas_oracle  # noqa: E402

DEV = "cuda"


def _mas_gpu(x, tl, ml):
    hard, dur, path = runtime.mas(x.to(DEV), tl.to(DEV), ml.to(DEV), want_dur=True, want_path=True)
    torch.cuda.synchronize()
    return hard.cpu().numpy(), dur.cpu().numpy(), path.cpu().numpy()


# ------------------------------------------------------------------------------------------------ MAS
def test_mas_golden_vectors():
    g = golden("mas.npz")
    n_cases = sum(1 for k in g.files if k.endswith("_shape"))
    assert n_cases >= 6
    for i in range(n_cases):
        B, M, L, var = (int(v) for v in g[f"case{i}_shape"])
        kind = str(g[f"case{i}_kind"])
        x, tl, ml = synth.make_mas_logits(B, M, L, bool(var), kind)
        assert crc(x) == int(g[f"case{i}_logits_crc"]), "synthetic generator drifted from the fixtures"
        hard, dur, path = _mas_gpu(x, tl, ml)
        assert np.array_equal(path, g[f"case{i}_path"]), f"case {i}: MAS path differs from the reference"
        assert np.array_equal(dur, g[f"case{i}_dur"].astype(np.int64))
        ref_hard = np.zeros((B, M, L), np.int16)
        bi, mi = np.nonzero(g[f"case{i}_path"] >= 0)
        ref_hard[bi, mi, g[f"case{i}_path"][bi, mi]] = 1
        assert np.array_equal(hard, ref_hard)


@pytest.mark.parametrize("B,M,L,var,kind", [
    (1, 1, 1, False, "realistic"), (3, 7, 1, False, "ties"), (2, 1, 9, False, "realistic"),   # degenerate shapes
    (5, 33, 64, True, "ties"), (5, 130, 65, True, "realistic"), (7, 257, 129, True, "ties"),  # chunk boundaries
    (4, 700, 300, True, "realistic"), (2, 1723, 300, True, "realistic"),                      # recipe maxima
    (3, 90, 400, True, "realistic"), (2, 64, 512, False, "ties"),                             # n < m, max L
    (64, 512, 100, False, "realistic"), (37, 512, 100, True, "realistic"),
    (1, 4096, 200, True, "realistic"), (2, 2048, 512, True, "ties"),                          # the ABI's M / L maxima
])
def test_mas_matches_oracle(B, M, L, var, kind):
    x, tl, ml = synth.make_mas_logits(B, M, L, var, kind)
    hard, dur, path = _mas_gpu(x, tl, ml)
    ref, ref_path = mas_oracle.b_mas(x.numpy(), tl.numpy(), ml.numpy(), return_path=True)
    assert np.array_equal(path, ref_path)
    assert np.array_equal(hard, ref)
    assert np.array_equal(dur, ref.sum(axis=1))


def test_mas_properties_full_size():
    """BASELINE size (B=64, M=512, L=100) with strided input: structural invariants of any MAS result."""
    B, M, L = 64, 512, 100
    x, tl, ml = synth.make_mas_logits(B, M, L, True, "realistic")
    big = torch.zeros(B, M, L + 28)
    big[..., :L] = x
    view = big.to(DEV)[..., :L]              # row stride 128, not contiguous
    assert view.stride(1) == L + 28
    hard, dur, path = runtime.mas(view, tl.to(DEV), ml.to(DEV), want_dur=True, want_path=True)
    hard, dur, path = hard.cpu().numpy(), dur.cpu().numpy(), path.cpu().numpy()
    for b in range(B):
        n, m = int(ml[b]), int(tl[b])
        p = path[b, :n]
        assert p[-1] == m - 1 and (path[b, n:] == -1).all()
        d = np.diff(p)
        assert ((d == 0) | (d == 1)).all(), "path must be monotone with unit steps"
        if n >= m:
            assert p[0] == 0
        assert hard[b].sum() == n and (hard[b, :n].sum(axis=1) == 1).all()
        assert hard[b, n:].sum() == 0 and hard[b, :, m:].sum() == 0
        assert dur[b].sum() == n
    ref = mas_oracle.b_mas(x.numpy(), tl.numpy(), ml.numpy())
    assert np.array_equal(hard, ref)


def test_mas_does_not_mutate_input_and_b_mas_signature():
    from isp_tts_amd.modules.aligner import b_mas
    x, tl, ml = synth.make_mas_logits(4, 96, 23, True, "realistic")
    xin = x.numpy().copy()
    out = b_mas(xin, tl.numpy(), ml.numpy())
    assert out.dtype == np.int16 and out.shape == xin.shape
    assert np.array_equal(xin, x.numpy())
    assert np.array_equal(out, mas_oracle.b_mas(x.numpy(), tl.numpy(), ml.numpy()))


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("D", [64, 256, 384, 1024])
def test_layernorm_plain_adaptive_mask(D):
    B, N = 3, 41
    x = synth._normal(f"t/ln/x{D}", (B, N, D), 2.0, 0.5)
    g, b = synth._normal("t/ln/g", (D,), 0.1, 1.0), synth._normal("t/ln/b", (D,), 0.1)
    mask = torch.arange(N)[None] < torch.tensor([N, 17, 1])[:, None]
    ref = F.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-5)
    out = runtime.layernorm(x.to(DEV), g.to(DEV), b.to(DEV))
    assert (out.cpu().double() - ref).abs().max() < 2e-6
    out = runtime.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), row_mask=mask.to(DEV))
    assert (out.cpu().double() - ref * mask[..., None]).abs().max() < 2e-6
    # adaptive: per-batch and broadcast conditions
    sc, sh = synth._normal("t/ln/sc", (B, D), 0.3, 1.0), synth._normal("t/ln/sh", (B, D), 0.3)
    nref = F.layer_norm(x.double(), (D,), None, None, 1e-5)
    out = runtime.layernorm(x.to(DEV), None, None, sc.to(DEV), sh.to(DEV), rows_per_batch=N, row_mask=mask.to(DEV))
    want = (sc.double()[:, None] * nref + sh.double()[:, None]) * mask[..., None]
    assert (out.cpu().double() - want).abs().max() < 3e-6
    out = runtime.layernorm(x.to(DEV), None, None, sc[:1].to(DEV), sh[:1].to(DEV), rows_per_batch=N)
    want = sc.double()[:1, None] * nref + sh.double()[:1, None]
    assert (out.cpu().double() - want).abs().max() < 3e-6


# ------------------------------------------------------------------------------------------------ GEMM
def _gemm_ref(a, w, bias=None, act=None, resid=None, mask=None, mask_acc=False, mask_out=False):
    v = a.double() @ w.double().T
    if bias is not None:
        v = v + bias.double()
    if act == "gelu":
        v = F.gelu(v)
    if act == "silu":
        v = F.silu(v)
    if mask_acc:
        v = v * mask[..., None]
    if resid is not None:
        v = v + resid.double()
    if mask_out:
        v = v * mask[..., None]
    return v


@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (32768, 512, 384), (300, 1536, 384), (777, 384, 1536),
                                   (64, 64, 8), (200, 384, 256), (130, 80, 384), (5, 3, 24)])
def test_gemm_f32_shapes(M, N, K):
    a = synth._normal(f"t/gemm/a{M}x{K}", (M, K))
    w = synth._normal(f"t/gemm/w{N}x{K}", (N, K), K ** -0.5)
    out = runtime.gemm(a.to(DEV), w.to(DEV)).cpu()
    ref = _gemm_ref(a, w)
    assert (out.double() - ref).abs().max() < 2e-5


def test_gemm_f32_epilogues():
    B, T, K, N = 3, 210, 384, 384
    a = synth._normal("t/gemm/ea", (B, T, K))
    w = synth._normal("t/gemm/ew", (N, K), K ** -0.5)
    bias = synth._normal("t/gemm/eb", (N,))
    resid = synth._normal("t/gemm/er", (B, T, N))
    mask = torch.arange(T)[None] < torch.tensor([T, 100, 1])[:, None]
    d = lambda t: t.to(DEV)  # noqa: E731
    cases = [
        (dict(bias=d(bias)), dict(bias=bias)),
        (dict(bias=d(bias), flags=runtime.EP_GELU), dict(bias=bias, act="gelu")),
        (dict(flags=runtime.EP_SILU), dict(act="silu")),
        (dict(resid=d(resid), mask=d(mask), flags=runtime.EP_MASK_ACC), dict(resid=resid, mask=mask, mask_acc=True)),
        (dict(resid=d(resid), mask=d(mask), flags=runtime.EP_MASK_OUT), dict(resid=resid, mask=mask, mask_out=True)),
        (dict(resid=d(resid)), dict(resid=resid)),
    ]
    for kw, rkw in cases:
        out = runtime.gemm(d(a), d(w), **kw).cpu()
        assert (out.double() - _gemm_ref(a, w, **rkw)).abs().max() < 2e-5, kw.keys()


def test_gemm_strided_operands():
    """A may be a column slice of a wider buffer (leading stride > K), like the K/V halves of the fused projection."""
    a_full = synth._normal("t/gemm/sa", (500, 512))
    w = synth._normal("t/gemm/sw", (384, 384), 384 ** -0.5)
    ad = a_full.to(DEV)
    out = runtime.gemm(ad[:, 128:], w.to(DEV)).cpu()
    assert (out.double() - _gemm_ref(a_full[:, 128:], w)).abs().max() < 2e-5


def test_to_mel_transposed_masked_store():
    B, T, D, C = 3, 203, 384, 80
    x = synth._normal("t/mel/x", (B, T, D))
    w, b = synth._normal("t/mel/w", (C, D), D ** -0.5), synth._normal("t/mel/b", (C,))
    mask = torch.arange(T)[None] < torch.tensor([T, 77, 5])[:, None]
    ref = (x.double() @ w.double().T + b.double()).transpose(1, 2) * mask[:, None]
    out = runtime.to_mel(x.to(DEV), w.to(DEV), b.to(DEV), mask.to(DEV)).cpu()
    assert out.shape == (B, C, T) and (out.double() - ref).abs().max() < 2e-5
    out = runtime.to_mel(x.to(DEV), w.to(DEV), b.to(DEV), None).cpu()
    assert (out.double() - (x.double() @ w.double().T + b.double()).transpose(1, 2)).abs().max() < 2e-5


@pytest.mark.parametrize("B,T,D", [(3, 203, 384), (2, 512, 384), (5, 64, 256)])
def test_to_mel_bf16_rows_t(B, T, D):
    """bf16 to_mel = panel GEMM with frames as rows + ISPK_EP_ROWS_T store: float64 reference on the bf16-rounded operands
    (products of bf16 values are exact in fp32; only the summation order differs)."""
    C = 80
    x = _bf(synth._normal(f"t/melb/x{T}", (B, T, D)))
    w, b = _bf(synth._normal(f"t/melb/w{D}", (C, D), D ** -0.5)), synth._normal("t/melb/b", (C,))
    lens = torch.tensor([T, max(1, T // 3), 5, T - 1, 1][:B])
    mask = torch.arange(T)[None] < lens[:, None]
    ref = (x.double() @ w.double().T + b.double()).transpose(1, 2) * mask[:, None]
    out = runtime.to_mel(x.to(DEV), w.to(DEV), b.to(DEV), mask.to(DEV)).cpu()
    assert out.shape == (B, C, T) and out.dtype == torch.float32
    assert (out.double() - ref).abs().max() < 2e-5
    out = runtime.to_mel(x.to(DEV), w.to(DEV), b.to(DEV), None).cpu()
    assert (out.double() - (x.double() @ w.double().T + b.double()).transpose(1, 2)).abs().max() < 2e-5


@pytest.mark.parametrize("M,N,K", [(7, 32, 65), (200, 256, 3), (200, 3, 256), (1, 256, 32), (200, 256, 2)])
def test_linear_small(M, N, K):
    a, w = synth._normal("t/ls/a", (M, K)), synth._normal("t/ls/w", (N, K), K ** -0.5)
    bias, resid = synth._normal("t/ls/b", (N,)), synth._normal("t/ls/r", (M, N))
    out = runtime.linear_small(a.to(DEV), w.to(DEV), bias.to(DEV), resid.to(DEV), act=runtime.EP_SILU).cpu()
    ref = F.silu(a.double() @ w.double().T + bias.double()) + resid.double()
    assert (out.double() - ref).abs().max() < 1e-5
    wide = synth._normal("t/ls/wide", (N, K + 5), K ** -0.5)
    out = runtime.linear_small(a.to(DEV), wide.to(DEV)[:, :K]).cpu()      # column slice of a wider weight
    assert (out.double() - a.double() @ wide[:, :K].double().T).abs().max() < 1e-5


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,N,H,lens", [(2, 100, 6, [100, 73]), (2, 512, 6, [512, 390]), (3, 37, 4, [37, 1, 20]),
                                        (2, 64, 6, None), (1, 129, 8, [65]), (2, 1000, 6, [1000, 333]),
                                        (2, 1723, 6, [1723, 911])])      # 1,723 frames: the recipes' data bound (core.yaml:33-47)
def test_attention_matches_reference_algorithm(B, N, H, lens):
    """Kernel vs the oracle's `attend` (materialised bias + masked_fill(min/2) + SDPA, attend.py:49-122)."""
    q = synth._normal(f"t/at/q{N}", (B, N, H * 64))
    kv = synth._normal(f"t/at/kv{N}", (B, N, 128))
    slopes = torch.tensor(synth.alibi_default_slopes(H)) * 1.1
    mask = None if lens is None else torch.arange(N)[None] < torch.tensor(lens)[:, None]
    bias = slopes.view(H, 1, 1) * orc.alibi_int_bias(N, N).float()
    ref = orc.attend(q.view(B, N, H, 64).transpose(1, 2), kv[..., :64], kv[..., 64:],
                     None if mask is None else mask[:, None, None, :], bias)
    ref = ref.transpose(1, 2).reshape(B, N, H * 64)
    qkv = torch.cat([q, kv], dim=-1).to(DEV)
    key_len = None if lens is None else torch.tensor(lens, device=DEV)
    out = runtime.alibi_mqa_attention(qkv, H, slopes.to(DEV), key_len).cpu()
    # float64 evaluation of the same formula: both fp32 implementations are judged against it.  The ALiBi bias reaches
    # slope * N (550 at N=1000, where one fp32 ulp is 6e-5), so the achievable agreement scales with N.
    qd = q.view(B, N, H, 64).transpose(1, 2).double()
    sc = qd @ kv[..., :64].double().transpose(1, 2)[:, None] / 8.0 + bias.double()[None]
    if mask is not None:
        sc = sc.masked_fill(~mask[:, None, None, :], float("-inf"))
    exact = (sc.softmax(-1) @ kv[..., 64:].double()[:, None]).transpose(1, 2).reshape(B, N, H * 64)
    err_kernel, err_ref = (out.double() - exact).abs().max().item(), (ref.double() - exact).abs().max().item()
    tol = 2e-5 * max(1.0, N / 256)
    assert err_kernel < tol, (err_kernel, err_ref)
    assert (out - ref).abs().max() < 2 * tol


def test_attention_online_softmax_rescale_is_exercised():
    """A key far down the sequence with a huge score forces the running max to jump in a late tile."""
    B, N, H = 1, 256, 6
    q = synth._normal("t/at/spike_q", (B, N, H * 64))
    kv = synth._normal("t/at/spike_kv", (B, N, 128))
    kv[0, 200, :64] = q[0, 10, :64] * 4.0          # key 200 aligns with query 10 / head 0
    slopes = torch.full((H,), 0.01)
    bias = slopes.view(H, 1, 1) * orc.alibi_int_bias(N, N).float()
    ref = orc.attend(q.view(B, N, H, 64).transpose(1, 2), kv[..., :64], kv[..., 64:], None, bias)
    ref = ref.transpose(1, 2).reshape(B, N, H * 64)
    out = runtime.alibi_mqa_attention(torch.cat([q, kv], -1).to(DEV), H, slopes.to(DEV), None).cpu()
    assert (out - ref).abs().max() < 2e-5


# ------------------------------------------------------------------------------------------------ ABI errors
def test_abi_argument_errors_are_reported():
    lib = runtime.lib()
    x = torch.zeros(8, 100, device=DEV)
    with pytest.raises(runtime.IspkError, match="multiple of 64"):
        runtime.layernorm(x, None, None)
    with pytest.raises(runtime.IspkError, match="multiple of 8"):
        runtime.gemm(torch.zeros(8, 100, device=DEV), torch.zeros(16, 100, device=DEV))
    with pytest.raises(runtime.IspkError, match="GPU tensors"):
        runtime.gemm(torch.zeros(8, 64), torch.zeros(16, 64))
    assert lib.ispk_mas_f32(None, None, None, None, None, None, 1, 1, 1, 1, 1, None) == -1
    # the LayerNorm-in-GEMM entries: shapes they are not built for are refused, never silently served by another path
    g128, w128 = torch.ones(128, device=DEV), torch.zeros(64, 128, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(runtime.IspkError, match="built for 256 / 384"):
        runtime.gemm_lnin(torch.zeros(8, 128, device=DEV), None, g128, g128, w128)
    g, w1 = torch.ones(384, device=DEV), torch.zeros(1536, 384, device=DEV, dtype=torch.bfloat16)
    w2p = torch.zeros(48, 384, 32, device=DEV, dtype=torch.bfloat16)
    assert lib.ispk_ffn_bf16_prenorm(None, 384, g.data_ptr(), g.data_ptr(), 1e-5, w1.data_ptr(), 384, w2p.data_ptr(), None,
                                     None, None, 384, 8, 384, 1536, 0, None, 1e-5, None) < 0
    assert "ffn" in lib.ispk_last_error_string().decode()
    x = torch.zeros(8, 384, device=DEV)
    assert lib.ispk_ffn_bf16_prenorm(x.data_ptr(), 384, None, g.data_ptr(), 1e-5, w1.data_ptr(), 384, w2p.data_ptr(), None,
                                     None, x.data_ptr(), 384, 8, 384, 1536, 0, None, 1e-5, None) == -1   # ISPK_E_NULL
    assert lib.ispk_ffn_bf16_prenorm(x.data_ptr(), 384, g.data_ptr(), g.data_ptr(), 1e-5, w1.data_ptr(), 384, w2p.data_ptr(),
                                     None, None, x.data_ptr(), 384, 0, 384, 1536, 0, None, 1e-5, None) == 0   # zero rows: no launch
    name, cus = runtime.device_info()
    assert "gfx950" in name and cus >= 128


# ------------------------------------------------------------------------------------------------ bf16 path
def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (32768, 512, 384), (300, 1536, 384), (777, 384, 1536),
                                   (64, 64, 8), (130, 80, 384),
                                   (8200, 384, 1536), (8300, 256, 1024),      # "wide" kernel (long K), ragged M
                                   (800, 384, 1536), (77, 256, 1024), (130, 768, 1920), (832, 128, 768),   # ... at few rows
                                   (6400, 384, 256), (515, 1024, 256)])       # "panel" kernel with K = 256
def test_gemm_bf16_shapes(M, N, K):
    """bf16 operands, fp32 accumulation: compared with a float64 product of the SAME bf16-rounded operands, so the only
    differences are accumulation order (fp32) and the output rounding (bf16: 2^-9 relative)."""
    a, w = _bf(synth._normal(f"t/gemm/a{M}x{K}", (M, K))), _bf(synth._normal(f"t/gemm/w{N}x{K}", (N, K), K ** -0.5))
    ref = a.double() @ w.double().T
    out32 = runtime.gemm(a.to(DEV), w.to(DEV), out_dtype=torch.float32).cpu()
    assert (out32.double() - ref).abs().max() < 2e-5
    out16 = runtime.gemm(a.to(DEV), w.to(DEV)).cpu()
    assert out16.dtype == torch.bfloat16
    assert ((out16.double() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-6).all()


@pytest.mark.parametrize("M,N,K", [(800, 384, 1536), (77, 256, 1024), (832, 768, 1920), (130, 128, 768), (1600, 384, 1536)])
def test_gemm_bf16_split_k_for_few_rows(M, N, K):
    """ispk_gemm_bf16_splitk (few rows, long K: the text-side stacks of a rank under strong scaling): K slices on separate
    workgroups into fp32 slabs + one ordered combine pass that applies the epilogue - against float64 on the same bf16 operands
    for every epilogue the model uses with these shapes (GELU + bias, residual with either mask position, bf16 residual, bf16
    output), deterministic, and the plan: split only below 2,048 rows."""
    lib = runtime.lib()
    ks = lib.ispk_gemm_bf16_splitk_plan(M, N, K, 0)
    assert ks >= 2 and K % (64 * ks) == 0 and lib.ispk_gemm_bf16_splitk_plan(4096, N, K, 0) == 1
    assert lib.ispk_gemm_bf16_splitk_plan(M, N, 384, 0) == 1          # (K = 256 / 384 is the panel kernel's)
    a, w = _bf(synth._normal(f"t/sk/a{M}{K}", (M, K))), _bf(synth._normal(f"t/sk/w{N}{K}", (N, K), K ** -0.5))
    bias, resid = synth._normal("t/sk/b", (N,)), synth._normal(f"t/sk/r{M}{N}", (M, N))
    mask = torch.arange(M) % 5 != 2
    d = lambda t: t.to(DEV)  # noqa: E731
    out = runtime.gemm(d(a), d(w), out_dtype=torch.float32)
    assert torch.equal(out, runtime.gemm(d(a), d(w), out_dtype=torch.float32))
    assert (out.cpu().double() - _gemm_ref(a, w)).abs().max() < 2e-5
    out = runtime.gemm(d(a), d(w), bias=d(bias), flags=runtime.EP_GELU).cpu()
    ref = _gemm_ref(a, w, bias=bias, act="gelu")
    assert out.dtype == torch.bfloat16 and ((out.double() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-5).all()
    out = runtime.gemm(d(a), d(w), resid=d(resid), mask=d(mask), flags=runtime.EP_MASK_ACC, out_dtype=torch.float32).cpu()
    assert (out.double() - _gemm_ref(a, w, resid=resid, mask=mask, mask_acc=True)).abs().max() < 2e-5
    out = runtime.gemm(d(a), d(w), resid=d(_bf(resid)), mask=d(mask), flags=runtime.EP_MASK_OUT, out_dtype=torch.float32).cpu()
    assert (out.double() - _gemm_ref(a, w, resid=_bf(resid), mask=mask, mask_out=True)).abs().max() < 2e-5
    assert out[~mask].abs().max().item() == 0.0
    # the C ABI refuses shapes outside the plan instead of serving them some other way
    ws = torch.empty(2 * M * N, device=DEV)
    c = torch.empty(M, N, device=DEV)
    assert lib.ispk_gemm_bf16_splitk(d(a).data_ptr(), K, d(w).data_ptr(), K, c.data_ptr(), N, None, None, 0, None, M, N, K, 0,
                                     ws.data_ptr(), 7, None) < 0


@pytest.mark.parametrize("K", [384, 1536])
def test_gemm_bf16_epilogues(K):
    B, T, N = 3, 2810, 384      # 8430 rows: K=384 -> panel kernel, K=1536 -> wide kernel
    a, w = _bf(synth._normal("t/gemm/ea", (B, T, K))), _bf(synth._normal("t/gemm/ew", (N, K), K ** -0.5))
    bias, resid = synth._normal("t/gemm/eb", (N,)), synth._normal("t/gemm/er", (B, T, N))
    mask = torch.arange(T)[None] < torch.tensor([T, 100, 1])[:, None]
    d = lambda t: t.to(DEV)  # noqa: E731
    out = runtime.gemm(d(a), d(w), bias=d(bias), flags=runtime.EP_GELU).cpu()      # bf16 out: fast GELU + rounding
    ref = _gemm_ref(a, w, bias=bias, act="gelu")
    assert ((out.double() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-5).all()
    out = runtime.gemm(d(a), d(w), bias=d(bias), flags=runtime.EP_GELU, out_dtype=torch.float32).cpu()
    assert (out.double() - _gemm_ref(a, w, bias=bias, act="gelu")).abs().max() < 2e-5
    out = runtime.gemm(d(a), d(w), resid=d(resid), mask=d(mask), flags=runtime.EP_MASK_ACC, out_dtype=torch.float32).cpu()
    assert (out.double() - _gemm_ref(a, w, resid=resid, mask=mask, mask_acc=True)).abs().max() < 2e-5
    out = runtime.gemm(d(a), d(w), resid=d(_bf(resid)), mask=d(mask), flags=runtime.EP_MASK_OUT,
                       out_dtype=torch.float32).cpu()
    assert (out.double() - _gemm_ref(a, w, resid=_bf(resid), mask=mask, mask_out=True)).abs().max() < 2e-5


@pytest.mark.parametrize("B,N,H,lens", [(2, 100, 6, [100, 73]), (2, 512, 6, [512, 390]), (3, 37, 4, [37, 1, 20]),
                                        (2, 64, 6, None), (1, 129, 8, [65]), (2, 700, 6, [700, 531]), (1, 1100, 4, None),
                                        (2, 300, 2, [300, 129]), (1, 200, 1, None), (2, 1723, 6, [1723, 911])])
def test_attention_bf16(B, N, H, lens):
    """bf16 Q/K/V and bf16 P (8 mantissa bits), fp32 statistics: compared with float64 attention on the same
    bf16-rounded inputs.  Bar: 1.5e-2 absolute on O(1) outputs (P quantisation 2^-9 relative per weight)."""
    q = _bf(synth._normal(f"t/at/q{N}", (B, N, H * 64)))
    kv = _bf(synth._normal(f"t/at/kv{N}", (B, N, 128)))
    slopes = torch.tensor(synth.alibi_default_slopes(H)) * 1.1
    mask = None if lens is None else torch.arange(N)[None] < torch.tensor(lens)[:, None]
    bias = slopes.view(H, 1, 1).double() * orc.alibi_int_bias(N, N).double()
    qd = q.view(B, N, H, 64).transpose(1, 2).double()
    sc = qd @ kv[..., :64].double().transpose(1, 2)[:, None] / 8.0 + bias[None]
    if mask is not None:
        sc = sc.masked_fill(~mask[:, None, None, :], float("-inf"))
    exact = (sc.softmax(-1) @ kv[..., 64:].double()[:, None]).transpose(1, 2).reshape(B, N, H * 64)
    key_len = None if lens is None else torch.tensor(lens, device=DEV)
    out = runtime.alibi_mqa_attention(torch.cat([q, kv], -1).to(DEV), H, slopes.to(DEV), key_len).cpu()
    assert out.dtype == torch.bfloat16
    err = (out.double() - exact).abs().max().item()
    assert err < 1.5e-2, err


@pytest.mark.parametrize("R,D,N", [(5000, 384, 512), (777, 256, 384), (128 * 130, 384, 512), (31, 384, 192)])
def test_linear_with_layernorm_prologue(R, D, N):
    """ispk_gemm_bf16_lnin == LayerNorm kernel (bf16 out) followed by ispk_gemm_bf16, given the rows' (mean, rstd):
    the normalised operand is rounded to bf16 exactly as the LayerNorm kernel rounds it, up to 1-ulp flips where the
    two reduction trees leave the fp32 value on a rounding boundary."""
    x = synth._normal(f"t/lnin/x{R}", (R, D), 2.0, 0.5)
    g, b = synth._normal("t/lnin/g", (D,), 0.1, 1.0), synth._normal("t/lnin/b", (D,), 0.1)
    w = _bf(synth._normal(f"t/lnin/w{D}", (N, D), D ** -0.5))
    mean = x.double().mean(1)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mean, rstd], 1).float().contiguous()
    d = lambda t: t.to(DEV)  # noqa: E731
    out = runtime.gemm_lnin(d(x), d(stats), d(g), d(b), d(w)).cpu()
    h = runtime.layernorm(d(x), d(g), d(b), out_dtype=torch.bfloat16)
    ref = runtime.gemm(h, d(w)).cpu()
    assert out.dtype == torch.bfloat16 and out.shape == (R, N)
    err = (out.float() - ref.float()).abs()
    assert err.max().item() <= 2 ** -5 and err.pow(2).mean().sqrt().item() <= 2e-3
    # and against fp64 on the fp32-normalised operand: bounded by the bf16 operand rounding
    hn = ((x.double() - mean[:, None]) * rstd[:, None] * g.double() + b.double())
    ref64 = hn @ w.double().t()
    assert (out.double() - ref64).abs().max().item() <= 0.08
    # row_stats NULL: the kernel's waves compute the statistics themselves (deterministic), also under other epilogues
    own = runtime.gemm_lnin(d(x), None, d(g), d(b), d(w))
    assert torch.equal(own, runtime.gemm_lnin(d(x), None, d(g), d(b), d(w)))
    err = (own.cpu().float() - ref.float()).abs()
    assert err.max().item() <= 2 ** -5 and err.pow(2).mean().sqrt().item() <= 2e-3
    gel = runtime.gemm_lnin(d(x), None, d(g), d(b), d(w), flags=runtime.EP_GELU).cpu()
    gref = runtime.gemm(h, d(w), flags=runtime.EP_GELU).cpu()
    assert (gel.float() - gref.float()).abs().max().item() <= 2 ** -5
    f32 = runtime.gemm_lnin(d(x), None, d(g), d(b), d(w), out_dtype=torch.float32).cpu()
    assert (f32.double() - ref64).abs().max().item() <= 0.08




@pytest.mark.parametrize("R,Fi,masked", [(128 * 9 + 17, 1536, True), (128 * 3, 1536, False), (70, 64, True), (128 * 40, 1536, True)])
def test_eight_wave_ffn_kernel(R, Fi, masked):
    """ispk_ffn_bf16_prenorm2 (csrc/ffn2.hip: two waves per SIMD, K-split first product, partial sums and P handed over
    through LDS, three-stage pipeline) against float64 on the un-rounded LayerNorm, and against the four-wave kernel -
    same math, another summation order and a coarser (still sub-bf16-ulp) GELU; ragged row counts, masked rows exactly
    zero, the output rows' statistics, run-to-run determinism."""
    D = 384
    x = synth._normal(f"t/ffn2/x{R}", (R, D), 1.5, 0.4)
    w1, w2 = _bf(synth._normal(f"t/ffn2/w1{Fi}", (Fi, D), D ** -0.5)), _bf(synth._normal(f"t/ffn2/w2{Fi}", (D, Fi), Fi ** -0.5))
    g, b = synth._normal("t/ffn2/g", (D,), 0.1, 1.0), synth._normal("t/ffn2/b", (D,), 0.1)
    mask = (torch.arange(R) % 7 != 3) if masked else None
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    w2c = runtime.ffn_chunk_w2(d(w2))
    assert torch.equal(w2c.cpu(), w2.view(D, Fi // 32, 32).permute(1, 0, 2).contiguous())
    fl = runtime.EP_MASK_OUT if masked else 0
    out, stats = runtime.ffn_prenorm2(d(x), d(g), d(b), d(w1), w2c, mask=d(mask), flags=fl, want_stats=True)
    again = runtime.ffn_prenorm2(d(x), d(g), d(b), d(w1), w2c, mask=d(mask), flags=fl)
    assert torch.equal(out, again)
    out = out.cpu()
    x64 = x.double()
    hn = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5) * g.double() + b.double()
    ref64 = x64 + F.gelu(hn @ w1.double().t()) @ w2.double().t()
    if masked:
        ref64 = ref64 * mask[:, None]
        assert out[~mask].abs().max().item() == 0.0
    err = (out.double() - ref64).abs()
    print(f"ffn2 R={R} Fi={Fi}: max err vs float64 = {err.max().item():.3e}, rms = {err.pow(2).mean().sqrt().item():.3e}")
    assert err.max().item() <= 0.06 and err.pow(2).mean().sqrt().item() <= 6e-3
    if Fi % 32 == 0 and Fi >= 64:
        old = runtime.ffn_prenorm(d(x), d(g), d(b), d(w1), runtime.ffn_pack_w2(d(w2)), mask=d(mask), flags=fl).cpu()
        e2 = (out - old).abs()
        assert e2.max().item() <= 3e-2 and e2.pow(2).mean().sqrt().item() <= 2e-3
    o64 = out.double()
    assert (stats[:, 0].cpu().double() - o64.mean(1)).abs().max().item() <= 1e-6
    rstd = 1.0 / torch.sqrt(o64.var(1, unbiased=False) + 1e-5)
    assert ((stats[:, 1].cpu().double() - rstd) / rstd).abs().max().item() <= 1e-5


@pytest.mark.parametrize("R,Fi,masked", [(128 * 9 + 17, 1536, True), (128 * 3, 1536, False), (70, 64, True), (128 * 40, 1536, True)])
def test_projection_and_ffn_kernel(R, Fi, masked):
    """ispk_attn_out_ffn_bf16 (csrc/ffn2.hip, projection mode): x1 = x + mask * (o Wo^T) in the accumulators, LayerNorm from the
    accumulators, the feed-forward block on top - against float64 on the un-rounded x1 / LayerNorm, and against the two
    launches it replaces (to_out GEMM with residual epilogue, then ispk_ffn_bf16_prenorm2): same products in another summation
    order.  Ragged row counts, masked rows exactly zero (garbage - NaN bit patterns - in the masked rows of the attention
    output must not leak), output statistics, determinism."""
    D = 384
    x = synth._normal(f"t/pj/x{R}", (R, D), 1.5, 0.4)
    o = _bf(synth._normal(f"t/pj/o{R}", (R, D), 1.0))
    wo = _bf(synth._normal("t/pj/wo", (D, D), D ** -0.5))
    w1, w2 = _bf(synth._normal(f"t/pj/w1{Fi}", (Fi, D), D ** -0.5)), _bf(synth._normal(f"t/pj/w2{Fi}", (D, Fi), Fi ** -0.5))
    g, b = synth._normal("t/pj/g", (D,), 0.1, 1.0), synth._normal("t/pj/b", (D,), 0.1)
    mask = (torch.arange(R) % 7 != 3) if masked else None
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    woc, w2c = runtime.ffn_chunk_w2(d(wo)), runtime.ffn_chunk_w2(d(w2))
    o_dev = d(o).clone()
    if masked:
        o_dev[~d(mask)] = float("nan")        # what a padded query row may hold
    out, stats = runtime.attn_out_ffn(d(x), o_dev, woc, d(g), d(b), d(w1), w2c, mask=d(mask), want_stats=True)
    again = runtime.attn_out_ffn(d(x), o_dev, woc, d(g), d(b), d(w1), w2c, mask=d(mask))
    assert torch.equal(out, again)
    out = out.cpu()
    assert torch.isfinite(out).all()
    x64 = x.double()
    pr = o.double() @ wo.double().t()
    x1 = x64 + (pr * mask[:, None] if masked else pr)
    hn = (x1 - x1.mean(1, keepdim=True)) / torch.sqrt(x1.var(1, unbiased=False, keepdim=True) + 1e-5) * g.double() + b.double()
    ref64 = x1 + F.gelu(hn @ w1.double().t()) @ w2.double().t()
    if masked:
        ref64 = ref64 * mask[:, None]
        assert out[~mask].abs().max().item() == 0.0
    err = (out.double() - ref64).abs()
    print(f"attn_out_ffn R={R} Fi={Fi}: max err vs float64 = {err.max().item():.3e}, rms = {err.pow(2).mean().sqrt().item():.3e}")
    assert err.max().item() <= 0.06 and err.pow(2).mean().sqrt().item() <= 6e-3
    # the two launches it replaces
    o_clean = d(o) if not masked else torch.where(d(mask)[:, None], d(o), torch.zeros_like(d(o)))
    x1k = runtime.gemm(o_clean, d(wo), resid=d(x), mask=d(mask), flags=runtime.EP_MASK_ACC if masked else 0, out_dtype=torch.float32)
    two = runtime.ffn_prenorm2(x1k, d(g), d(b), d(w1), w2c, mask=d(mask), flags=runtime.EP_MASK_OUT if masked else 0).cpu()
    e2 = (out - two).abs()
    print(f"   vs to_out GEMM + ffn_prenorm2: max {e2.max().item():.3e}, rms {e2.pow(2).mean().sqrt().item():.3e}")
    assert e2.max().item() <= 3e-2 and e2.pow(2).mean().sqrt().item() <= 2e-3
    o64 = out.double()
    assert (stats[:, 0].cpu().double() - o64.mean(1)).abs().max().item() <= 1e-6
    rstd = 1.0 / torch.sqrt(o64.var(1, unbiased=False) + 1e-5)
    assert ((stats[:, 1].cpu().double() - rstd) / rstd).abs().max().item() <= 1e-5
    # ispk_attn_out_ffn_norm_bf16: the same rows and the stack's FINAL LayerNorm of them (row-masked), bf16 / fp32; rows not stored
    fg, fb_ = synth._normal("t/pj/fg", (D,), 0.1, 1.0), synth._normal("t/pj/fb", (D,), 0.1)
    for fdt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2.0 ** -7)):
        out_f, ln = runtime.attn_out_ffn(d(x), o_dev, woc, d(g), d(b), d(w1), w2c, mask=d(mask), final_norm=(d(fg), d(fb_), 1e-5, True, fdt))
        none, ln2 = runtime.attn_out_ffn(d(x), o_dev, woc, d(g), d(b), d(w1), w2c, mask=d(mask), final_norm=(d(fg), d(fb_), 1e-5, True, fdt),
                                         want_out=False)
        assert none is None and torch.equal(ln, ln2) and torch.equal(out_f.cpu(), out) and ln.dtype == fdt
        sep = runtime.layernorm(out_f, d(fg), d(fb_), row_mask=d(mask), eps=1e-5, out_dtype=fdt)
        assert (ln.float() - sep.float()).abs().max().item() <= tol * max(1.0, sep.float().abs().max().item())
        if masked:
            assert ln.cpu()[~mask].float().abs().max().item() == 0.0
    # ispk_attn_out_ffn_qkv_bf16: the same rows, and the NEXT layer's attention_norm + q/kv projection from the epilogue
    wq = _bf(synth._normal("t/pj/wq", (512, D), D ** -0.5))
    g2, b2 = synth._normal("t/pj/g2", (D,), 0.1, 1.0), synth._normal("t/pj/b2", (D,), 0.1)
    wqc = runtime.chunk_k16(d(wq))
    assert torch.equal(wqc.cpu(), wq.view(512, D // 16, 16).permute(1, 0, 2).contiguous())
    out2, qkv = runtime.attn_out_ffn(d(x), o_dev, woc, d(g), d(b), d(w1), w2c, mask=d(mask), next_qkv=(d(g2), d(b2), 1e-5, wqc))
    out3, qkv3 = runtime.attn_out_ffn(d(x), o_dev, woc, d(g), d(b), d(w1), w2c, mask=d(mask), next_qkv=(d(g2), d(b2), 1e-5, wqc))
    assert torch.equal(out2.cpu(), out) and torch.equal(qkv, qkv3) and torch.equal(out2, out3)
    assert qkv.dtype == torch.bfloat16 and qkv.shape == (R, 512)
    two_q = runtime.gemm_lnin(out2, None, d(g2), d(b2), d(wq)).cpu().float()       # the launch it replaces
    eq = (qkv.cpu().float() - two_q).abs()
    hn2 = (o64 - o64.mean(1, keepdim=True)) * rstd[:, None] * g2.double() + b2.double()
    e64 = (qkv.cpu().double() - hn2 @ wq.double().t()).abs()
    print(f"   q/kv epilogue: vs gemm_lnin max {eq.max().item():.3e} rms {eq.pow(2).mean().sqrt().item():.3e}; vs float64 max {e64.max().item():.3e}")
    assert eq.max().item() <= 2 ** -5 and eq.pow(2).mean().sqrt().item() <= 2e-3 and e64.max().item() <= 0.08


@pytest.mark.parametrize("R,splits,masked", [(6400, 4, True), (128 * 9 + 17, 8, True), (1500, 3, False), (800, 16, True), (130, 12, False),
                                             (16384, 2, True)])
def test_split_ffn_for_small_batches(R, splits, masked):
    """ispk_ffn_bf16_prenorm2_split + ispk_ffn_combine_ln_f32 (inner dimension split over workgroups, partials added in split
    order with residual and mask, consumer LayerNorm from the same pass) against the unsplit eight-wave kernel - the same
    products, only the fp32 accumulation is cut into `splits` segments - and float64; LayerNorm output against float64 on
    the kernel's own y; masked rows exactly zero; run-to-run determinism."""
    D, Fi = 384, 1536
    x = synth._normal(f"t/ffns/x{R}", (R, D), 1.5, 0.4)
    w1, w2 = _bf(synth._normal("t/ffns/w1", (Fi, D), D ** -0.5)), _bf(synth._normal("t/ffns/w2", (D, Fi), Fi ** -0.5))
    g, b = synth._normal("t/ffns/g", (D,), 0.1, 1.0), synth._normal("t/ffns/b", (D,), 0.1)
    ng, nb = synth._normal("t/ffns/ng", (D,), 0.1, 1.0), synth._normal("t/ffns/nb", (D,), 0.1)
    mask = (torch.arange(R) % 7 != 3) if masked else None
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    w2c = runtime.ffn_chunk_w2(d(w2))
    nn_ = (d(ng), d(nb), 1e-5, False, torch.bfloat16)
    y, hn = runtime.ffn_prenorm2_split(d(x), d(g), d(b), d(w1), w2c, d(mask), splits, next_norm=nn_)
    y2, hn2 = runtime.ffn_prenorm2_split(d(x), d(g), d(b), d(w1), w2c, d(mask), splits, next_norm=nn_)
    assert torch.equal(y, y2) and torch.equal(hn, hn2)
    whole = runtime.ffn_prenorm2(d(x), d(g), d(b), d(w1), w2c, mask=d(mask), flags=runtime.EP_MASK_OUT if masked else 0)
    e = (y - whole).abs()
    assert e.max().item() <= 2e-5 * max(1.0, whole.abs().max().item()), e.max().item()   # fp32 sums in another grouping
    x64 = x.double()
    h64 = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5) * g.double() + b.double()
    ref64 = x64 + F.gelu(h64 @ w1.double().t()) @ w2.double().t()
    if masked:
        ref64 = ref64 * mask[:, None]
        assert y.cpu()[~mask].abs().max().item() == 0.0
    err = (y.cpu().double() - ref64).abs()
    assert err.max().item() <= 0.06 and err.pow(2).mean().sqrt().item() <= 6e-3
    y64 = y.cpu().double()
    ln64 = (y64 - y64.mean(1, keepdim=True)) / torch.sqrt(y64.var(1, unbiased=False, keepdim=True) + 1e-5) * ng.double() + nb.double()
    assert (hn.cpu().double() - ln64).abs().max().item() <= 2.0 ** -7 * max(1.0, ln64.abs().max().item())   # bf16 rounding
    y3, none = runtime.ffn_prenorm2_split(d(x), d(g), d(b), d(w1), w2c, d(mask), splits)
    assert none is None and torch.equal(y3, y)


@pytest.mark.parametrize("R,splits,masked", [(6400, 4, True), (800, 8, True), (130, 12, False)])
def test_projection_in_the_split_ffn(R, splits, masked):
    """ispk_attn_out_ffn_split_bf16 + ispk_ffn_combine_ln_f32 with x = NULL: every split forms x1 = x + mask * (o Wo^T) in its
    accumulators, only split 0's partial product carries it - against the to_out GEMM followed by the plain split path, and
    float64; NaN bit patterns in masked rows of the attention output do not leak; determinism."""
    D, Fi = 384, 1536
    x = synth._normal(f"t/pjs/x{R}", (R, D), 1.5, 0.4)
    o = _bf(synth._normal(f"t/pjs/o{R}", (R, D), 1.0))
    wo = _bf(synth._normal("t/pjs/wo", (D, D), D ** -0.5))
    w1, w2 = _bf(synth._normal("t/pjs/w1", (Fi, D), D ** -0.5)), _bf(synth._normal("t/pjs/w2", (D, Fi), Fi ** -0.5))
    g, b = synth._normal("t/pjs/g", (D,), 0.1, 1.0), synth._normal("t/pjs/b", (D,), 0.1)
    ng, nb = synth._normal("t/pjs/ng", (D,), 0.1, 1.0), synth._normal("t/pjs/nb", (D,), 0.1)
    mask = (torch.arange(R) % 7 != 3) if masked else None
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    woc, w2c = runtime.ffn_chunk_w2(d(wo)), runtime.ffn_chunk_w2(d(w2))
    o_dev = d(o).clone()
    if masked:
        o_dev[~d(mask)] = float("nan")
    nn_ = (d(ng), d(nb), 1e-5, False, torch.bfloat16)
    y, hn = runtime.ffn_prenorm2_split(d(x), d(g), d(b), d(w1), w2c, d(mask), splits, next_norm=nn_, attn_proj=(o_dev, woc))
    y2, hn2 = runtime.ffn_prenorm2_split(d(x), d(g), d(b), d(w1), w2c, d(mask), splits, next_norm=nn_, attn_proj=(o_dev, woc))
    assert torch.equal(y, y2) and torch.equal(hn, hn2) and torch.isfinite(y).all() and torch.isfinite(hn.float()).all()
    o_clean = d(o) if not masked else torch.where(d(mask)[:, None], d(o), torch.zeros_like(d(o)))
    x1k = runtime.gemm(o_clean, d(wo), resid=d(x), mask=d(mask), flags=runtime.EP_MASK_ACC if masked else 0, out_dtype=torch.float32)
    two, hn_two = runtime.ffn_prenorm2_split(x1k, d(g), d(b), d(w1), w2c, d(mask), splits, next_norm=nn_)
    e2 = (y - two).abs()
    print(f"split + projection R={R} x{splits}: vs to_out GEMM + split path max {e2.max().item():.3e}, rms {e2.pow(2).mean().sqrt().item():.3e}")
    assert e2.max().item() <= 3e-2 and e2.pow(2).mean().sqrt().item() <= 2e-3
    x64 = x.double()
    pr = o.double() @ wo.double().t()
    x1 = x64 + (pr * mask[:, None] if masked else pr)
    h64 = (x1 - x1.mean(1, keepdim=True)) / torch.sqrt(x1.var(1, unbiased=False, keepdim=True) + 1e-5) * g.double() + b.double()
    ref64 = x1 + F.gelu(h64 @ w1.double().t()) @ w2.double().t()
    if masked:
        ref64 = ref64 * mask[:, None]
        assert y.cpu()[~mask].abs().max().item() == 0.0
    err = (y.cpu().double() - ref64).abs()
    assert err.max().item() <= 0.06 and err.pow(2).mean().sqrt().item() <= 6e-3
    y64 = y.cpu().double()
    ln64 = (y64 - y64.mean(1, keepdim=True)) / torch.sqrt(y64.var(1, unbiased=False, keepdim=True) + 1e-5) * ng.double() + nb.double()
    assert (hn.cpu().double() - ln64).abs().max().item() <= 2.0 ** -7 * max(1.0, ln64.abs().max().item())   # bf16 rounding


@pytest.mark.parametrize("R,D,Fi,masked", [(128 * 9 + 17, 384, 1536, True), (300, 256, 1024, True), (128 * 3, 384, 1536, False)])
def test_fused_ffn_with_layernorm_prologue(R, D, Fi, masked):
    """ispk_ffn_bf16_prenorm == LayerNorm kernel (bf16 out, row mask) -> ispk_ffn_bf16 with the fp32 rows as residual:
    two-pass fp32 statistics on the same values (another summation order), so the normalised operand differs by 1-ulp
    bf16 flips only; masked rows are exactly zero; the optional statistics describe the output rows."""
    x = synth._normal(f"t/ffnpre/x{R}", (R, D), 1.5, 0.4)
    w1, w2 = _bf(synth._normal(f"t/ffnpre/w1{D}", (Fi, D), D ** -0.5)), _bf(synth._normal(f"t/ffnpre/w2{D}", (D, Fi), Fi ** -0.5))
    g, b = synth._normal("t/ffnpre/g", (D,), 0.1, 1.0), synth._normal("t/ffnpre/b", (D,), 0.1)
    mask = (torch.arange(R) % 7 != 3) if masked else None
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    w2p = runtime.ffn_pack_w2(d(w2))
    fl = runtime.EP_MASK_OUT if masked else 0
    h = runtime.layernorm(d(x), d(g), d(b), row_mask=d(mask), out_dtype=torch.bfloat16)
    ref = runtime.ffn_fused(h, d(w1), w2p, resid=d(x), mask=d(mask), flags=fl).cpu()
    out = runtime.ffn_prenorm(d(x), d(g), d(b), d(w1), w2p, mask=d(mask), flags=fl)
    out2, stats = runtime.ffn_prenorm(d(x), d(g), d(b), d(w1), w2p, mask=d(mask), flags=fl, want_stats=True)
    assert torch.equal(out, out2)                       # deterministic, and the statistics epilogue leaves `out` alone
    assert torch.equal(out, runtime.ffn_prenorm(d(x), d(g), d(b), d(w1), w2p, mask=d(mask), flags=fl))
    out = out.cpu()
    err = (out - ref).abs()
    assert err.max().item() <= 2e-2 and err.pow(2).mean().sqrt().item() <= 1e-3
    if masked:
        assert out[~mask].abs().max().item() == 0.0
    # against float64 on the un-rounded LayerNorm: bounded by the bf16 operand / hidden roundings
    x64 = x.double()
    hn = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5) * g.double() + b.double()
    ref64 = x64 + F.gelu(hn @ w1.double().t()) @ w2.double().t()
    if masked:
        ref64 = ref64 * mask[:, None]
    assert (out.double() - ref64).abs().max().item() <= 0.06
    o64 = out.double()
    assert (stats[:, 0].cpu().double() - o64.mean(1)).abs().max().item() <= 1e-6
    rstd = 1.0 / torch.sqrt(o64.var(1, unbiased=False) + 1e-5)
    assert ((stats[:, 1].cpu().double() - rstd) / rstd).abs().max().item() <= 1e-5




def test_attention_bf16_several_query_tiles_per_workgroup():
    """With the whole key range resident in LDS a workgroup serves several 64-query tiles off one K/V fetch (the launcher
    does this by itself only for large batches; forced here).  Same values as one tile per workgroup, bit for bit."""
    B, N, H = 3, 450, 6
    qkv = _bf(synth._normal("t/at/qpw", (B, N, H * 64 + 128))).to(DEV)
    slopes = torch.tensor(synth.alibi_default_slopes(H), device=DEV)
    key_len = torch.tensor([450, 123, 300], device=DEV)
    one = runtime.alibi_mqa_attention(qkv, H, slopes, key_len, q_tiles=1).cpu()
    for qpw in (2, 4, 0):   # 0 = the launcher's own choice
        got = runtime.alibi_mqa_attention(qkv, H, slopes, key_len, q_tiles=qpw).cpu()
        assert torch.equal(got.view(torch.int16), one.view(torch.int16))


def test_layernorm_bf16_output_and_cast():
    x = synth._normal("t/ln/xb", (5, 33, 384), 2.0, 0.5)
    g, b = synth._normal("t/ln/g", (384,), 0.1, 1.0), synth._normal("t/ln/b", (384,), 0.1)
    ref = F.layer_norm(x, (384,), g, b, 1e-5).to(torch.bfloat16)
    out = runtime.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), out_dtype=torch.bfloat16).cpu()
    assert out.dtype == torch.bfloat16 and (out.float() - ref.float()).abs().max() <= 2 ** -6
    assert torch.equal(runtime.cast_bf16(x.to(DEV)).cpu(), x.to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------ aligner front-end
def test_flow_matching_algebra_kernels():
    """ispk_flow_mix_f32 / ispk_flow_finish_f32 against the PyTorch expressions of the reference's
    FlowTransformerTemporalModule.forward (temporal_adaptor.py:120-147) and utils.masked_mean: the mix is bit-exact (same
    operations, one rounding each), the loss agrees to fp32 summation-order noise."""
    from isp_tts_amd.utils import masked_mean
    B, L, C, sigma = 5, 37, 3, 1e-5
    x0, x1 = synth._normal("t/flow/x0", (B, L, C)), synth._normal("t/flow/x1", (B, L, C), 2.0, 0.3)
    t = synth._normal("t/flow/t", (B,)).abs().clamp(max=1.0)
    raw = synth._normal("t/flow/raw", (B, L, C))
    mask = torch.arange(L)[None] < torch.tensor([L, 1, 20, L - 1, 7])[:, None]
    tt = t[:, None, None]
    xt_ref = (1 - (1 - sigma) * tt) * x0 + tt * x1
    flow_ref = x1 - (1 - sigma) * x0
    xt, flow = runtime.flow_mix(x0.to(DEV), x1.to(DEV), t.to(DEV), sigma)
    assert torch.equal(xt.cpu(), xt_ref) and torch.equal(flow.cpu(), flow_ref)
    m3 = mask[..., None].expand(-1, -1, C)
    pf = raw * m3
    pred_ref = (x0 + pf) * m3
    loss_ref = masked_mean(F.mse_loss(pf, flow_ref, reduction="none"), m3)
    pred, dur, ratio, loss = runtime.flow_finish(raw.to(DEV), flow, x0.to(DEV), mask.to(DEV))
    assert torch.equal(pred.cpu(), pred_ref)
    assert (dur.cpu() - torch.clamp(torch.exp(pred_ref[..., 0]) - 1, min=0)).abs().max() < 1e-5
    assert abs(float(ratio.mean()) - float(loss_ref)) < 1e-5 * max(1.0, float(loss_ref))
    assert loss.shape == () and abs(float(loss) - float(loss_ref)) < 1e-5 * max(1.0, float(loss_ref))
    # a batch larger than the kernel's 16 waves (every wave takes several utterances)
    B2 = 70
    x0b, rawb = synth._normal("t/flow/x0b", (B2, L, C)), synth._normal("t/flow/rawb", (B2, L, C))
    flb = synth._normal("t/flow/flb", (B2, L, C))
    mb = torch.arange(L)[None] < (torch.arange(B2) % L + 1)[:, None]
    m3b = mb[..., None].expand(-1, -1, C)
    ref2 = masked_mean(F.mse_loss(rawb * m3b, flb, reduction="none"), m3b)
    _, _, ratio2, loss2 = runtime.flow_finish(rawb.to(DEV), flb.to(DEV), x0b.to(DEV), mb.to(DEV))
    assert abs(float(loss2) - float(ref2)) < 1e-5 * max(1.0, float(ref2)) and ratio2.shape == (B2,)


# ------------------------------------------------------------------------------------------------ between the stacks
def test_embed_tokens_is_the_embedding_lookup_plus_mask():
    """ispk_embed_tokens_f32 == F.embedding (a row copy: bit-exact) and arange(L) < text_len."""
    V, D, B, L = 149, 384, 5, 77
    table = synth._normal("t/emb/table", (V, D))
    table[0].zero_()
    g = synth._rng("t/emb/ids")
    text = torch.from_numpy(g.integers(0, V, size=(B, L)).astype(np.int64))
    lens = torch.tensor([77, 1, 40, 76, 13])
    emb, mask = runtime.embed_tokens(text.to(DEV), table.to(DEV), lens.to(DEV))
    assert torch.equal(emb.cpu(), F.embedding(text, table, padding_idx=0))
    assert mask.dtype == torch.bool and torch.equal(mask.cpu(), torch.arange(L)[None] < lens[:, None])
    emb2, none = runtime.embed_tokens(text.to(DEV), table.to(DEV), None, want_mask=False)
    assert none is None and torch.equal(emb2, emb)


def test_time_embedding_against_the_oracle(state_dict):
    """ispk_time_embedding_f32 (sinusoid with the raw step, Linear + SiLU, Linear) vs oracle.time_embedding."""
    p = "temporal_adaptor.predictor.time_embedding"
    t = torch.cat([synth._normal("t/temb/t", (64,)).abs().clamp(max=1.0), torch.tensor([0.0, 1.0, 0.36572])])
    ref = orc.time_embedding(state_dict, t)
    inv_freq = 1000.0 ** -(torch.arange(32).float() / 32)
    d = lambda k: state_dict[f"{p}.{k}"].to(DEV)          # noqa: E731
    out = runtime.time_embedding(t.to(DEV), inv_freq.to(DEV), d("freq_emb.freq_scale"), d("mlp.0.weight"), d("mlp.0.bias"),
                                 d("mlp.2.weight"), d("mlp.2.bias"))
    assert out.shape == (67, 32)
    # sin / cos of arguments up to 1000 rad: fp32 argument reduction differs by an ulp or two between libms
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    out11 = runtime.time_embedding(t[:1].view(1, 1).to(DEV), inv_freq.to(DEV), d("freq_emb.freq_scale"), d("mlp.0.weight"),
                                   d("mlp.0.bias"), d("mlp.2.weight"), d("mlp.2.bias"))
    assert out11.shape == (1, 1, 32) and torch.equal(out11.view(-1), out[0])


@pytest.mark.parametrize("B,M,L,D", [(3, 512, 100, 384), (2, 390, 73, 384), (4, 65, 17, 256), (1, 1024, 200, 384)])
def test_length_regulate_with_an_alignment(B, M, L, D):
    """ispk_length_regulate_f32 (forward mode): alignment @ x with exact-fp32 MFMA products vs float64, the decoder
    lengths (sum of the int64 MAS durations, clamped) and the decoder mask (temporal_adaptor.py:419-434)."""
    x = synth._normal(f"t/lr/x{M}", (B, L, D))
    a = torch.softmax(synth._normal(f"t/lr/a{M}", (B, M, L), 3.0), dim=-1)
    g = synth._rng(f"t/lr/d{M}")
    dur = torch.from_numpy(g.integers(0, 2 * M // L + 2, size=(B, L)).astype(np.int64))
    out, dec, mask = runtime.length_regulate(x.to(DEV), dur.to(DEV), a.to(DEV), M, max_len=M)
    ref = torch.bmm(a.double(), x.double())
    assert out.shape == (B, M, D) and (out.cpu().double() - ref).abs().max().item() < 2e-5
    want = torch.clamp((dur.sum(1) + 0.5).long(), max=M)
    assert torch.equal(dec.cpu(), want) and torch.equal(mask.cpu(), torch.arange(M)[None] < want[:, None])
    # no clamp when max_len is not given
    _, dec2, _ = runtime.length_regulate(x.to(DEV), dur.to(DEV), a.to(DEV), M)
    assert torch.equal(dec2.cpu(), (dur.sum(1) + 0.5).long())
    # the durations are only summed: a [B, 1] array with the same row sums gives the same lengths (what the teacher-forced
    # forward passes - mel_len - so that the decoder path does not wait for MAS)
    out3, dec3, mask3 = runtime.length_regulate(x.to(DEV), dur.sum(1, keepdim=True).to(DEV), a.to(DEV), M, max_len=M)
    assert torch.equal(dec3, dec) and torch.equal(out3, out) and torch.equal(mask3, mask)


@pytest.mark.parametrize("B,M,L,D", [(3, 512, 100, 384), (2, 301, 37, 256)])
def test_length_regulate_split_bf16(B, M, L, D):
    """ispk_length_regulate_split_bf16 (the bf16 compute path): every product as three bf16 MFMAs over hi / lo splits of the
    fp32 operands - against float64 to 1e-4 (2^-16 relative per product; the exact-fp32 kernel holds 2e-5), the same
    lengths and mask as the fp32 kernel, and the in-kernel soft path too."""
    x = synth._normal(f"t/lrs/x{M}", (B, L, D))
    a = torch.softmax(synth._normal(f"t/lrs/a{M}", (B, M, L), 3.0), dim=-1)
    dur = torch.full((B, 1), M, dtype=torch.int64)
    out, dec, mask = runtime.length_regulate(x.to(DEV), dur.to(DEV), a.to(DEV), M, max_len=M, split_bf16=True)
    ref = torch.bmm(a.double(), x.double())
    assert (out.cpu().double() - ref).abs().max().item() < 1e-4
    out32, dec32, mask32 = runtime.length_regulate(x.to(DEV), dur.to(DEV), a.to(DEV), M, max_len=M)
    assert torch.equal(dec, dec32) and torch.equal(mask, mask32)
    durf = synth._normal(f"t/lrs/d{M}", (B, L)).abs() * (1.8 * M / L)
    o_s, d_s, _ = runtime.length_regulate(x.to(DEV), durf.to(DEV), None, M, split_bf16=True)
    o_f, d_f, _ = runtime.length_regulate(x.to(DEV), durf.to(DEV), None, M)
    assert torch.equal(d_s, d_f) and (o_s - o_f).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,M,L,D", [(3, 512, 100, 384), (2, 301, 37, 256)])
def test_length_regulate_split_f16(B, M, L, D):
    """ispk_length_regulate_split_f16 (the split-fp16 parity path): three fp16 MFMAs per product over 22-bit operands - held to
    the exact-fp32 kernel's own bound against float64 (2e-5), same lengths and mask, and the in-kernel soft path."""
    x = synth._normal(f"t/lrs/x{M}", (B, L, D))
    a = torch.softmax(synth._normal(f"t/lrs/a{M}", (B, M, L), 3.0), dim=-1)
    dur = torch.full((B, 1), M, dtype=torch.int64)
    out, dec, mask = runtime.length_regulate(x.to(DEV), dur.to(DEV), a.to(DEV), M, max_len=M, split_bf16="f16")
    ref = torch.bmm(a.double(), x.double())
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5
    out32, dec32, mask32 = runtime.length_regulate(x.to(DEV), dur.to(DEV), a.to(DEV), M, max_len=M)
    assert torch.equal(dec, dec32) and torch.equal(mask, mask32)
    durf = synth._normal(f"t/lrs/d{M}", (B, L)).abs() * (1.8 * M / L)
    o_s, d_s, _ = runtime.length_regulate(x.to(DEV), durf.to(DEV), None, M, split_bf16="f16")
    o_f, d_f, _ = runtime.length_regulate(x.to(DEV), durf.to(DEV), None, M)
    assert torch.equal(d_s, d_f) and (o_s - o_f).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,L,M,D", [(3, 100, 512, 384), (2, 37, 300, 384), (1, 50, 130, 256)])
def test_length_regulate_from_a_soft_path(B, L, M, D):
    """ispk_length_regulate_f32 (infer mode): the soft path generated inside the kernel from fractional durations vs
    the oracle's generate_soft_path + matmul (temporal_adaptor.py:388-397, :468-478), masked by token and frame lengths."""
    x = synth._normal(f"t/sp/x{M}", (B, L, D))
    dur = synth._normal(f"t/sp/d{M}", (B, L)).abs() * (1.8 * M / L)
    enc_len = torch.tensor([L, max(1, L // 2), L - 3][:B])
    dur = dur * (torch.arange(L)[None] < enc_len[:, None])
    dec_ref = (dur.sum(1) + 0.5).long()
    m3 = ((torch.arange(L)[None] < enc_len[:, None]).unsqueeze(2) & (torch.arange(M)[None] < dec_ref[:, None]).unsqueeze(1))
    path = orc.generate_soft_path(dur, m3.float()).transpose(1, 2)                  # [B, M, L]
    ref = torch.bmm(path.double(), x.double())
    # The path weights are DIFFERENCES of clamped (cumsum - frame) ramps: an ulp of the fp32 cumulative sums (6e-5 at 700
    # frames) moves a weight by as much, so implementations that add in a different order agree only to ~1e-4 * |x| - the
    # reference's own arithmetic is that ill-conditioned.  Pin the kernel tightly against the SAME formula with the
    # cumulative sums taken sequentially in fp32 (numpy's cumsum; the kernel's order) ...
    cum = torch.from_numpy(np.cumsum(dur.numpy(), axis=1, dtype=np.float32))
    ramp = (cum.unsqueeze(2) - torch.arange(M, dtype=torch.float32)).clamp(0., 1.)            # [B, L, M]
    seq = ((ramp - F.pad(ramp, [0, 0, 1, 0])[:, :-1]) * m3.float()).transpose(1, 2)
    ref_seq = torch.bmm(seq.double(), x.double())
    out, dec, mask = runtime.length_regulate(x.to(DEV), dur.to(DEV), None, M, enc_len=enc_len.to(DEV))
    assert torch.equal(dec.cpu(), dec_ref)
    assert (out.cpu().double() - ref_seq).abs().max().item() < 2e-5
    # ... and against the oracle's (torch.cumsum's summation order) at the conditioning of the formula
    assert (out.cpu().double() - ref).abs().max().item() < 2e-3
    assert torch.equal(mask.cpu(), torch.arange(M)[None] < dec_ref[:, None])
    # B = 1 without masks (infer of a single utterance, model.py:191-201): enc_len None = every token
    out1, dec1, _ = runtime.length_regulate(x[:1].to(DEV), dur[:1].to(DEV), None, M)
    assert torch.equal(dec1.cpu(), dec_ref[:1]) and (out1.cpu().double() - ref_seq[:1]).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,T,C", [(3, 203, 80), (2, 512, 384), (4, 33, 5)])
@pytest.mark.parametrize("channel_first", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pad_rows(B, T, C, channel_first, dtype):
    """ispk_pad_rows_f32: masked, zero-padded channel-last [B, T+4, C] from [B,T,C] or channel-first [B,C,T] input (the
    latter through the tile-transposing kernel) - exact (a copy, plus bf16 rounding)."""
    x = synth._normal(f"t/pad/x{T}", (B, C, T) if channel_first else (B, T, C))
    lens = torch.tensor([T, max(1, T // 3), 1, T - 1][:B])
    out = runtime.pad_rows(x.to(DEV), lens.to(DEV), channel_first=channel_first, out_dtype=dtype).cpu()
    xt = x.transpose(1, 2) if channel_first else x
    ref = torch.zeros(B, T + 4, C)
    for b in range(B):
        ref[b, 2:2 + int(lens[b])] = xt[b, :int(lens[b])]
    assert out.shape == (B, T + 4, C) and out.dtype == dtype
    assert torch.equal(out.float(), ref.to(dtype).float())


@pytest.mark.parametrize("B,M,L", [(2, 512, 100), (3, 203, 37), (2, 64, 9), (1, 700, 300)])
def test_conv_attention_front_end_matches_oracle(state_dict, B, M, L):
    """Product ConvAttention (pad -> conv-as-GEMM -> masked instance norm -> fused scores) vs the oracle's
    conv_attention (alignment.py:159-208 restated), variable lengths, L not a multiple of 4 included."""
    from isp_tts_amd.acoustic import Aligner
    from isp_tts_amd.config import AcousticDims
    al = Aligner.init(AcousticDims().model_config()["aligner"], mel_dim=80, text_dim=384)
    sd = {k[len("aligner."):]: v for k, v in state_dict.items() if k.startswith("aligner.")}
    al.load_state_dict(sd, strict=True)
    al = al.to(DEV)
    inp = synth.make_inputs(B, L, M, variable=True, seed=5)
    enc = synth._normal(f"t/al/enc{L}", (B, L, 384))
    soft_ref, logits_ref = orc.conv_attention(state_dict, inp["mel"], enc.transpose(1, 2), inp["mel_len"], inp["text_len"])
    soft, logits = al.attention(inp["mel"].to(DEV), enc.to(DEV).transpose(1, 2), inp["mel_len"].to(DEV),
                                inp["text_len"].to(DEV))
    assert (soft.cpu() - soft_ref).abs().max() < 2e-5
    assert (logits.cpu() - logits_ref).abs().max() < 2e-4       # logits reach -35; relative 1e-5


@pytest.mark.parametrize("B,M,L", [(2, 512, 100), (3, 203, 37), (1, 300, 290)])
def test_aligner_scores_fast_mode(B, M, L):
    """ispk_aligner_scores_fast_f32 (bf16 compute path): split-bf16 score products and hardware exp / log against the exact
    kernel on the same inputs - logits to 1e-3 absolute (they reach -35: 3e-5 relative), soft attention to 1e-4."""
    q = synth._normal(f"t/asf/q{M}", (B, M + 4, 128), 2.0).to(DEV)
    k = synth._normal(f"t/asf/k{L}", (B, L + 4, 128), 2.0).to(DEV)
    text_len = torch.tensor([L, max(1, L // 2), L - 3][:B], device=DEV)
    mel_len = torch.tensor([M, max(1, M - 40), M // 2][:B], device=DEV)
    soft, logits = runtime.aligner_scores(q, k, text_len, mel_len, M, L)
    soft_f, logits_f = runtime.aligner_scores(q, k, text_len, mel_len, M, L, fast=True)
    fin = torch.isfinite(logits)
    assert torch.equal(fin, torch.isfinite(logits_f))
    assert (logits_f[fin] - logits[fin]).abs().max().item() < 1e-3
    assert (soft_f - soft).abs().max().item() < 1e-4


def test_soft_average_targets():
    B, M, L = 3, 130, 41
    g = synth._rng("t/avg")
    attn = torch.from_numpy(g.random((B, M, L)).astype(np.float32))
    pitch = torch.from_numpy(g.standard_normal((B, M)).astype(np.float32))
    energy = torch.from_numpy(g.standard_normal((B, M)).astype(np.float32))
    dur = torch.from_numpy(g.integers(0, 9, size=(B, L)).astype(np.int64))
    tl = torch.tensor([41, 7, 20])
    mask = (torch.arange(L)[None] < tl[:, None])[..., None]
    want = torch.cat([torch.log1p(dur)[..., None], orc.soft_average(pitch[:, None], attn).transpose(1, 2) * mask,
                      orc.soft_average(energy[:, None], attn).transpose(1, 2) * mask], dim=-1)
    got = runtime.soft_average(attn.to(DEV), pitch.to(DEV), energy.to(DEV), dur.to(DEV), tl.to(DEV)).cpu()
    assert (got - want).abs().max() < 2e-6






@pytest.mark.parametrize("R,D,Fi", [(777, 384, 1536), (16500, 384, 1536), (300, 256, 1024)])
def test_fused_ffn_bf16(R, D, Fi):
    """ispk_ffn_bf16 vs float64 FFN on the same bf16-rounded operands, with the hidden rounded to bf16 as the kernel
    does (it feeds the second MFMA product as bf16)."""
    x = _bf(synth._normal(f"t/ffn/x{R}", (R, D)))
    w1, w2 = _bf(synth._normal(f"t/ffn/w1{D}", (Fi, D), D ** -0.5)), _bf(synth._normal(f"t/ffn/w2{D}", (D, Fi), Fi ** -0.5))
    resid = synth._normal("t/ffn/r", (R, D))
    mask = torch.arange(R) % 5 != 2
    hid = F.gelu(x.double() @ w1.double().T).to(torch.bfloat16).double()
    ref = (resid.double() + hid @ w2.double().T) * mask[:, None]
    d = lambda t: t.to(DEV)  # noqa: E731
    out = runtime.ffn_fused(d(x), d(w1), d(w2), resid=d(resid), mask=d(mask), flags=runtime.EP_MASK_OUT).cpu()
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-2, err          # bf16 rounding of the hidden: a 1-ulp flip of a hidden of size ~3 moves an output by ~1e-3
    assert (out.double() - ref).pow(2).mean().sqrt() < 1e-3
    b1, b2 = synth._normal("t/ffn/b1", (Fi,), 0.1), synth._normal("t/ffn/b2", (D,), 0.1)
    out = runtime.ffn_fused(d(x), d(w1), d(w2), bias1=d(b1), bias2=d(b2)).cpu()
    ref = F.gelu(x.double() @ w1.double().T + b1.double()).to(torch.bfloat16).double() @ w2.double().T + b2.double()
    assert (out.double() - ref).pow(2).mean().sqrt() < 1e-3 and (out.double() - ref).abs().max() < 2e-2
    # the packed-W2 image computes the same products in the same order: bit-identical to the row-major path
    w2p = runtime.ffn_pack_w2(d(w2))
    pos = torch.arange(32)
    hid_of_pos = (pos & 16) | ((pos & 4) << 1) | ((pos & 8) >> 1) | (pos & 3)
    expect = w2.view(D, Fi // 32, 32)[:, :, hid_of_pos].permute(1, 0, 2).contiguous()
    assert torch.equal(w2p.cpu().view(torch.int16), expect.view(torch.int16))
    out_p = runtime.ffn_fused(d(x), d(w1), w2p, bias1=d(b1), bias2=d(b2)).cpu()
    assert torch.equal(out_p, out)
