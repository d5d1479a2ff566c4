import sys, torch
sys.path.insert(0, ".")
from isp_tts_amd import runtime, synth
from oracle import acoustic_oracle as orc
for B, N, H, ragged in [(2, 77, 4, False), (2, 77, 6, False), (2, 64, 4, False), (2, 100, 4, False), (2, 128, 4, True), (2, 130, 2, True), (2, 64, 1, False), (2, 200, 8, True), (1, 1000, 6, False)]:
    qkv = synth._normal(f"split/attn/{B}/{N}/{H}", (B, N, H * 64 + 128), 1.5)
    slopes = torch.tensor(synth.alibi_default_slopes(H))
    lens = torch.tensor([N, max(N // 3, 1), max(N - 5, 1)][:B]) if ragged else None
    out = runtime.alibi_mqa_attention_split(qkv.cuda(), H, slopes.cuda(), None if lens is None else lens.cuda(), out_split=False)
    torch.cuda.synchronize()
    q = qkv[..., :H * 64].view(B, N, H, 64).transpose(1, 2).double()
    k, v = qkv[..., H * 64:H * 64 + 64].double(), qkv[..., H * 64 + 64:].double()
    mask = None if lens is None else (torch.arange(N)[None] < lens[:, None])[:, None, None, :]
    bias = slopes.double().view(H, 1, 1) * orc.alibi_int_bias(N, N).double()
    ref = orc.attend(q, k, v, mask, bias).transpose(1, 2).reshape(B, N, H * 64)
    valid = torch.ones(B, N, dtype=torch.bool) if lens is None else torch.arange(N)[None] < lens[:, None]
    err = ((out.double().cpu() - ref).abs() * valid[..., None])
    e = err.view(B, N, H, 64).amax(dim=(0, 3))      # [N, H]
    bad = (e > 5e-6).nonzero()
    print(B, N, H, ragged, "max err %.3e" % err.max().item(), "nan" if torch.isnan(out).any() else "", "bad (query, head):", bad[:6].tolist(), flush=True)
