#!/usr/bin/env python3
"""Graph-replay timing of single kernels at the benchmark shapes (20 launches per replay, 7 rounds, min / median)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
if os.environ.get("BENCH_LIB"):      # A/B: time another build of the library (path relative to the package)
    runtime.LIB_PATH = os.path.join(os.path.dirname(runtime.__file__), os.environ["BENCH_LIB"])
R, dev, dt = 32768, "cuda", torch.bfloat16
x = synth._normal("b/x", (R, 384)).to(dev).to(dt)
x1536 = synth._normal("b/x2", (R, 1536)).to(dev).to(dt)
resid = synth._normal("b/r", (R, 384)).to(dev)
mask = torch.ones(R, dtype=torch.bool, device=dev)
w1 = synth._normal("b/wf1", (1536, 384), 384 ** -0.5).to(dev).to(dt)
w2 = synth._normal("b/wf2", (384, 1536), 1536 ** -0.5).to(dev).to(dt)
wqkv = synth._normal("b/wqkv", (512, 384), 384 ** -0.5).to(dev).to(dt)
wo = synth._normal("b/wo", (384, 384), 384 ** -0.5).to(dev).to(dt)
w2p = runtime.ffn_pack_w2(w2)
qkv = (synth._normal("b/qkv", (64, 512, 512)) * 2.0).to(dev).to(dt)
qkv100 = (synth._normal("b/qkv100", (64, 100, 512)) * 2.0).to(dev).to(dt)
slopes = torch.tensor(synth.alibi_default_slopes(6), device=dev)
klen = torch.full((64,), 512, dtype=torch.int64, device=dev)
x6k = synth._normal("b/x6k", (6400, 1536)).to(dev).to(dt)
r6k = synth._normal("b/r6k", (6400, 384)).to(dev)
m6k = torch.ones(6400, dtype=torch.bool, device=dev)
x6k384 = synth._normal("b/x6k384", (6400, 384)).to(dev).to(dt)
cases = {
    "small panel 6400x384->1536 gelu": (lambda: runtime.gemm(x6k384, w1, flags=runtime.EP_GELU), 2.0 * 6400 * 384 * 1536),
    "small panel 6400x384->512": (lambda: runtime.gemm(x6k384, wqkv), 2.0 * 6400 * 384 * 512),
    "small panel 6400x384->384 f32+resid": (lambda: runtime.gemm(x6k384, wo, resid=r6k, mask=m6k, flags=runtime.EP_MASK_ACC, out_dtype=torch.float32), 2.0 * 6400 * 384 * 384),
    "wide 6400x1536->384 f32+resid": (lambda: runtime.gemm(x6k, w2, resid=r6k, mask=m6k, flags=runtime.EP_MASK_OUT, out_dtype=torch.float32), 2.0 * 6400 * 384 * 1536),
    "attn B64 T512 H6": (lambda: runtime.alibi_mqa_attention(qkv, 6, slopes, klen), 256.0 * 64 * 512 * 512 * 6),
    "attn B64 T100 H6": (lambda: runtime.alibi_mqa_attention(qkv100, 6, slopes, None), 256.0 * 64 * 100 * 100 * 6),
    "ffn_fused": (lambda: runtime.ffn_fused(x, w1, w2p, resid=resid, mask=mask, flags=runtime.EP_MASK_OUT), 4.0 * R * 384 * 1536),
    "panel 384->1536 gelu bf16": (lambda: runtime.gemm(x, w1, flags=runtime.EP_GELU), 2.0 * R * 384 * 1536),
    "panel 384->512 bf16": (lambda: runtime.gemm(x, wqkv), 2.0 * R * 384 * 512),
    "panel 384->384 f32+resid": (lambda: runtime.gemm(x, wo, resid=resid, mask=mask, flags=runtime.EP_MASK_OUT, out_dtype=torch.float32), 2.0 * R * 384 * 384),
    "wide 1536->384 f32+resid": (lambda: runtime.gemm(x1536, w2, resid=resid, mask=mask, flags=runtime.EP_MASK_OUT, out_dtype=torch.float32), 2.0 * R * 384 * 1536),
}
gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
stats32 = torch.stack([resid.mean(1), 1.0 / torch.sqrt(resid.var(1, unbiased=False) + 1e-5)], 1).contiguous()
cases.update({
    "lnin panel 384->512 (fp32 x + stats)": (lambda: runtime.gemm_lnin(resid, stats32, gam, bet, wqkv), 2.0 * R * 384 * 512),
    "lnin panel 384->512 (own stats)": (lambda: runtime.gemm_lnin(resid, None, gam, bet, wqkv), 2.0 * R * 384 * 512),
    "lnin small 6400x384->512 (own stats)": (lambda: runtime.gemm_lnin(r6k, None, gam, bet, wqkv), 2.0 * 6400 * 384 * 512),
    "lnin small 6400x384->1536 gelu (own)": (lambda: runtime.gemm_lnin(r6k, None, gam, bet, w1, flags=runtime.EP_GELU), 2.0 * 6400 * 384 * 1536),
    "layernorm 6400x384 bf16 out": (lambda: runtime.layernorm(r6k, gam, bet, out_dtype=dt), 0.0),
    "layernorm 384 bf16 out": (lambda: runtime.layernorm(resid, gam, bet, out_dtype=dt), 0.0),
    "ffn_prenorm (fp32 x, own LN)": (lambda: runtime.ffn_prenorm(resid, gam, bet, w1, w2p, mask=mask, flags=runtime.EP_MASK_OUT), 4.0 * R * 384 * 1536),
    "ffn_prenorm + stats": (lambda: runtime.ffn_prenorm(resid, gam, bet, w1, w2p, mask=mask, flags=runtime.EP_MASK_OUT, want_stats=True), 4.0 * R * 384 * 1536),
})
only = sys.argv[1:]
graphs = {}
for name, (fn, _) in cases.items():
    if only and not any(o in name for o in only):
        continue
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    graphs[name] = g
res = {k: [] for k in graphs}
for _ in range(7):
    for name, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for k, v in res.items():
    v.sort()
    print(f"{k:28s} min {v[0]:7.1f} us  median {v[len(v)//2]:7.1f} us   {cases[k][1]/v[0]*1e-6:6.0f} TF/s")
