#!/usr/bin/env python3
"""Micro-benchmark of the split-fp16 kernels at the BASELINE config-3 shapes beside the exact-fp32 kernels they replace.
usage: python tools/bench_split.py [--rows 32768]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth  # noqa: E402

R = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 32768
dev = "cuda"


def time_it(fn, rounds=5, inner=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


x384 = synth._normal("b/x", (R, 384)).to(dev)
x1536 = synth._normal("b/x2", (R, 1536)).to(dev)
resid = synth._normal("b/r", (R, 384)).to(dev)
mask = torch.ones(R, dtype=torch.bool, device=dev)
w = {n: synth._normal(f"b/w{n}", s, s[1] ** -0.5).to(dev) for n, s in
     {"qkv": (512, 384), "o": (384, 384), "f1": (1536, 384), "f2": (384, 1536)}.items()}
ws = {n: runtime.split_f16(t) for n, t in w.items()}
p384, p1536 = runtime.split_f16(x384), runtime.split_f16(x1536)
N = 512 if R % 512 == 0 else 100
B = R // N
qkv = synth._normal("b/qkv", (B, N, 512)).to(dev)
slopes = torch.tensor(synth.alibi_default_slopes(6), device=dev)
cases = [
    ("qkv  [R,384]x[512,384]", lambda: runtime.gemm(x384, w["qkv"]), lambda: runtime.gemm_split(p384, ws["qkv"]), 2 * R * 512 * 384),
    ("out  [R,384]x[384,384]+res", lambda: runtime.gemm(x384, w["o"], resid=resid, mask=mask, flags=runtime.EP_MASK_ACC),
     lambda: runtime.gemm_split(p384, ws["o"], resid=resid, mask=mask, flags=runtime.EP_MASK_ACC), 2 * R * 384 * 384),
    ("ffn1 [R,384]x[1536,384]+gelu", lambda: runtime.gemm(x384, w["f1"], flags=runtime.EP_GELU),
     lambda: runtime.gemm_split(p384, ws["f1"], flags=runtime.EP_GELU, out_split=True), 2 * R * 1536 * 384),
    ("ffn2 [R,1536]x[384,1536]+res", lambda: runtime.gemm(x1536, w["f2"], resid=resid, mask=mask, flags=runtime.EP_MASK_OUT),
     lambda: runtime.gemm_split(p1536, ws["f2"], resid=resid, mask=mask, flags=runtime.EP_MASK_OUT), 2 * R * 384 * 1536),
    ("layernorm [R,384]", lambda: runtime.layernorm(resid, None, None), lambda: runtime.layernorm_split(resid, None, None), 0),
    (f"attn B={B} N={N} H=6", lambda: runtime.alibi_mqa_attention(qkv, 6, slopes, None),
     lambda: runtime.alibi_mqa_attention_split(qkv, 6, slopes, None), 256 * B * N * N * 6),
]
print(f"rows={R}")
for name, f32, sp, flops in cases:
    a, b = time_it(f32), time_it(sp)
    print(f"{name:34s} fp32 {a:8.1f} us   split {b:8.1f} us   x{a / b:4.1f}   split = {flops / b / 1e6:7.1f} TF/s (fp32-equivalent FLOPs)")
