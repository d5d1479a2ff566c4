#!/usr/bin/env python3
"""Micro-benchmark of the hot kernels at the BASELINE config-3 shapes (B=64: 32768 decoder rows, 6400 encoder rows).
HIP-event timing, interleaved rounds, median.  Usage: python tools/bench_kernels.py [f32|bf16] [--rows 32768]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
R = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 32768
dev = "cuda"


def time_it(fn, rounds=7, inner=20):
    """Median over rounds of (time of `inner` back-to-back launches) / inner: excludes the ~15 us idle-launch +
    event floor that a single timed launch carries."""
    for _ in range(3):
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
    return ts[len(ts) // 2], ts[0]


x384 = synth._normal("b/x", (R, 384)).to(dev).to(dt)
x1536 = synth._normal("b/x2", (R, 1536)).to(dev).to(dt)
resid = synth._normal("b/r", (R, 384)).to(dev)
mask = torch.ones(R, dtype=torch.bool, device=dev)
w = {n: synth._normal(f"b/w{n}", s, s[1] ** -0.5).to(dev).to(dt) for n, s in
     {"qkv": (512, 384), "o": (384, 384), "f1": (1536, 384), "f2": (384, 1536)}.items()}
odt = dt
cases = [
    ("qkv   [R,384]x[512,384]", lambda: runtime.gemm(x384, w["qkv"]), 2 * R * 512 * 384, R * (384 + 512) * x384.element_size()),
    ("out   [R,384]x[384,384]+res", lambda: runtime.gemm(x384, w["o"], resid=resid, mask=mask, flags=runtime.EP_MASK_ACC,
                                                         out_dtype=torch.float32), 2 * R * 384 * 384, R * 384 * (x384.element_size() + 8)),
    ("ffn1  [R,384]x[1536,384]+gelu", lambda: runtime.gemm(x384, w["f1"], flags=runtime.EP_GELU), 2 * R * 1536 * 384,
     R * (384 + 1536) * x384.element_size()),
    ("ffn2  [R,1536]x[384,1536]+res", lambda: runtime.gemm(x1536, w["f2"], resid=resid, mask=mask, flags=runtime.EP_MASK_OUT,
                                                           out_dtype=torch.float32), 2 * R * 384 * 1536,
     R * (1536 * x384.element_size() + 384 * 8)),
    ("layernorm [R,384]", lambda: runtime.layernorm(resid, None, None, out_dtype=dt), 0, R * 384 * (4 + x384.element_size())),
]
B, N = R // 512, 512
qkv = synth._normal("b/qkv", (B, N, 512)).to(dev).to(dt)
slopes = torch.tensor(synth.alibi_default_slopes(6), device=dev)
cases.append((f"attn  B={B} N={N} H=6", lambda: runtime.alibi_mqa_attention(qkv, 6, slopes, None), 256 * B * N * N * 6,
              B * N * (2 * 384 + 128) * qkv.element_size()))
print(f"dtype={dt} rows={R}")
for name, fn, flops, nbytes in cases:
    med, mn = time_it(fn)
    print(f"{name:34s} median {med:8.1f} us  min {mn:8.1f} us  {flops / med / 1e6:8.1f} TFLOP/s  {nbytes / med / 1e3:8.1f} GB/s")
if dt == torch.bfloat16:
    med, mn = time_it(lambda: runtime.ffn_fused(x384, w["f1"], w["f2"], resid=resid, mask=mask, flags=runtime.EP_MASK_OUT))
    fl = 4 * R * 384 * 1536
    print(f"{'fused ffn [R,384]->1536->384+res':34s} median {med:8.1f} us  min {mn:8.1f} us  {fl / med / 1e6:8.1f} TFLOP/s")
