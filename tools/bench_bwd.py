"""Times the backward kernels at the decoder's shapes (B = 64 x 512 frames, dim 384) with HIP events. usage: bench_bwd.py"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime
dev = "cuda"
M = 32768


def timeit(name, fn, flops, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s")


dy = torch.randn(M, 384, device=dev)
a = torch.randn(M, 1536, device=dev)
h = torch.randn(M, 384, device=dev)
dq = torch.randn(M, 512, device=dev)
mask = torch.ones(M, dtype=torch.bool, device=dev)
timeit("gemm_tn 384x1536 (dW2)", lambda: runtime.gemm_tn(dy, a, row_mask=mask), 2.0 * M * 384 * 1536)
timeit("gemm_tn 1536x384 (dW1)", lambda: runtime.gemm_tn(a, h), 2.0 * M * 384 * 1536)
timeit("gemm_tn 384x384 (dWo)", lambda: runtime.gemm_tn(dy, h, row_mask=mask), 2.0 * M * 384 * 384)
timeit("gemm_tn 512x384 (dWqkv)", lambda: runtime.gemm_tn(dq, h), 2.0 * M * 512 * 384)
w = torch.randn(1536, 384, device=dev)
timeit("gemm NT 32768x1536x384", lambda: runtime.gemm(h, w), 2.0 * M * 384 * 1536)
wt = torch.randn(384, 1536, device=dev)
timeit("gemm NT 32768x384x1536", lambda: runtime.gemm(a, wt), 2.0 * M * 384 * 1536)
B, N, H = 64, 512, 6
qkv = torch.randn(B, N, H * 64 + 128, device=dev) * 0.7
slopes = torch.tensor([2.0 ** (-(i + 1) * 8.0 / H) for i in range(H)], device=dev)
klen = torch.full((B,), N, dtype=torch.int64, device=dev)
o = runtime.alibi_mqa_attention(qkv, H, slopes, klen)
d_o = torch.randn_like(o)
timeit("attention fwd f32", lambda: runtime.alibi_mqa_attention(qkv, H, slopes, klen), 4.0 * B * H * N * N * 64)
timeit("attention bwd f32 (dq + dkv)", lambda: runtime.alibi_mqa_attention_bwd(qkv, o, d_o, H, slopes, klen), 10.0 * B * H * N * N * 64)
x = torch.randn(M, 384, device=dev)
g = torch.ones(384, device=dev)
timeit("layernorm bwd", lambda: runtime.layernorm_bwd(x, dy, g, row_mask=mask), 0.0)
u = torch.randn(M, 1536, device=dev)
timeit("gelu bwd", lambda: runtime.gelu_bwd(a, u), 0.0)
