#!/usr/bin/env python3
"""Training attention (csrc/attention_train.hip) at the decoder / encoder / adaptor shapes of the B = 64 step.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
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
    return sorted(ts)[len(ts) // 2]


for name, (B, N, H) in {"decoder": (64, 512, 6), "encoder": (64, 100, 6), "adaptor": (64, 100, 4)}.items():
    qkv = (synth._normal(f"bat/qkv/{N}/{H}", (B, N, H * 64 + 128)) * 1.0).to(dev).bfloat16()
    d_o = synth._normal(f"bat/do/{N}/{H}", (B, N, H * 64)).to(dev).bfloat16()
    slopes = torch.tensor(synth.alibi_default_slopes(H)).to(dev)
    klen = torch.full((B,), N, device=dev)
    for p in (0.0, 0.1):
        o, lse = runtime.alibi_mqa_attention_train(qkv, H, slopes, klen, p, 7)
        tf = time_it(lambda: runtime.alibi_mqa_attention_train(qkv, H, slopes, klen, p, 7))
        tb = time_it(lambda: runtime.alibi_mqa_attention_bwd(qkv, o, d_o, H, slopes, klen, lse=lse, dropout_p=p, seed=7))
        fl = 4.0 * B * H * N * N * 64
        print(f"{name:8s} B={B} N={N} H={H} p={p}: fwd {tf:7.1f} us ({fl / tf * 1e-6:6.1f} TFLOP/s)   bwd {tb:7.1f} us ({2.5 * fl / tb * 1e-6:6.1f} TFLOP/s)", flush=True)
