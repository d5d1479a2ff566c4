#!/usr/bin/env python3
"""Sweep of ispk_gemm_bf16 over N (and rows) at K=384 to separate fixed cost from per-tile cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth

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

K = int(os.environ.get("K", "384"))
for R in (4096, 8192, 16384, 32768, 65536):
    x = torch.randn(R, K, device=dev).to(torch.bfloat16)
    line = f"rows={R:6d} "
    for N in (64, 128, 256, 512, 1024, 1536):
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
        t, _ = time_it(lambda: runtime.gemm(x, w, out=out))
        line += f" N={N}:{t:6.1f}us"
    print(line)
# empty-kernel launch floor for reference
tiny = torch.empty(64, 64, device=dev)
t, _ = time_it(lambda: runtime.cast_bf16(tiny))
print(f"tiny kernel launch+event floor: {t:.1f} us")
