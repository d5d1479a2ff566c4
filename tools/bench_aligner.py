#!/usr/bin/env python3
"""Graph-replay timing of the aligner-side kernels at the benchmark shape (B=64, M=512, L=100)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
B, M, L, dev = 64, 512, 100, "cuda"
attn = torch.softmax(synth._normal("ba/attn", (B, M, L)), -1).to(dev)
pitch, energy = synth._normal("ba/p", (B, M)).to(dev), synth._normal("ba/e", (B, M)).to(dev)
dur = torch.full((B, L), 5, dtype=torch.int64, device=dev)
tl = torch.full((B,), L, dtype=torch.int64, device=dev)
qe, ke = synth._normal("ba/qe", (B, M + 4, 128)).to(dev), synth._normal("ba/ke", (B, L + 4, 128)).to(dev)
mlen = torch.full((B,), M, dtype=torch.int64, device=dev)
cases = {"soft_average": lambda: runtime.soft_average(attn, pitch, energy, dur, tl),
         "aligner_scores": lambda: runtime.aligner_scores(qe, ke, tl, mlen, M, L)}
for name, fn in cases.items():
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ts.sort()
    print(f"{name:24s} min {ts[0]:7.1f} us  median {ts[3]:7.1f} us")
