#!/usr/bin/env python3
"""Times the weight-gradient product dW = dY^T X at the decoder's shapes: fp32 operands vs bf16 operands (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
R, dev = 32768, "cuda"
for n1, n2 in ((1536, 384), (384, 1536), (512, 384), (384, 384)):
    a, b = synth._normal(f"tn/a{n1}", (R, n1)).to(dev), synth._normal(f"tn/b{n2}", (R, n2)).to(dev)
    for bf in (False, True):
        for _ in range(3):
            runtime.gemm_tn(a, b, bf16=bf)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                runtime.gemm_tn(a, b, bf16=bf)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        t = sorted(ts)[2]
        print(f"dW [{n1} x {n2}] over {R} rows, {'bf16' if bf else 'fp32'} operands: {t:7.1f} us  {2.0 * R * n1 * n2 / t * 1e-6:6.1f} TF/s  "
              f"{4.0 * R * (n1 + n2) / t * 1e-6:5.2f} TB/s of operand bytes")
