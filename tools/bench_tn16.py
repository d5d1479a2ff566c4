#!/usr/bin/env python3
"""dW = dY^T X with operands stored in bf16 (ispk_gemm_tn_b16) at the decoder's shapes; run under rocprofv3 --kernel-trace
--stats to separate the product kernel from the ordered sum of the row-range partials (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
R, dev = 32768, "cuda"
for n1, n2 in ((384, 1536), (1536, 384), (512, 384), (384, 384)):
    a, b = synth._normal(f"tn/a{n1}", (R, n1)).to(dev).to(torch.bfloat16), synth._normal(f"tn/b{n2}", (R, n2)).to(dev).to(torch.bfloat16)
    for _ in range(3):
        runtime.gemm_tn(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        runtime.gemm_tn(a, b)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"dW [{n1} x {n2}] over {R} rows, bf16-stored operands: {t:7.1f} us  {2.0 * R * n1 * n2 / t * 1e-6:6.1f} TF/s")
