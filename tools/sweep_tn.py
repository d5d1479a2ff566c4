#!/usr/bin/env python3
"""LDS-DMA weight-gradient kernel: ring depth x workgroup target (experiments build: ISPK_TN_STAGES / ISPK_TN_TARGET).  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP
R, dev = 32768, "cuda"
for n1, n2 in ((384, 1536), (1536, 384), (512, 384), (384, 384)):
    a, b = synth._normal(f"tn/a{n1}", (R, n1)).to(dev).to(torch.bfloat16), synth._normal(f"tn/b{n2}", (R, n2)).to(dev).to(torch.bfloat16)
    row = []
    for stages in (3, 4):
        for target in (192, 256, 384, 512):
            os.environ["ISPK_TN_STAGES"], os.environ["ISPK_TN_TARGET"] = str(stages), str(target)
            for _ in range(3):
                runtime.gemm_tn(a, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                runtime.gemm_tn(a, b)
            e1.record(); torch.cuda.synchronize()
            row.append((stages, target, e0.elapsed_time(e1) / 20 * 1e3))
    print(f"dW [{n1} x {n2}]: " + "  ".join(f"s{s}/t{t}:{v:.0f}" for s, t, v in row), flush=True)
