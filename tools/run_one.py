#!/usr/bin/env python3
"""Runs one kernel a few times (for rocprofv3 --pmc passes).  usage: run_one.py ffn|panel|wide|attn"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
which = sys.argv[1]
R, dev, dt = 32768, "cuda", torch.bfloat16
x384 = synth._normal("b/x", (R, 384)).to(dev).to(dt)
x1536 = synth._normal("b/x2", (R, 1536)).to(dev).to(dt)
resid = synth._normal("b/r", (R, 384)).to(dev)
mask = torch.ones(R, dtype=torch.bool, device=dev)
w1 = synth._normal("b/wf1", (1536, 384), 384 ** -0.5).to(dev).to(dt)
w2 = synth._normal("b/wf2", (384, 1536), 1536 ** -0.5).to(dev).to(dt)
for _ in range(4):
    if which == "ffn":
        runtime.ffn_fused(x384, w1, w2, resid=resid, mask=mask, flags=runtime.EP_MASK_OUT)
    elif which == "panel":
        runtime.gemm(x384, w1, flags=runtime.EP_GELU)
    elif which == "wide":
        runtime.gemm(x1536, w2, resid=resid, mask=mask, flags=runtime.EP_MASK_OUT, out_dtype=torch.float32)
    elif which == "attn":
        qkv = synth._normal("b/qkv", (64, 512, 512)).to(dev).to(dt)
        runtime.alibi_mqa_attention(qkv, 6, torch.tensor(synth.alibi_default_slopes(6), device=dev), None)
torch.cuda.synchronize()
