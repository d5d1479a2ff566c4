#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
if os.environ.get("ISPK_MAS_ABLATE"):          # phase ablations (1: DP only, 2: DP + backtrack) live in the experiments build
    runtime.LIB_PATH = build.LIB_EXP
x, tl, ml = synth.make_mas_logits(64, 512, 100)
x, tl, ml = x.cuda(), tl.cuda(), ml.cuda()
for _ in range(3): runtime.mas(x, tl, ml)
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): runtime.mas(x, tl, ml)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
print(f"MAS B=64 M=512 L=100: {sorted(ts)[3]:.1f} us per launch")
