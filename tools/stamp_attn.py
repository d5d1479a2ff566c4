#!/usr/bin/env python3
"""In-kernel phase stamps of the bf16 attention kernel (ISPK_ATTN_STAMP)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP      # stamps exist in the experiments build only
B, T, H, dev = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 6, "cuda"
qkv = (synth._normal("b/qkv", (B, T, 512)) * 2.0).to(dev).to(torch.bfloat16)
slopes = torch.tensor(synth.alibi_default_slopes(H), device=dev)
nw = B * ((T + 63) // 64) * 2 * H
dbg = torch.zeros(nw, 6, dtype=torch.int64, device=dev)
for _ in range(3):
    runtime.alibi_mqa_attention(qkv, H, slopes, None)
os.environ["ISPK_ATTN_STAMP"] = hex(dbg.data_ptr())
runtime.alibi_mqa_attention(qkv, H, slopes, None)
torch.cuda.synchronize()
d = dbg.cpu().double()
names = ["prologue", "wait+barrier", "init+QK", "softmax", "PV", "TOTAL (kernel)"]
print(f"T={T}: cycles per wave, mean / min / max over {nw} waves")
for i, n in enumerate(names):
    print(f"  {n:16s} {d[:, i].mean():9.0f} {d[:, i].min():9.0f} {d[:, i].max():9.0f}")
