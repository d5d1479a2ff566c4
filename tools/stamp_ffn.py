#!/usr/bin/env python3
"""In-kernel phase stamps of the fused FFN (ISPK_FFN_STAMP): per-wave cycle sums of phase A / phase B / barrier."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
R, dev, dt = 32768, "cuda", torch.bfloat16
x = synth._normal("b/x", (R, 384)).to(dev).to(dt)
resid = synth._normal("b/r", (R, 384)).to(dev)
mask = torch.ones(R, dtype=torch.bool, device=dev)
w1 = synth._normal("b/wf1", (1536, 384), 384 ** -0.5).to(dev).to(dt)
w2p = runtime.ffn_pack_w2(synth._normal("b/wf2", (384, 1536), 1536 ** -0.5).to(dev).to(dt))
dbg = torch.zeros((R // 128) * 4, 5, dtype=torch.int64, device=dev)
for _ in range(3):
    runtime.ffn_fused(x, w1, w2p, resid=resid, mask=mask, flags=runtime.EP_MASK_OUT)
os.environ["ISPK_FFN_STAMP"] = hex(dbg.data_ptr())
runtime.ffn_fused(x, w1, w2p, resid=resid, mask=mask, flags=runtime.EP_MASK_OUT)
torch.cuda.synchronize()
d = dbg.cpu().double()
d[:, :3] /= 48.0   # per chunk
names = ["phase A (G1)", "phase B (G2+GELU)", "barrier", "PROLOGUE (total)", "EPILOGUE (total)"]
print("cycles per chunk (s_memtime units), mean / min / max over waves:")
for i, n in enumerate(names):
    print(f"  {n:14s} {d[:, i].mean():8.0f} {d[:, i].min():8.0f} {d[:, i].max():8.0f}")
print("  loop total     %8.0f" % (d[:, :3].sum(1).mean() * 48))
