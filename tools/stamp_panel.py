#!/usr/bin/env python3
"""In-kernel phase stamps of the panel GEMM (ISPK_PANEL_STAMP): per-wave cycle sums per phase."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP      # stamps exist in the experiments build only
R, dev, dt = 32768, "cuda", torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
flags = runtime.EP_GELU if N == 1536 else 0
x = synth._normal("b/x", (R, 384)).to(dev).to(dt)
w = synth._normal("b/w", (N, 384), 384 ** -0.5).to(dev).to(dt)
nwg = 8 * 4096
dbg = torch.zeros(nwg * 4, 6, dtype=torch.int64, device=dev)
for _ in range(3):
    runtime.gemm(x, w, flags=flags)
os.environ["ISPK_PANEL_STAMP"] = hex(dbg.data_ptr())
runtime.gemm(x, w, flags=flags)
torch.cuda.synchronize()
d = dbg.cpu().double()
d = d[d.sum(1) > 0]
print(f"N={N}: {len(d)} waves; total cycles per wave (s_memtime units), mean / min / max:")
names = ["prologue", "half 0 MFMAs", "barrier1+wstore", "half 1 MFMAs", "barrier2+wstore", "epilogue"]
for i, n in enumerate(names):
    print(f"  {n:16s} {d[:, i].mean():9.0f} {d[:, i].min():9.0f} {d[:, i].max():9.0f}")
print("  total            %9.0f" % d.sum(1).mean())
