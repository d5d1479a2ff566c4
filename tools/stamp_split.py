#!/usr/bin/env python3
"""Per-wave phase cycle sums of gemm_split_f16_kernel (experiments build, ISPK_SPLIT_ABLATE=7).  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP
R = 32768
for name, (N, K), tile in [("ffn1", (1536, 384), "442"), ("ffn2", (384, 1536), "342"), ("qkv", (512, 384), "442"), ("ffn1", (1536, 384), "441")]:
    a = runtime.split_f16(synth._normal(f"ab/{K}", (R, K)).cuda())
    w = runtime.split_f16(synth._normal(f"ab/{N}/{K}", (N, K), K ** -0.5).cuda())
    nwg, nwv = 4096, 8
    buf = torch.zeros((nwg, nwv, 8), dtype=torch.int64, device="cuda")
    os.environ["ISPK_SPLIT_TILE"] = tile
    split_out = name == "ffn1"
    for _ in range(3):
        runtime.gemm_split(a, w, flags=runtime.EP_GELU if split_out else 0, out_split=split_out)
    os.environ["ISPK_SPLIT_ABLATE"] = "7"
    os.environ["ISPK_SPLIT_STAMPS"] = hex(buf.data_ptr())
    runtime.gemm_split(a, w, flags=runtime.EP_GELU if split_out else 0, out_split=split_out)
    torch.cuda.synchronize()
    os.environ.pop("ISPK_SPLIT_ABLATE"); os.environ.pop("ISPK_SPLIT_STAMPS"); os.environ.pop("ISPK_SPLIT_TILE")
    b = buf.cpu().double()
    used = b[..., 5].sum(1) > 0
    b = b[used]
    names = ["wait+barrier", "reads ks0", "mfma ks0", "wait ks1", "mfma ks1", "main loop", "epilogue"]
    nk = (K + 31) // 32
    print(f"{name} tile {tile}: {int(used.sum())} workgroups, {nk} chunks; mean cycles per wave (per chunk):")
    for i, nm in enumerate(names):
        m = b[..., i].mean().item()
        print(f"   {nm:14s} {m:10.0f}" + (f"  ({m / nk:7.0f})" if i < 5 else ""))
    t0 = b[..., 7].amin(1)
    span = (b[..., 7].amin(1) - b[..., 7].min()).sort().values   # start times relative to the first workgroup
    print("   workgroup start offsets (cycles) quantiles:", [int(span[int(q * (len(span) - 1))]) for q in (0, .25, .5, .75, 1)])
