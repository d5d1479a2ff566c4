#!/usr/bin/env python3
"""Where does gemm_split_f16_kernel spend its time?  Experiments build (libispk_exp.so): ISPK_SPLIT_ABLATE = 1 no MFMAs,
2 no operand DMA, 3 DMA + barriers only; ISPK_SPLIT_TILE = TN*10 + WM forces a tile.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP
R = int(os.environ.get("R", 32768))
dev = "cuda"


def time_it(fn, rounds=5, inner=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


shapes = {"qkv": (512, 384), "out": (384, 384), "ffn1": (1536, 384), "ffn2": (384, 1536)}
for name, (N, K) in shapes.items():
    a = runtime.split_f16(synth._normal(f"ab/{K}", (R, K)).to(dev))
    w = runtime.split_f16(synth._normal(f"ab/{N}/{K}", (N, K), K ** -0.5).to(dev))
    split_out = name == "ffn1"
    tiles = [342] if N % 384 == 0 else [442]
    for tile in tiles:
        row = []
        for ab in (None, "3", "9", "8"):
            if tile is not None:
                os.environ["ISPK_SPLIT_TILE"] = str(tile)
            if ab:
                os.environ["ISPK_SPLIT_ABLATE"] = ab
            t = time_it(lambda: runtime.gemm_split(a, w, flags=runtime.EP_GELU if split_out else 0, out_split=split_out))
            os.environ.pop("ISPK_SPLIT_ABLATE", None)
            os.environ.pop("ISPK_SPLIT_TILE", None)
            row.append(t)
        print(f"{name:5s} tile {tile or 'auto':>4}: full {row[0]:7.1f} us | DMA + barriers only {row[1]:7.1f} | the same as whole 128-B lines {row[2]:7.1f} | epilogue only {row[3]:7.1f}",
              flush=True)
