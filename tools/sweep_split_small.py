#!/usr/bin/env python3
"""Tile sweep of gemm_split_f16_kernel at the text encoder's 6,400 rows (experiments build, ISPK_SPLIT_TILE).  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP
R = int(os.environ.get("R", 6400))
dev = "cuda"


def time_it(fn, rounds=5, inner=20):
    for _ in range(3):
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
    return sorted(ts)[len(ts) // 2]


for name, (N, K) in {"qkv": (512, 384), "out": (384, 384), "ffn1": (1536, 384), "ffn2": (384, 1536)}.items():
    a = runtime.split_f16(synth._normal(f"ab/{K}", (R, K)).to(dev))
    w = runtime.split_f16(synth._normal(f"ab/{N}/{K}", (N, K), K ** -0.5).to(dev))
    row = []
    for tile in (None, 221, 241, 321, 341, 421, 441):
        if tile is not None and N % (64 * (tile // 100)) != 0:
            row.append((tile, None))
            continue
        if tile is not None:
            os.environ["ISPK_SPLIT_TILE"] = str(tile)
        row.append((tile, time_it(lambda: runtime.gemm_split(a, w, flags=runtime.EP_GELU if name == "ffn1" else 0, out_split=name == "ffn1"))))
        os.environ.pop("ISPK_SPLIT_TILE", None)
    print(f"{name:5s} rows {R}: " + "  ".join(f"{t or 'auto'}: {'-' if v is None else f'{v:6.1f}'}" for t, v in row), flush=True)
