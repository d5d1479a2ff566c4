#!/usr/bin/env python3
"""Tile sweep of gemm_split_f16_kernel at the aligner / adaptor shapes of the B = 64 forward (experiments build).  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP
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
    return sorted(ts)[len(ts) // 2]


shapes = {"key conv0": (6656, 768, 1920), "key conv1": (6656, 128, 768), "query conv0": (33020, 160, 400), "query conv1": (33020, 80, 800),
          "query conv2": (33020, 128, 80), "pred ffn1": (6400, 1024, 256), "pred ffn2": (6400, 256, 1024), "emb out": (6400, 384, 256),
          "pred qkv": (6400, 384, 256), "to_mel": (32768, 80, 384),
          "enc qkv": (6400, 512, 384), "enc out": (6400, 384, 384), "enc ffn1": (6400, 1536, 384), "enc ffn2": (6400, 384, 1536),
          "dec qkv": (32768, 512, 384), "dec out": (32768, 384, 384), "dec ffn1": (32768, 1536, 384), "dec ffn2": (32768, 384, 1536)}
if len(sys.argv) > 1:
    shapes = {k: v for k, v in shapes.items() if any(a in k for a in sys.argv[1:])}
for name, (M, N, K) in shapes.items():
    a = runtime.split_f16(synth._normal(f"sw/a/{K}", (M, K)).to(dev))
    w = runtime.split_f16(synth._normal(f"sw/w/{N}/{K}", (N, K), K ** -0.5).to(dev))
    row = []
    for tile in (None, 221, 241, 242, 321, 341, 342, 421, 441, 442):
        if tile is not None and (tile // 100) * 64 > ((N + 63) // 64) * 64 + 64:
            continue
        if tile is not None:
            os.environ["ISPK_SPLIT_TILE"] = str(tile)
        row.append((tile, time_it(lambda: runtime.gemm_split(a, w))))
        os.environ.pop("ISPK_SPLIT_TILE", None)
    best = min(row[1:], key=lambda tv: tv[1])
    print(f"{name:12s} {M}x{N}x{K}: auto {row[0][1]:6.1f} (tile {runtime.lib().ispk_gemm_split_f16_tile(M, N, K)})  best {best[0]} {best[1]:6.1f}   "
          + " ".join(f"{t}:{v:.0f}" for t, v in row[1:]), flush=True)
