#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; they do not fit one pass on
gfx950) into HBM bytes per launch per kernel and writes profiles/traffic.json (read by bench.py's roofline object).

gfx950 corrections (MI355X_MICROARCH.md, "HBM"): both counters are in KiB; FETCH_SIZE reports exactly half of the bytes of
wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

usage: pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [out.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def label_of(kernel_name: str):
    """rocprofv3 kernel name -> the label bench.py / runtime.py use: the kernel's name plus its leading template arguments
    (two for the tile / row-block GEMMs, one for the rest)."""
    m = re.search(r"(\w+_kernel)(?:<([^>]*)>)?", kernel_name)
    if not m:
        return None
    name, targs = m.group(1), [a.strip() for a in (m.group(2) or "").split(",") if a.strip()]
    if not targs:
        return name
    if name == "gemm_bf16_panel_kernel" and len(targs) >= 4 and targs[3] not in ("0", "false"):
        return f"{name}<{targs[0]},lnin>"          # the LayerNorm-prologue instances (ispk_gemm_bf16_lnin)
    n = 2 if name in ("gemm_bf16_wide_kernel", "gemm_bf16_kernel", "gemm_f32_kernel", "gemm_bf16_panel_kernel") else 1
    return f"{name}<{','.join(targs[:n])}>"


def per_kernel(d, counter):
    f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)   # newest run
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        key = label_of(r["Kernel_Name"])
        if key is None:
            continue
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) & set(write)):
    out[k] = {"launches": fetch[k][0], "fetch_KiB_raw": round(fetch[k][1], 1), "write_KiB": round(write[k][1], 1),
              "hbm_bytes_per_launch": round((2.0 * fetch[k][1] + write[k][1]) * 1024.0),
              "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE counts wide streaming reads at 1/2 on gfx950"}
path = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
json.dump(out, open(path, "w"), indent=1)
for k, v in out.items():
    print(f"{k:34s} n={v['launches']:4d} fetch(raw) {v['fetch_KiB_raw'] / 1024:8.2f} MiB  write {v['write_KiB'] / 1024:8.2f} MiB "
          f"-> {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch")
