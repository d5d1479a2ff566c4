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
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(gemm_\w+<[^>]*>|attn_\w+_kernel|layernorm_kernel<\d+|mas_kernel<\d>|aligner_scores_kernel"
                      r"|masked_instnorm_kernel|soft_average_kernel|pad_rows_kernel|linear_small_kernel)", r["Kernel_Name"])
        if not m:
            continue
        key = m.group(1).replace(" ", "")
        if key.startswith("layernorm_kernel<"):
            key += ">"
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
