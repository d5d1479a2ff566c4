#!/usr/bin/env python3
"""Compressed view of a kernel's instruction stream (loads / LDS / MFMA / waits / barriers / branches in order).
Usage: isa_summary.py file.s kernel_name_substring"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
keys = ("s_waitcnt", "s_barrier", "global_load", "v_mfma", "ds_write", "ds_read", "s_cbranch", "global_store", "scratch",
        "buffer_", "v_exp", "v_rcp")
out = []
for l in src[start:end]:
    t = l.strip()
    if t.startswith(".LBB") and t.endswith(":"):
        out.append([t, 1])
        continue
    for k in keys:
        if t.startswith(k):
            key = t.split()[0]
            if k in ("s_waitcnt", "s_cbranch"):
                key = t.split(";")[0].strip()
            if out and out[-1][0] == key:
                out[-1][1] += 1
            else:
                out.append([key, 1])
            break
print(f"{end - start} lines")
for k, c in out:
    print(f"{c:4d} x {k}")
