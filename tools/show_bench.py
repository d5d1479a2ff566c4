#!/usr/bin/env python3
"""Prints the headline and the per-kernel table of a bench.py JSON line.  usage: show_bench.py file.json [n_rows]"""
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(d["ms_per_step"], "ms/step", d["value"], d["unit"])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for k, v in list(d["roofline"]["kernels"].items())[:rows]:
    print(f"{k:34s} x{v['launches_per_step']:5.1f}  {v['avg_us']:7.2f} us  {v['ms_per_step']:.4f} ms  {v['bound']} {v['frac']}")
