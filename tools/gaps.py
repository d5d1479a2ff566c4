#!/usr/bin/env python3
"""Idle gaps (no kernel running on the device) longer than a threshold in a rocprofv3 kernel trace, with the kernels on
either side.  usage: gaps.py DIR [min_us=20] [last_n_kernels=1500]"""
import csv, glob, os, sys
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
last = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
f = max(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))[-last:]
end = rows[0][1]
prev = rows[0][2]
t0 = rows[0][0]
for s, e, n in rows[1:]:
    if (s - end) / 1e3 > thr:
        print(f"{(end - t0) / 1e3:10.1f} us  idle {(s - end) / 1e3:7.1f} us   after {prev[:60]}  |  before {n[:60]}")
    if e > end:
        end, prev = e, n
