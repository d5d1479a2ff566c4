"""Prints VGPR / spill / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage), one line each."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
srcs = sys.argv[1:] or sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))
for src in srcs:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(HERE, src),
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur, vals = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            if cur:
                print(f"{src:16s} {cur[:70]:70s} {vals}")
            cur, vals = m.group(1), {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|SGPRs|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m:
            vals[m.group(1).split(" [")[0]] = int(m.group(2))
    if cur:
        print(f"{src:16s} {cur[:70]:70s} {vals}")
