#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left under gpurun_out/refresh/ into profiles/ (tracked): kernel-stats CSVs, the
default bench line, and traffic.json (via tools/pmc_traffic.py).  usage: collect_profiles.py <tag>   e.g. r01_final"""
import glob, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst, tag = os.path.join(root, "gpurun_out", "refresh"), os.path.join(root, "profiles"), sys.argv[1]
for dt in ("bf16", "f32"):
    f = max(glob.glob(f"{src}/trace_{dt}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)   # newest run
    shutil.copy(f, f"{dst}/{tag.replace('_final', '')}_{dt}_final_kernel_stats.csv" if tag.endswith("_final") else f"{dst}/{tag}_{dt}_kernel_stats.csv")
shutil.copy(f"{src}/bench.json", f"{dst}/{tag}_bench.json")
subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), f"{src}/pmc_fetch", f"{src}/pmc_write",
                       f"{dst}/traffic.json"])
