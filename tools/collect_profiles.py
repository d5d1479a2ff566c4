#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left under gpurun_out/refresh/ into profiles/ (tracked): kernel-stats CSVs, the
timeline of one step, the default bench line, the one-rank distributed rehearsal, traffic.json / other_kernels.json and the
feed-forward kernels' SQ counter summary.  usage: collect_profiles.py <tag>   e.g. r02"""
import glob, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst, tag = os.path.join(root, "gpurun_out", "refresh"), os.path.join(root, "profiles"), sys.argv[1]
for dt in ("bf16", "f32", "split", "train"):
    f = max(glob.glob(f"{src}/trace_{dt}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)   # newest run
    shutil.copy(f, f"{dst}/{tag}_{dt}_kernel_stats.csv")
for name, to in (("bench.json", f"{tag}_bench.json"), ("bench_force_dist.json", f"{tag}_bench_force_dist.json"),
                 ("timeline_bf16.txt", f"{tag}_bf16_timeline.txt"), ("pmc_ffn_summary.txt", f"{tag}_ffn_sq_counters.txt"),
                 ("traffic.json", "traffic.json"), ("other_kernels.json", "other_kernels.json")):
    if os.path.exists(f"{src}/{name}"):
        shutil.copy(f"{src}/{name}", f"{dst}/{to}")
        print("copied", to)
