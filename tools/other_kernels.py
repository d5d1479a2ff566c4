#!/usr/bin/env python3
"""Per-step time of the kernels in a rocprofv3 trace of bench.py that are NOT libispk launches (ATen element-wise glue,
Tensile GEMMs, runtime copies / fills) -> profiles/other_kernels.json, which bench.py reports beside its HIP-event table
(those kernels carry no event label).  usage: other_kernels.py <trace dir> <step executions in the trace> [out.json]"""
import csv, glob, json, os, re, sys
d, steps = sys.argv[1], int(sys.argv[2])
out = sys.argv[3] if len(sys.argv) > 3 else "profiles/other_kernels.json"
f = max(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
ours = re.compile(r"_kernel\b|ffn2_bf16|ffn_bf16|gemm_|attn_|mas_kernel|layernorm")
other, mine = {}, 0.0
for r in csv.DictReader(open(f)):
    name, calls, total = r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])
    if ours.search(name) and "at::native" not in name and not name.startswith("Cijk_"):
        mine += total
    else:
        other[name[:90]] = {"calls_per_step": round(calls / steps, 2), "us_per_step": round(total / steps / 1e3, 2)}
res = {"source": "rocprofv3 --kernel-trace --stats of `bench.py --no-extras` (graph warm-up, warm-up and timed steps; "
                 "one-time weight staging is included in the average)",
       "steps_in_trace": steps, "libispk_us_per_step": round(mine / steps / 1e3, 1),
       "other_us_per_step": round(sum(v["us_per_step"] for v in other.values()), 2), "kernels": other}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
