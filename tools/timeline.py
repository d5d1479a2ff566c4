#!/usr/bin/env python3
"""Timeline of ONE step from a rocprofv3 kernel trace (run on the GPU box after
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-extras --no-kernel-events --steps S --warmup W`):
kernels of the last graph replay in start order with their start offset, duration and the idle gap in front of them; then
totals per kernel.  usage: timeline.py DIR [kernels per step]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
f = max(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
        for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]
# a step = the kernels between two mas_kernel launches (one per forward)
mas = [i for i, r in enumerate(rows) if "mas_kernel" in r[2]]
if len(mas) < 3:
    sys.exit("need at least 3 steps in the trace")
# a replay from the MIDDLE of the timed region: the last one runs after the closing synchronisation, with the host no
# longer ahead of the device (its launch latency shows up as idle time that steady-state replays do not have)
k = -5 if len(mas) >= 8 else -2
a, b = mas[k - 1], mas[k]
step = rows[a:b]
t0 = step[0][0]
print(f"# one step: {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us from the MAS kernel of one replay to the next")
end_prev = t0
busy = 0
QUEUES = os.environ.get("TIMELINE_QUEUES")      # also print the HSA queue / stream ids of every kernel
for s, e, n, qid, sid in step:
    gap = (s - end_prev) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {short(n)}" + (f"   [queue {qid} stream {sid}]" if QUEUES else ""))
    end_prev = max(end_prev, e)
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, _q, _s in step:
    tot[short(n)][0] += 1
    tot[short(n)][1] += (e - s) / 1e3
print("# totals per kernel")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  x{c:3d}  {n}")
