#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel traces (bf16 / f32), the two PMC traffic passes, the SQ counter passes of
# the feed-forward kernels, a one-rank rehearsal of the distributed path, and LAST the default bench line.
# Everything lands under gpurun_out/refresh/; tools/collect_profiles.py then copies the summaries into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# per-kernel evidence: the headline configuration (ONE batch in flight), no extras
B="$R/bench.py --no-kernel-events --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_bf16" -- python3 $B --steps 10 --warmup 2 > "$O/trace_bf16.log" 2>&1
echo "trace bf16 rc=$?"
python3 $R/tools/timeline.py "$O/trace_bf16" > "$O/timeline_bf16.txt" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_f32" -- python3 $B --steps 10 --warmup 2 --dtype f32 > "$O/trace_f32.log" 2>&1
echo "trace f32 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_split" -- python3 $B --steps 10 --warmup 2 --dtype split > "$O/trace_split.log" 2>&1
echo "trace split rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train" -- python3 $R/bench.py --no-kernel-events --no-cpu-baseline --extras train_step --steps 8 --warmup 2 > "$O/trace_train.log" 2>&1
echo "trace train rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 $B --steps 2 --warmup 1 --no-graph > "$O/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 $B --steps 2 --warmup 1 --no-graph > "$O/pmc_write.log" 2>&1
echo "pmc write rc=$?"
bash $R/tools/pmc_ffn.sh > "$O/pmc_ffn.log" 2>&1
cp $R/gpurun_out/pmc_ffn/summary.txt "$O/pmc_ffn_summary.txt" 2>/dev/null
cd "$R"
python3 tools/pmc_traffic.py "$O/pmc_fetch" "$O/pmc_write" profiles/traffic.json > "$O/traffic.log" 2>&1
python3 tools/other_kernels.py "$O/trace_bf16" 14 profiles/other_kernels.json > "$O/other.log" 2>&1
# one-rank rehearsal of the distributed path (RCCL init, gather pipeline, barriers, strong / config-4 code with world = 1)
ISPK_BENCH_FORCE_DIST=1 timeout -k 10 400 python3 bench.py --no-kernel-events --no-cpu-baseline > "$O/bench_force_dist.json" 2> "$O/bench_force_dist.err"
echo "force-dist bench rc=$?"
# the default bench line LAST, with the traffic figures of THIS build (bench.py reads profiles/*.json)
timeout -k 10 500 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
echo "bench rc=$?"
cp profiles/traffic.json profiles/other_kernels.json "$O/" 2>/dev/null
# keep only the small summaries (the merged-back directory is capped at 64 MiB)
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*counter_collection.csv" -delete
ls -la "$O" | head -40
