#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): default bench line, rocprofv3 kernel traces (bf16 / f32) and the two PMC passes.
# Everything lands under gpurun_out/refresh/; tools/collect_profiles.py then copies the summaries into profiles/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf "$O" && mkdir -p "$O"
cd "$R"
cd /tmp && export TMPDIR=/tmp
# per-kernel evidence is taken with ONE batch in flight (two overlapping graphs stretch each other's kernels in a trace)
B="$R/bench.py --no-kernel-events --no-cpu-baseline --no-f32-line --in-flight 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_bf16" -- python3 $B --steps 10 --warmup 2 > "$O/trace_bf16.log" 2>&1
echo "trace bf16 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_f32" -- python3 $B --steps 10 --warmup 2 --dtype f32 > "$O/trace_f32.log" 2>&1
echo "trace f32 done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 $B --steps 2 --warmup 1 --no-graph > "$O/pmc_fetch.log" 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 $B --steps 2 --warmup 1 --no-graph > "$O/pmc_write.log" 2>&1
echo "pmc write done"
# the default bench line LAST, with the traffic figures of THIS build (bench.py reads profiles/traffic.json)
cd "$R"
python3 tools/pmc_traffic.py "$O/pmc_fetch" "$O/pmc_write" profiles/traffic.json > "$O/traffic.log"
timeout -k 10 400 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
echo "bench done"
# keep only the small summaries (the merged-back directory is capped at 64 MiB)
find "$O" -name "*kernel_trace.csv" -delete
ls -la "$O" "$O"/*/* | head -40
