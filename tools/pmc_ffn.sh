#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes over the two fused feed-forward kernels (tools/run_ffn_once.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_ffn
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/counters.txt" 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/pass$i" -- python3 $R/tools/run_ffn_once.py > "$O/pass$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_ffn"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = ("ffn2" + n[n.index("ffn2_bf16_kernel") + 16:].split(">")[0] + ">") if "ffn2_bf16_kernel" in n else ("ffn1" if "ffn_bf16_kernel" in n else None)
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O + "/summary.txt", "w") as out:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            v = sorted(v)
            line = f"{k} {c:32s} median {v[len(v)//2]:.4g}  (n={len(v)})"
            print(line); out.write(line + "\n")
PY
find "$O" -name "*.csv" -size +2M -delete
