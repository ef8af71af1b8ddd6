#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_conv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for h in 6 5 1; do HINT=$h python $REPO/tools/pmc_conv.py; done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $REPO/tools/pmc_conv.py > $OUT/p$i.log 2>&1 || echo "set $i failed: $set"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
find $OUT -name "*kernel_trace.csv" -delete
