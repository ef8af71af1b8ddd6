#!/bin/bash
# SQ / LDS / cache counters of the generic 256x256 tile (5) and the 8-phase tile (16) on one GEMM; separate --pmc passes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_gemm8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $REPO/tools/pmc_gemm8.py > $OUT/p$i.log 2>&1 || echo "set $i failed: $set"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gemm8" if "gemm8_kernel" in n else ("tile5" if "gemm_kernel" in n else None)
        if k:
            acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gemm8" if "gemm8_kernel" in n else ("tile5" if "gemm_kernel" in n else None)
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("durations under the profiler (us):", {k: round(sorted(v)[len(v) // 2], 1) for k, v in dur.items()})
print(f"{'counter':34s} {'tile5':>16s} {'gemm8':>16s}")
for c, d in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in d.items()}
    print(f"{c:34s} {m.get('tile5', 0):16.0f} {m.get('gemm8', 0):16.0f}")
PY
find $OUT -name "*kernel_trace.csv" -delete
