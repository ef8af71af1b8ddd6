#!/bin/bash
# Dev tool (GPU box): SQ counters of the halo tile (18) on the VAE's 512^2 128->128 conv beside the 256x256 8-phase tile
# (16) on the 256^2 256->256 conv (same FLOPs): what the waves of each spend their cycles on.  Separate --pmc passes.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_halo
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # tag env...
  tag=$1; shift
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    env "$@" REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${tag}_p$i -- python $REPO/tools/pmc_conv.py > $OUT/${tag}_p$i.log 2>&1 || echo "$tag set $i failed: $set"
  done
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("$OUT/${tag}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm8_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/${tag}_p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm8_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== $tag: kernel us (under counters)", [round(d, 1) for d in dur])
for k, v in sorted(acc.items()):
    print(f"{k:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
}
run halo18 HINT=18 KORDER=1 HW=512 CI=128 CO=128
run tile16 HINT=16 KORDER=1 HW=256 CI=256 CO=256
find $OUT -name "*kernel_trace.csv" -delete
