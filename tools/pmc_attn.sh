#!/bin/bash
# counters of the three attention kernels at the UNet's 64x64 self-attention shape (B=4, H=8, N=4096, D=40 by default)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/attn_probe.py
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $REPO/tools/attn_probe.py > $OUT/p$i.log 2>&1 || echo "set $i failed: $set"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("attn_fwd", "attn_dq", "attn_dkv"):
            if k in n and "reduce" not in n:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print("==", k)
    for c, v in sorted(acc[k].items()):
        print(f"  {c:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
find $OUT -name "*kernel_trace.csv" -delete
