#!/bin/bash
# round 6: instruction-level evidence for the d = 40 attention kernels at N = 4096.
#  1. try a rocprofv3 thread trace (ATT / SQTT) of one launch per kernel — needs the trace-decoder library, which this
#     image may not ship (the attempt's outcome is recorded either way);
#  2. the SQ stall counters of the same launches in separate --pmc passes (tools/pmc_attn.sh): WAVE_CYCLES split into
#     WAIT_ANY (s_waitcnt / barrier), WAIT_INST_ANY (issue stalls), ACTIVE_INST_* per pipe.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/att_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== decoder library search" > $OUT/att_attempt.txt
find / -xdev \( -name "librocprof-trace-decoder*" -o -name "libatt_decoder*" -o -name "*trace_decoder*.so*" \) 2>/dev/null >> $OUT/att_attempt.txt
for k in attn_fwd attn_dq attn_dkv; do
  echo "== rocprofv3 --att $k" >> $OUT/att_attempt.txt
  timeout 240 rocprofv3 --att --att-target-cu 1 --att-activity 8 --kernel-include-regex "$k" --kernel-iteration-range "[4]" \
      -d $OUT/att_$k -- python $REPO/tools/attn_probe.py >> $OUT/att_attempt.txt 2>&1
  echo "rc=$?" >> $OUT/att_attempt.txt
  find $OUT/att_$k -type f | head -30 >> $OUT/att_attempt.txt
done
du -sh $OUT/att_* >> $OUT/att_attempt.txt 2>&1
# keep the merged-back payload small: raw .att / .out blobs stay on the box unless they are tiny
find $OUT -type f -size +4M -delete
bash $REPO/tools/pmc_attn.sh > $OUT/pmc_attn.txt 2>&1
tail -60 $OUT/pmc_attn.txt
