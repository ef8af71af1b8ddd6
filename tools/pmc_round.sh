#!/bin/bash
# counter passes of a round (GEMM replay, see tools/pmc_gemm.py):  tools/pmc_round.sh r01c
TAG=${1:-r01x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/pmc_gemm.py --dump $OUT/gemm_specs.json 2>&1 | tail -1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- \
    python $REPO/tools/pmc_gemm.py --replay $OUT/gemm_specs.json > $OUT/${TAG}_pmc_$c.log 2>&1
  echo "$c rc=$?"; tail -1 $OUT/${TAG}_pmc_$c.log
done
python $REPO/tools/pmc_summary.py $TAG
find $OUT -name "*kernel_trace.csv" -path "*_pmc_*" -delete
