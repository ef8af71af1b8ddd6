#!/bin/bash
# Dev tool (run on the GPU box through gpurun): the measurement set of one round.
#   tools/profile_round.sh r01c
# 1. bench.py default run (JSON line)                      -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command  -> gpurun_out/<tag>_stats/
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only, eager launches)
#                                                          -> gpurun_out/<tag>_pmc_{fetch,write}/
# 4. tools/pmc_summary.py folds 2+3 into small CSV/JSON files to be copied into profiles/.
set -u
TAG=${1:-r01x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json; echo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- \
  python $REPO/bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- \
    python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-roofline > /dev/null 2> $OUT/${TAG}_pmc_$c.err
done
python $REPO/tools/pmc_summary.py $TAG
