#!/bin/bash
# Dev tool (run on the GPU box through gpurun): the measurement set of one round.
#   tools/profile_round.sh r01c
# 1. bench.py default run (JSON line)                      -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command  -> gpurun_out/<tag>_stats/
# 3. tools/pmc_summary.py folds 2 (and the counter passes of tools/pmc_round.sh, if present) into small
#    CSV/JSON files to be copied into profiles/.
set -u
TAG=${1:-r01x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the first run autotunes and saves its picks; the profiled run reuses them, so its kernel averages are the step's own
export VNETI_AUTOTUNE_CACHE=$OUT/${TAG}_autotune.json
rm -f $VNETI_AUTOTUNE_CACHE
python $REPO/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json; echo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- \
  python $REPO/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
# (the counter passes live in tools/pmc_round.sh: rocprofv3 + TCC counters segfaults around the whole bench process)
python $REPO/tools/pmc_summary.py $TAG
# the raw per-dispatch trace is tens of MB: keep only the summaries (gpurun copies back <= 64 MiB)
find $OUT/${TAG}_stats -name "*kernel_trace.csv" -delete
