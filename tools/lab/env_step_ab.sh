#!/bin/bash
# Lab: the train step under several ENVIRONMENT settings (one per arm, "K=V[,K=V]" or "-" for none), alternating processes on
# one box, all arms replaying the tile picks of ONE autotune pass.   env_step_ab.sh <rounds> <arm> [<arm> ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
export VNETI_AUTOTUNE_CACHE=/tmp/env_ab_picks.json
rm -f $VNETI_AUTOTUNE_CACHE
python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > /dev/null 2>&1
for i in $(seq $R); do
  line=""
  for arm in "$@"; do
    envs=""; [ "$arm" != "-" ] && envs=$(echo $arm | tr ',' ' ')
    v=$(env $envs python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1)
    line="$line | [$arm] $(echo $v | python -c 'import sys,json; print("%.2f" % json.loads(sys.stdin.read())["value"])' 2>/dev/null || echo FAIL)"
  done
  echo "$line"
done
