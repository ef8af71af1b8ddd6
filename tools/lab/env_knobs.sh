#!/bin/bash
# Dev tool (GPU box): runtime environment knobs against the default, same autotune picks on every leg.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VNETI_AUTOTUNE_CACHE=/tmp/env_knobs_autotune.json
run() { python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])"; }
run > /dev/null
for i in 1 2; do
  echo "default $(run) | HIP_FORCE_DEV_KERNARG=1 $(HIP_FORCE_DEV_KERNARG=1 run) | HIP_FORCE_DEV_KERNARG=0 $(HIP_FORCE_DEV_KERNARG=0 run) | GPU_MAX_HW_QUEUES=2 $(GPU_MAX_HW_QUEUES=2 run) | DEBUG_HIP_GRAPH_... n/a"
done
env | grep -i "^HIP_\|^HSA_\|^GPU_\|^ROC" | head
