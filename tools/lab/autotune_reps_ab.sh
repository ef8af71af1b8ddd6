#!/bin/bash
# Dev tool (GPU box): does a steadier autotune (more cold timings per candidate) pick a faster schedule?
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])"; }
for i in 1 2 3; do
  echo "reps 9: $(VNETI_AUTOTUNE_REPS=9 run)  reps 25: $(VNETI_AUTOTUNE_REPS=25 run)  reps 5: $(VNETI_AUTOTUNE_REPS=5 run)"
done
