#!/bin/bash
# Lab: end-to-end Coach.train rate (tools/bench_coach.py, device and host input pipelines) with the pinned upload stager and with
# blocking copies (VNETI_NO_STAGER=1), alternating, same autotune picks.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VNETI_AUTOTUNE_CACHE=/tmp/coach_ab_picks.json VNETI_ALLOW_SYNTHETIC_WEIGHTS=1
rm -f $VNETI_AUTOTUNE_CACHE
python tools/bench_coach.py --steps 20 --variants device > /dev/null 2>&1
for i in 1 2; do
  for arm in 0 1; do
    VNETI_NO_STAGER=$arm python tools/bench_coach.py --steps 100 2>/dev/null | grep steps_per_s | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('blocking' if $arm else 'stager  ', d['variant'], '%.2f steps/s' % d['steps_per_s'])"
  done
done
python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1
