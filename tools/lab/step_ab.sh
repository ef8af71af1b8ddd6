#!/bin/bash
# Dev tool (GPU box): the train step under tools/lab/libvneti_prev.so and under the in-tree library, alternating processes
# (each autotunes for itself), steps/s of `bench.py --no-roofline`.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  a=$(VNETI_LIB_PATH=tools/lab/libvneti_prev.so VNETI_AUTOTUNE_CANDS=${PREV_CANDS:-1,2,3,5,6,7,8,9,10,11,12,13,14,15,16,17,18} python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1)
  b=$(python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1)
  echo "prev $a | new $b"
done
