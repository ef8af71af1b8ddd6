#!/bin/bash
# Lab: the train step under several builds of the library (VNETI_LIB_PATH), alternating processes on one box, ALL replaying the
# tile picks of one autotune pass (VNETI_AUTOTUNE_CACHE) so that only the kernels differ.   lib_ab.sh <rounds> <lib> [<lib> ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
export VNETI_AUTOTUNE_CACHE=/tmp/lib_ab_picks.json
rm -f $VNETI_AUTOTUNE_CACHE
python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > /dev/null 2>&1   # writes the picks (product library)
for i in $(seq $R); do
  line=""
  for lib in "$@"; do
    if [ "$lib" = product ]; then v=$(python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1)
    else v=$(VNETI_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1); fi
    line="$line | $(basename $lib .so | sed s/libvneti_//) $(echo $v | python -c 'import sys,json; print("%.2f" % json.loads(sys.stdin.read())["value"])')"
  done
  echo "$line"
done
