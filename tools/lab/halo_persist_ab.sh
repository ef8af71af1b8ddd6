#!/bin/bash
# Lab (round 6): the persistent halo kernel (tools/lab/build_halo_persist.sh) against the product tile in
# the train step WITH THE FORK REMOVED (VNETI_NO_OVERLAP=1), alternating processes on one box; the product step with the fork
# beside them.  First the lab kernel's results: the halo kernel tests through the lab library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
LAB=tools/lab/libvneti_halo_persist.so
VNETI_LIB_PATH=$LAB timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "halo" 2>&1 | tail -3
for i in 1 2 3; do
  a=$(VNETI_NO_OVERLAP=1 python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1)
  b=$(VNETI_NO_OVERLAP=1 VNETI_LIB_PATH=$LAB python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1)
  c=$(python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 20 2>/dev/null | tail -1)
  echo "product, no fork $a | persistent halo, no fork $b | product, fork $c"
done
