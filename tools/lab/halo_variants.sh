#!/bin/bash
# Dev tool (GPU box): tile 18 on the step's convolutions under several builds of the library (tools/lab/libvneti_*.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in prev v1 v2; do
    echo "== $v =="; VNETI_LIB_PATH=tools/lab/libvneti_$v.so python tools/conv_halo_ab.py 2>&1 | grep "^conv" | grep -v 2x32x48 | sed 's/h16.*//'
  done
  echo "== tree (v3) =="; python tools/conv_halo_ab.py 2>&1 | grep "^conv" | grep -v 2x32x48 | sed 's/h16.*//'
done
