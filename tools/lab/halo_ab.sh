#!/bin/bash
# Dev tool (GPU box): the halo tile (18) of the in-tree library against tools/lab/libvneti_prev.so (the build before a
# kernel change), alternating processes; then the bit-identity / parity tests of the tile.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  echo "== prev =="; VNETI_LIB_PATH=tools/lab/libvneti_prev.so python tools/conv_halo_ab.py 2>&1 | grep "^conv"
  echo "== new ==";  python tools/conv_halo_ab.py 2>&1 | grep "^conv"
done
echo "== dgrad (MODE=2) new =="; MODE=2 python tools/conv_halo_ab.py 2>&1 | grep "^conv"
python -m pytest tests/test_kernels_gpu.py -q -k "halo or conv" 2>&1 | tail -3
