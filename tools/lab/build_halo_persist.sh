#!/bin/bash
# Lab (round 6): the persistent halo kernel of round 4 (tools/lab/attic/gemm_halo_persistent.hip) as the
# step's tile 18, in a lab library — tools/lab/libvneti_halo_persist.so — for a same-box A/B with the fork removed
# (VNETI_NO_OVERLAP=1: its only recorded fault was starving the side stream).  Needs view_neti_amd/csrc/build/*.o.
set -e
cd "$(dirname "$0")/../.."
CS=$PWD/view_neti_amd/csrc
T=/tmp/halo_persist_lab; mkdir -p $T
cp $CS/common.h $CS/gemm_args.h $T/
sed 's#"../../include/vneti.h"#"'$PWD'/include/vneti.h"#' tools/lab/attic/gemm_halo_persistent.hip > $T/gemm_halo_persistent.hip
# tile 18 -> the persistent kernel (declared beside vneti_launch_gemm8)
python - "$CS/gemm_conv.hip" "$T/gemm_conv.hip" "$PWD" <<'PY'
import sys
src = open(sys.argv[1]).read().replace('"../../include/vneti.h"', '"%s/include/vneti.h"' % sys.argv[3])
old = "      const int rc = vneti_launch_gemm8(&g, cfg == 16 ? 256 : 128, cfg == 18 ? 1 : 0, st);"
new = "      const int rc = cfg == 18 ? vneti_launch_gemm_halo(&g, st) : vneti_launch_gemm8(&g, cfg == 16 ? 256 : 128, 0, st);"
assert old in src
src = src.replace(old, new)
i = src.index("namespace {")
src = src[:i] + "int vneti_launch_gemm_halo(void* gemm_args, hipStream_t st);\n" + src[i:]
open(sys.argv[2], "w").write(src)
PY
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16"
hipcc $FL -c $T/gemm_conv.hip -o $T/gemm_conv.o &
hipcc $FL -c $T/gemm_halo_persistent.hip -o $T/gemm_halo_persistent.o &
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o tools/lab/libvneti_halo_persist.so $T/gemm_conv.o $T/gemm_halo_persistent.o \
  $(ls $CS/build/*.o | grep -v "/gemm_conv.o")
ls -la tools/lab/libvneti_halo_persist.so
