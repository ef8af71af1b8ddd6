#!/bin/bash
# Lab: tile 13 (= tile 7 as a persistent kernel, tools/lab/gemm_persist.patch on csrc/gemm_conv.hip) against tile 7 on the VAE's
# 512x512 N=128 convs.  Not in the product library: the restructured kernel body cost the ordinary tiles 1.7 % of the step, and
# tile 13 itself is 14-17 % slower than tile 7 on the launches it was meant for.
# Builds the patched GEMM into tools/lab/libvneti_persist.so (needs view_neti_amd/csrc/build/*.o: run csrc/build.py first),
# checks tile 13 bit-for-bit against tile 8 (plain / conv / row-add+residual+act / GroupNorm sums), then sweeps K on one box.
# r02 (same box): Ci=64/128/256: tile 7 349.7 / 530.0 / 925.5 us, tile 13 420.8 / 615.2 / 1085.7 us
#   => per tile 9.5 us + 1.37 us per k-step (7) against 13-15 us + 1.4-1.5 us per k-step (13).
# NOTE: the patch was cut against gemm_conv.hip as of commit 7fdf30c (before the epilogue levels / CONV flag / 4-stage
# rings); it needs a re-base onto the current kernel before this script runs again.
set -e
cd "$(dirname "$0")/../.."
cp view_neti_amd/csrc/common.h /tmp/common.h
sed 's#"../../include/vneti.h"#"'$PWD'/include/vneti.h"#' view_neti_amd/csrc/gemm_conv.hip > /tmp/gemm_persist_lab.hip
sed 's#^+++ .*#+++ /tmp/gemm_persist_lab.hip#; s#^--- .*#--- /tmp/gemm_persist_lab.hip#' tools/lab/gemm_persist.patch | patch -p0 /tmp/gemm_persist_lab.hip
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-fast-math -c /tmp/gemm_persist_lab.hip -o /tmp/gemm_persist_lab.o
hipcc -shared -fPIC --offload-arch=gfx950 -o tools/lab/libvneti_persist.so /tmp/gemm_persist_lab.o $(ls view_neti_amd/csrc/build/*.o | grep -v gemm_conv.o)
export VNETI_LIB_PATH=tools/lab/libvneti_persist.so
python tools/lab/persist_check.py
for ci in 64 128 256; do
  for h in 7 13; do
    HW=512 CI=$ci CO=128 HINT=$h python tools/pmc_conv.py 2>/dev/null | tail -1
  done
done
