#!/bin/bash
# Lab (round 6): the library with write-through (sc1) 16-byte output stores, -DVN_WT_STORES=1 -> tools/lab/libvneti_wt.so
set -e
cd "$(dirname "$0")/../.."
CS=$PWD/view_neti_amd/csrc
T=/tmp/wt_lab; mkdir -p $T
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -DVN_WT_STORES=${WT:-1}"
for f in $CS/*.hip; do hipcc $FL -c $f -o $T/$(basename $f .hip).o & done; wait
hipcc -shared -fPIC --offload-arch=gfx950 -o tools/lab/libvneti_wt.so $T/*.o
ls -la tools/lab/libvneti_wt.so
