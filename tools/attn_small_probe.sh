#!/bin/bash
# the step's latency-bound attention launches (low-resolution UNet levels, CLIP) — tools/attn_probe.py per shape
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "4 8 1024 1024 80" "4 8 1024 77 80" "4 8 256 256 160" "4 8 256 77 160" "4 8 64 64 160" "4 8 64 77 160" "64 12 77 77 64" "4 8 4096 77 40" "4 8 4096 4096 40"; do
  set -- $cfg
  B=$1 H=$2 N=$3 NK=$4 D=$5 python $REPO/tools/attn_probe.py 2>&1 | grep "^attn" | sed "s/^/B=$1 H=$2 /"
done
