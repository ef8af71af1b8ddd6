#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_try
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run d python -c "import torch; x=torch.arange(1000, device='cuda'); print(x.sum().item())"
run e python -c "import torch; x=torch.randn(1000, device='cuda'); y=torch.arange(16000000, device='cuda'); print(y.sum().item())"
run f python -c "import torch; torch.cuda.set_device(0); x=torch.randn(1000, device='cuda'); print(x.sum().item())"
run g python -c "
import sys; sys.path.insert(0,'$REPO')
import torch
from view_neti_amd import synth, sd_config as sc
w = synth.clip_weights(sc.tiny().clip, device='cuda'); print(len(w))"
