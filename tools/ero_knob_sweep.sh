#!/bin/bash
# dense erosion knob sweep on the current code (4096^2, 10^6 droplets)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
run() { echo "== $*"; env "$@" timeout 120 python tools/ero_sweep.py 4096 1000000 ${COMBO:-0:0} 2>&1 | grep "^W"; }
(
run A=0
run TERRA_ERO_CK=64:16
run TERRA_ERO_CK=16:16
run TERRA_ERO_CK=32:8
run TERRA_ERO_CK=48:16
run TERRA_ERO_NEAR=256
run TERRA_ERO_NEAR=1024
run TERRA_ERO_NEAR=2048
run TERRA_ERO_LEAD=1
run TERRA_ERO_LIVE=0
COMBO="2048:0,3072:0,6144:0,8192:0,0:64,0:256,0:512" run A=0
) 2>&1 | tee gpurun_out/sweep/sweep4096.txt
