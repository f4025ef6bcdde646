#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-erodiag}
mkdir -p $OUT
cd $ROOT
for lm in 1 2 0; do
  echo "== TERRA_ERO_LEAD=$lm"
  TERRA_ERO_LEAD=$lm timeout 300 python tools/ero_sweep.py 16384 1000 "2048:1024" 2>&1 | tail -1
  TERRA_ERO_LEAD=$lm timeout 300 python tools/ero_sweep.py 4096 200000 "2048:1024" 2>&1 | tail -1
done > $OUT/lead_modes.txt 2>&1
cat $OUT/lead_modes.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "erosion or proc_gen or bench_step" > $OUT/pytest_ero.log 2>&1; tail -3 $OUT/pytest_ero.log
