#!/bin/bash
# erosion parity tests on the GPU + the scheduler's clock breakdown
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-erocheck}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "erosion or bench_step or smoke or proc_gen" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
bash tools/gpu_ero_clk.sh ${1:-erocheck}
