#!/bin/bash
# quick GPU check of the fBm kernels: the grid / tile parity cases + timings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-quick}
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or grid or noise or tiles or random or minmax or strips or generator" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 300 python tools/prof_noise.py 4096 5 1,2,4 > $OUT/noise_4096.txt 2>&1
timeout 300 python tools/prof_noise.py 16384 2 1,2,4 > $OUT/noise_16384.txt 2>&1
cat $OUT/noise_4096.txt $OUT/noise_16384.txt
