#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02e}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile or voxel or smoke" > $OUT/pytest_tiles.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_tiles.log
tail -3 $OUT/pytest_tiles.log
timeout 300 python tools/prof_tiles.py > $OUT/prof_tiles.txt 2>&1; tail -5 $OUT/prof_tiles.txt
timeout 600 python tools/ero_sweep.py 4096 1000000 "2048:1024,8192:64,16384:64,16384:32,16384:128,16384:256" > $OUT/ero_sweep.txt 2>&1; cat $OUT/ero_sweep.txt
