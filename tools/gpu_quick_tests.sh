#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-quick2}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "${2:-dense or strips or ao or generator or erosion}" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python tools/prof_tiles.py 0 3 > $OUT/prof_tiles.txt 2>&1; tail -3 $OUT/prof_tiles.txt
timeout 300 python tools/ero_sweep.py 16384 1000 "2048:1024" 2>&1 | tail -1
timeout 300 python tools/ero_sweep.py 4096 200000 "2048:1024" 2>&1 | tail -1
timeout 300 python tools/prof_ao.py 2>&1 | tail -2
