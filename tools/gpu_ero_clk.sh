#!/bin/bash
# where the time of a dense-erosion trace goes (device clocks in the scheduler report)
set -u
export TERRA_ERO_DIAG=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-eroclk}
mkdir -p $OUT
cd $ROOT
(
echo "== 4096 100000"; timeout 120 python tools/ero_sweep.py 4096 100000 "2048:128" 2>&1 | tail -4
echo "== 4096 1000000"; timeout 120 python tools/ero_sweep.py 4096 1000000 "2048:128" 2>&1 | tail -4
echo "== 16384 1000000"; timeout 120 python tools/ero_sweep.py 16384 1000000 "0:128" 2>&1 | tail -4
echo "== 1024 30000"; timeout 120 python tools/ero_sweep.py 1024 30000 "0:128" 2>&1 | tail -4
echo "== 16384 1000 (headline)"; timeout 120 python tools/ero_sweep.py 16384 1000 "0:128" 2>&1 | tail -4
) > $OUT/clk.txt 2>&1
cat $OUT/clk.txt
