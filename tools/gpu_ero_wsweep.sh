#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-erow}
mkdir -p $OUT
cd $ROOT
(
for n in 16384 8192; do
for w in 2048 4096 8192 16384; do
  echo "== $n 1000000 W $w"; timeout 120 python tools/ero_sweep.py $n 1000000 "$w:128" 2>&1 | head -1
done
done
for w in 1024 2048 4096; do echo "== 16384 1000 W $w"; timeout 120 python tools/ero_sweep.py 16384 1000 "$w:128" 2>&1 | head -1; done
for w in 2048 4096 8192 16384; do echo "== 16384 100000 W $w"; timeout 120 python tools/ero_sweep.py 16384 100000 "$w:128" 2>&1 | head -1; done
) > $OUT/wsweep.txt 2>&1
cat $OUT/wsweep.txt | cut -c1-140
