#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-erow2}
mkdir -p $OUT
cd $ROOT
(
for w in 24576 32768 65536; do echo "== 16384 1000000 W $w"; timeout 120 python tools/ero_sweep.py 16384 1000000 "$w:128" 2>&1 | head -1; done
for w in 32768; do echo "== 16384 100000 W $w"; timeout 120 python tools/ero_sweep.py 16384 100000 "$w:128" 2>&1 | head -1; done
for w in 3072 4096; do echo "== 4096 1000000 W $w"; timeout 120 python tools/ero_sweep.py 4096 1000000 "$w:128" 2>&1 | head -1; done
for w in 2048 4096; do echo "== 4096 100000 W $w"; timeout 120 python tools/ero_sweep.py 4096 100000 "$w:128" 2>&1 | head -1; done
for w in 2048 4096; do echo "== 1024 30000 W $w"; timeout 120 python tools/ero_sweep.py 1024 30000 "$w:128" 2>&1 | head -1; done
) > $OUT/wsweep.txt 2>&1
cat $OUT/wsweep.txt | cut -c1-140
