#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-eronear}
mkdir -p $OUT
cd $ROOT
(
for cfg in "16384 1000000" "4096 100000" "1024 30000" "16384 100000"; do
  set -- $cfg
  echo "== $1 $2 baseline"; timeout 60 python tools/ero_sweep.py $1 $2 "0:1024" 2>&1 | tail -1
  echo "== $1 $2 near 512 slice 128"; TERRA_ERO_NEAR=512 timeout 60 python tools/ero_sweep.py $1 $2 "0:128" 2>&1 | tail -1
  echo "== $1 $2 near 25% slice 128"; TERRA_ERO_NEAR=-4 timeout 60 python tools/ero_sweep.py $1 $2 "0:128" 2>&1 | tail -1
done
) > $OUT/near_sweep2.txt 2>&1
cat $OUT/near_sweep2.txt | cut -c1-140
