#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-erosweep2}
mkdir -p $OUT
cd $ROOT
(
for near in 128 256 512 1024; do
  for sl in 64 128 256; do
    echo "== near $near slice $sl"; TERRA_ERO_NEAR=$near timeout 60 python tools/ero_sweep.py 4096 1000000 "2048:$sl" 2>&1 | head -1
  done
done
for w in 1024 1536 3072; do echo "== W $w near 512 slice 128"; TERRA_ERO_NEAR=512 timeout 60 python tools/ero_sweep.py 4096 1000000 "$w:128" 2>&1 | head -1; done
) > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt | cut -c1-150
