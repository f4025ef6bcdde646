#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-erosweep3}
mkdir -p $OUT
cd $ROOT
(
for w in 3072 4096 6144; do
for near in 256 512 1024; do
  for sl in 64 128 256; do
    echo "== W $w near $near slice $sl"; TERRA_ERO_NEAR=$near timeout 60 python tools/ero_sweep.py 4096 1000000 "$w:$sl" 2>&1 | head -1
  done
done
done
) > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt | cut -c1-110
