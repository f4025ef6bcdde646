#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-erocks}
mkdir -p $OUT
cd $ROOT
(
for ck in 64:8 48:12 32:16 24:16 16:16 64:0; do
  for w in 2048 4096; do
    echo "== CK $ck W $w"; TERRA_ERO_CK=$ck timeout 60 python tools/ero_sweep.py 4096 1000000 "$w:128" 2>&1 | head -1
  done
done
for ck in 64:8 32:16 64:0; do echo "== 16384 1e6 CK $ck"; TERRA_ERO_CK=$ck timeout 60 python tools/ero_sweep.py 16384 1000000 "0:128" 2>&1 | head -1; done
for ck in 64:8 32:16 64:0; do echo "== 16384 1000 CK $ck"; TERRA_ERO_CK=$ck timeout 60 python tools/ero_sweep.py 16384 1000 "0:128" 2>&1 | head -1; done
) > $OUT/cksweep.txt 2>&1
cat $OUT/cksweep.txt | cut -c1-130
