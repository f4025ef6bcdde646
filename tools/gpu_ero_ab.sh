#!/bin/bash
set -u
export TERRA_ERO_DIAG=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-eroab}
mkdir -p $OUT
cd $ROOT
(
for cfg in "16384 1000000 0:128" "16384 100000 0:128" "8192 1000000 0:128"; do
  set -- $cfg
  echo "== A $1 $2"; (cd $ROOT/tmp_ab/A && TERRA_ERO_NEAR=512 timeout 120 python tools/ero_sweep.py $1 $2 "$3" 2>&1 | tail -4)
  echo "== B $1 $2"; (cd $ROOT && timeout 120 python tools/ero_sweep.py $1 $2 "$3" 2>&1 | tail -4)
done
) > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-230
