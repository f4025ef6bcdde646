#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-psweep2}
mkdir -p $OUT
cd $ROOT
for k in 20 64; do
for p in 2 3 4 5 6 8; do
  for rep in 1 2; do
  timeout 300 python bench.py --pipelines $p --steps $k --warmup 3 --no-extras --no-cpu-baseline > $OUT/t.json 2>> $OUT/err.txt
  python -c "import json;d=json.load(open('$OUT/t.json'));print('steps $k P=$p',d['value'],d['ms_per_step'])"
  done
done
done
