#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-psweep}
mkdir -p $OUT
cd $ROOT
for p in 2 4 6 8 12; do
  timeout 300 python bench.py --pipelines $p --steps 48 --warmup 8 --no-extras --no-cpu-baseline > $OUT/p$p.json 2> $OUT/p$p.err
  python -c "import json;d=json.load(open('$OUT/p$p.json'));print('P=$p',d['value'],d['ms_per_step'])"
done
for p in 4 8; do
  timeout 300 python bench.py --pipelines $p --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/p${p}_20.json 2> $OUT/p${p}_20.err
  python -c "import json;d=json.load(open('$OUT/p${p}_20.json'));print('P=$p steps 20',d['value'],d['ms_per_step'])"
done
