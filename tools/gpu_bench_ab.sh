#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-benchab}
mkdir -p $OUT
cd $ROOT
run() { for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-extras > $OUT/t.json 2>> $OUT/err.txt; python -c "import json;d=json.load(open('$OUT/t.json'));print('$1',d['value'],d['ms_per_step'])"; done; }
run default
TERRA_ERO_BATCH=1 run batch1
TERRA_ERO_BATCH=4 run batch4
