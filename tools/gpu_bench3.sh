#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-bench3}
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-extras > $OUT/b$i.json 2>> $OUT/err.txt; python -c "import json;d=json.load(open('$OUT/b$i.json'));print(d['value'],d['ms_per_step'],d.get('latency_ms_single'))"; done
timeout 300 python bench.py --steps 64 --warmup 3 --no-extras > $OUT/b64.json 2>> $OUT/err.txt; python -c "import json;d=json.load(open('$OUT/b64.json'));print('64 steps',d['value'],d['ms_per_step'])"
