#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-sinetry}
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/prof_noise.py 16384 3 0 2>&1 | tail -2
timeout 300 python tools/prof_noise.py 4096 5 0 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/t.json 2>> $OUT/err.txt; python -c "import json;d=json.load(open('$OUT/t.json'));print('20',d['value'],d['ms_per_step'])"; done
timeout 300 python bench.py --steps 64 --warmup 3 --no-extras --no-cpu-baseline > $OUT/t.json 2>> $OUT/err.txt; python -c "import json;d=json.load(open('$OUT/t.json'));print('64',d['value'],d['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or grid or bench_step or tiles or full_size_sine" 2>&1 | tail -2
