#!/bin/bash
# same-box A/B of the headline: tmp_ab/A (an older commit, built in place) against the working tree, interleaved
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-benchab2}
mkdir -p $OUT
cd $ROOT
one() { (cd $2 && timeout 300 python bench.py --steps ${3:-20} --warmup 3 --no-extras > $OUT/t.json 2>> $OUT/err.txt); python -c "import json;d=json.load(open('$OUT/t.json'));print('$1',d['value'],d['ms_per_step'])"; }
for i in 1 2 3; do one A $ROOT/tmp_ab/A; one B $ROOT; done
one A64 $ROOT/tmp_ab/A 64; one B64 $ROOT 64
