#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-eroprof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ero_dense -- python $ROOT/tools/ero_sweep.py 4096 200000 "2048:128" > $OUT/stats_ero_dense.log 2>&1
cd $ROOT
python tools/summarize_rocprof.py $OUT/stats_ero_dense > $OUT/stats_ero_dense.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete
head -14 $OUT/stats_ero_dense.txt; tail -3 $OUT/stats_ero_dense.log | cut -c1-200
