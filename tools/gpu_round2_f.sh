#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02f}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ero_dense -- python $ROOT/tools/ero_sweep.py 4096 200000 "2048:1024" > $OUT/stats_ero_dense.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_tiles -- python $ROOT/tools/prof_tiles.py 0 3 > $OUT/stats_tiles.log 2>&1
cd $ROOT
python tools/summarize_rocprof.py $OUT/stats_ero_dense > $OUT/stats_ero_dense.txt 2>&1
python tools/summarize_rocprof.py $OUT/stats_tiles > $OUT/stats_tiles.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +3M -delete
cat $OUT/stats_ero_dense.log | tail -2; head -24 $OUT/stats_ero_dense.txt; head -16 $OUT/stats_tiles.txt
