#!/bin/bash
set -u
export TERRA_ERO_DIAG=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-eroclk2}
mkdir -p $OUT
cd $ROOT
(
echo "== 4096 1000000 auto"; timeout 120 python tools/ero_sweep.py 4096 1000000 "0:128" 2>&1 | tail -4
echo "== 4096 100000 auto"; timeout 120 python tools/ero_sweep.py 4096 100000 "0:128" 2>&1 | tail -4
) > $OUT/clk.txt 2>&1
cat $OUT/clk.txt | cut -c1-330
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/tools/ero_sweep.py 4096 300000 "0:128" > $OUT/stats.log 2>&1
cd $ROOT; python tools/summarize_rocprof.py $OUT/stats > $OUT/stats.txt 2>&1; find $OUT -name "*kernel_trace.csv" -size +1M -delete; head -8 $OUT/stats.txt
