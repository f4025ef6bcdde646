#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-eroq}
mkdir -p $OUT
cd $ROOT
(
for cfg in "4096 1000000" "4096 100000" "16384 1000000" "8192 1000000" "1024 30000" "16384 1000"; do set -- $cfg; echo "== $1 $2 auto"; timeout 120 python tools/ero_sweep.py $1 $2 "0:128" 2>&1 | head -1; done
) > $OUT/quick.txt 2>&1
cat $OUT/quick.txt | cut -c1-120
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/tools/ero_sweep.py 4096 300000 "0:128" > $OUT/stats.log 2>&1
cd $ROOT; f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r['Name'][-40:], r['Calls'], round(float(r['AverageNs'])/1000,1), round(float(r['MaxNs'])/1000,1))
PY
find $OUT -name "*kernel_trace.csv" -size +1M -delete
