#!/bin/bash
# Run on the GPU box (gpurun): the round's evidence under gpurun_out/$1 (default r01).  Copy what should be judged into profiles/.
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (with the CPU baseline)
(cd $ROOT && timeout 600 python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err)
# 2. kernel stats of the same command (4 heightmaps in flight: kernels of different streams overlap) and of a single stream
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python $ROOT/bench.py --no-cpu-baseline > $OUT/stats_default.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p1 -- python $ROOT/bench.py --no-cpu-baseline --pipelines 1 > $OUT/stats_p1.log 2>&1
# 3. counters, one pass each (no trace domains besides --kernel-trace)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $ROOT/tools/prof_driver.py 16384 2 > $OUT/pmc_$c.log 2>&1
done
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python $ROOT/tools/prof_driver.py 16384 2 > $OUT/pmc_sq.log 2>&1
cd $ROOT
python tools/summarize_rocprof.py $OUT/stats_default > $OUT/stats_default.txt 2>&1
python tools/summarize_rocprof.py $OUT/stats_p1 > $OUT/stats_p1.txt 2>&1
for k in k_sine_grid k_waves k_minmax quantize16; do
  for c in FETCH_SIZE WRITE_SIZE sq; do echo "== $c $k"; python tools/pmc_summary.py $OUT/pmc_$c "$k" 2>&1 | tail -4; done
done > $OUT/pmc_summary.txt
python tools/bench_extra.py > $OUT/bench_extra.json 2> $OUT/bench_extra.err
# keep the merge small: drop the raw traces, keep stats
find $OUT -name "*kernel_trace.csv" -size +3M -delete
tail -c 600 $OUT/bench_default_line.json; echo; head -12 $OUT/stats_p1.txt; cat $OUT/pmc_summary.txt | head -40
