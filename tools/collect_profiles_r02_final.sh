#!/bin/bash
# Run on the GPU box (gpurun): refresh of round 2's evidence after the erosion rework (block-page versions, 8 launches per round) and the Perlin block records.
set -u
R=${1:-r02f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $ROOT && timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_driver_flags_line.json 2> $OUT/bench_driver_flags.err)
(cd $ROOT && timeout 900 python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err)
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/stats_default.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p1 -- python $ROOT/bench.py --no-cpu-baseline --no-extras --pipelines 1 > $OUT/stats_p1.log 2>&1
(
for cfg in "4096 1000000 2048:128" "4096 100000 2048:128" "16384 1000000 0:128" "1024 30000 0:128" "16384 1000 0:128"; do
  set -- $cfg
  echo "== $1 $2 ($3), TERRA_ERO_DIAG=1"; (cd $ROOT && TERRA_ERO_DIAG=1 timeout 120 python tools/ero_sweep.py $1 $2 "$3" 2>&1 | tail -4)
done
) > $OUT/erosion_clock_breakdown.txt 2>&1
(
for cfg in "4096 1000" "4096 100000" "4096 1000000" "8192 1000000" "16384 1000000" "1024 30000" "16384 1000"; do set -- $cfg; echo "== $1 $2, automatic ring, no diagnostics"; (cd $ROOT && timeout 120 python tools/ero_sweep.py $1 $2 "0:128" 2>&1 | head -1); done
) > $OUT/erosion_timings.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ero_dense -- python $ROOT/tools/ero_sweep.py 4096 200000 "0:128" > $OUT/stats_ero_dense.log 2>&1
timeout 300 python $ROOT/tools/prof_noise.py 4096 5 1,2,4 > $OUT/noise_4096_after.txt 2>&1
timeout 300 python $ROOT/tools/prof_noise.py 16384 2 1,2,4,0 > $OUT/noise_16384_after.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_noise16384 -- python $ROOT/tools/prof_noise.py 16384 2 1,2,4 > $OUT/stats_noise16384.log 2>&1
cd $ROOT
for s in stats_default stats_p1 stats_noise16384 stats_ero_dense; do python tools/summarize_rocprof.py $OUT/$s > $OUT/$s.txt 2>&1; done
timeout 600 python tools/bench_extra.py > $OUT/bench_extra.json 2> $OUT/bench_extra.err
find $OUT -name "*kernel_trace.csv" -size +1M -delete
tail -c 300 $OUT/bench_default_line.json; echo; head -6 $OUT/stats_p1.txt; cat $OUT/erosion_timings.txt | cut -c1-110; cat $OUT/noise_4096_after.txt $OUT/noise_16384_after.txt
