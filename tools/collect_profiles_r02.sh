#!/bin/bash
# Run on the GPU box (gpurun): round 2's evidence under gpurun_out/r02.  Copy what should be judged into profiles/.
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line: as the driver runs it (--steps 20 --warmup 3) and with the defaults
(cd $ROOT && timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_driver_flags_line.json 2> $OUT/bench_driver_flags.err)
(cd $ROOT && timeout 900 python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err)
# 2. kernel stats of the headline loop: 4 heightmaps in flight (kernels of different streams overlap) and a single stream
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/stats_default.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p1 -- python $ROOT/bench.py --no-cpu-baseline --no-extras --pipelines 1 > $OUT/stats_p1.log 2>&1
# 3. counters, one pass each (no trace domains besides --kernel-trace)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $ROOT/tools/prof_driver.py 16384 2 > $OUT/pmc_$c.log 2>&1
done
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq -- python $ROOT/tools/prof_driver.py 16384 2 > $OUT/pmc_sq.log 2>&1
# 4. the fBm kernels (table-driven): stats + SQ + LDS counters at 4096^2 and 16384^2, HBM counters at 16384^2
timeout 300 python $ROOT/tools/prof_noise.py 4096 5 1,2,4 > $OUT/noise_4096_after.txt 2>&1
timeout 300 python $ROOT/tools/prof_noise.py 16384 2 1,2,4,0 > $OUT/noise_16384_after.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_noise4096 -- python $ROOT/tools/prof_noise.py 4096 3 1,2,4 > $OUT/stats_noise4096.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_noise16384 -- python $ROOT/tools/prof_noise.py 16384 2 1,2,4 > $OUT/stats_noise16384.log 2>&1
for n in 4096 16384; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq_noise$n -- python $ROOT/tools/prof_noise.py $n 2 1,2,4 > $OUT/pmc_sq_noise$n.log 2>&1
done
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds_noise4096 -- python $ROOT/tools/prof_noise.py 4096 2 1,2,4 > $OUT/pmc_lds_noise4096.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_noise16384 -- python $ROOT/tools/prof_noise.py 16384 1 1,2,4 > $OUT/pmc_${c}_noise16384.log 2>&1
done
# 5. tiles (zvals + stats + normals, AO, shadows), voxels, dense erosion: kernel stats
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_tiles -- python $ROOT/tools/prof_tiles.py 0 3 > $OUT/stats_tiles.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_tiles_simplex -- python $ROOT/tools/prof_tiles.py 1 2 > $OUT/stats_tiles_simplex.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_voxels -- python $ROOT/tools/prof_voxels.py > $OUT/stats_voxels.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ero_dense -- python $ROOT/tools/ero_sweep.py 4096 200000 "2048:1024" > $OUT/stats_ero_dense.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq_tiles -- python $ROOT/tools/prof_tiles.py 0 2 > $OUT/pmc_sq_tiles.log 2>&1
cd $ROOT
for s in stats_default stats_p1 stats_noise4096 stats_noise16384 stats_tiles stats_tiles_simplex stats_voxels stats_ero_dense; do python tools/summarize_rocprof.py $OUT/$s > $OUT/$s.txt 2>&1; done
for k in k_sine_grid k_waves k_minmax k_quantize16; do
  for c in FETCH_SIZE WRITE_SIZE sq; do echo "== $c $k"; python tools/pmc_summary.py $OUT/pmc_$c "$k" 2>&1 | tail -4; done
done > $OUT/pmc_summary.txt
for n in 4096 16384; do for k in "k_noise_grid<1>" "k_noise_grid<2>" "k_noise_grid<4>"; do echo "== sq $n $k"; python tools/pmc_summary.py $OUT/pmc_sq_noise$n "$k" 2>&1 | tail -3; done; done > $OUT/pmc_noise_summary.txt
for k in "k_noise_grid<1>" "k_noise_grid<2>" "k_noise_grid<4>"; do echo "== lds 4096 $k"; python tools/pmc_summary.py $OUT/pmc_lds_noise4096 "$k" 2>&1 | tail -3 | head -2; for c in FETCH_SIZE WRITE_SIZE; do echo "== $c 16384 $k"; python tools/pmc_summary.py $OUT/pmc_${c}_noise16384 "$k" 2>&1 | tail -3 | head -2; done; done >> $OUT/pmc_noise_summary.txt
for k in k_tile_post k_tile_ao "k_sine_grid<true" k_tile_shadows_level; do echo "== sq tiles $k"; python tools/pmc_summary.py $OUT/pmc_sq_tiles "$k" 2>&1 | tail -3; done > $OUT/pmc_tiles_summary.txt
python tools/make_pmc_traffic.py $OUT 16384 > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
python tools/bench_extra.py > $OUT/bench_extra.json 2> $OUT/bench_extra.err
# keep the merge small: drop the raw traces, keep stats
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT -name "*counter_collection.csv" -size +3M -delete
tail -c 400 $OUT/bench_default_line.json; echo; head -8 $OUT/stats_p1.txt; cat $OUT/noise_4096_after.txt $OUT/noise_16384_after.txt; cat $OUT/pmc_traffic.json | head -30
