mkdir -p gpurun_out/r06k; O=$PWD/gpurun_out/r06k; export TMPDIR=/tmp; ROOT=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -q -x -k "weights or tile_batch_64x64_eroded" 2>&1 | tail -6 > $O/tests.log
prof() { name=$1; shift; (cd /tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$name" -- "$@" > "$O/stats_$name.log" 2>&1); python "$ROOT/tools/summarize_rocprof.py" "$O/stats_$name" > "$O/${name}_kernel_stats.txt" 2>&1; find "$O/stats_$name" -name "*kernel_trace.csv" -size +2M -delete; head -12 "$O/${name}_kernel_stats.txt"; }
prof weights python $ROOT/tools/prof_weights.py 5
TERRA_WEIGHTS_SIMPLE=1 python $ROOT/tools/prof_weights.py 3 > $O/simple.log 2>&1
cat $O/tests.log; grep weights $O/stats_weights.log $O/simple.log
