mkdir -p gpurun_out/r06p; O=$PWD/gpurun_out/r06p; export TMPDIR=/tmp; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -q -x -k "weights or fused_tiles or fast_mode_small" 2>&1 | tail -4 > $O/tests.log
prof() { name=$1; shift; (cd /tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$name" -- "$@" > "$O/stats_$name.log" 2>&1); python "$ROOT/tools/summarize_rocprof.py" "$O/stats_$name" > "$O/${name}_kernel_stats.txt" 2>&1; find "$O/stats_$name" -name "*kernel_trace.csv" -size +2M -delete; head -5 "$O/${name}_kernel_stats.txt"; }
prof weights python $ROOT/tools/prof_weights.py 4
cat $O/tests.log
