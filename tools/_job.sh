mkdir -p gpurun_out/r06o; O=$PWD/gpurun_out/r06o; export TMPDIR=/tmp; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -q -x -k "shadow or weights or tile_batch_64x64_eroded" 2>&1 | tail -4 > $O/tests.log
prof() { name=$1; shift; (cd /tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$name" -- "$@" > "$O/stats_$name.log" 2>&1); python "$ROOT/tools/summarize_rocprof.py" "$O/stats_$name" > "$O/${name}_kernel_stats.txt" 2>&1; find "$O/stats_$name" -name "*kernel_trace.csv" -size +2M -delete; head -5 "$O/${name}_kernel_stats.txt"; }
prof weights python $ROOT/tools/prof_weights.py 4
prof shadows python $ROOT/tools/prof_shadows.py 3
for f in 0 1 2; do TERRA_GEN_FUSED=$f python tools/prof_voxels.py 512 64 2>&1 | tail -1; done
cat $O/tests.log
