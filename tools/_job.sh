mkdir -p gpurun_out/r06n; O=$PWD/gpurun_out/r06n; export TMPDIR=/tmp; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -q -x -k "shadow or tile_batch_64x64_eroded" 2>&1 | tail -6 > $O/tests.log
timeout 300 python tools/prof_shadows.py 4 > $O/flow.log 2>&1
cat $O/tests.log $O/flow.log
