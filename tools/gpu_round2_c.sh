#!/bin/bash
# round 2: the new bench line (driver flags), the 2-rank orchestration on one GPU (gloo), the full-size oracle parity tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02c}
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err; echo "bench rc $?"
tail -c 3000 $OUT/bench_driver_flags.json; tail -5 $OUT/bench_driver_flags.err
TERRA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 2 --size 4096 --no-cpu-baseline > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "bench2 rc $?"
tail -c 1500 $OUT/bench_2rank_gloo.json; tail -5 $OUT/bench_2rank_gloo.err
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_step or whole_grid" > $OUT/pytest_fullsize.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_fullsize.log
tail -5 $OUT/pytest_fullsize.log
