#!/bin/bash
# what the driver does at round end: the GPU parity suite, smoke(), the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-final}
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_driver_flags.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open("$OUT/bench_driver_flags.json"))
print({k:d[k] for k in ("value","ms_per_step","latency_ms_single")})
print("tiles", d["detail"]["tiles"]["erosion_0"]["ms_per_batch"], d["detail"]["tiles"]["erosion_1000"]["ms_per_batch"], "voxels", d["detail"]["voxels"]["gvoxels_s"])
print("erosion", d["detail"]["erosion"])
PY
