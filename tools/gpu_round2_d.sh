#!/bin/bash
# round 2: full parity suite after the tile_post / sine_k / bench changes + bench with driver flags
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02d}
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open("$OUT/bench_driver_flags.json"))
print({k:d[k] for k in ("value","ms_per_step","latency_ms_single")})
print("tiles", d["detail"]["tiles"]["erosion_0"], d["detail"]["tiles"]["erosion_1000"])
print("strips", d["detail"]["strips"]["gcells_s"], "modes", {k:v["gcells_s_noise_only"] for k,v in d["detail"]["modes"].items()})
print({k:d["detail"][k] for k in ("ms_noise_kernels","ms_grid_kernel","ms_erosion")})
PY
timeout 300 python bench.py --steps 64 --warmup 8 --no-extras --no-cpu-baseline > $OUT/bench_64.json 2> $OUT/bench_64.err
python -c "import json;d=json.load(open('$OUT/bench_64.json'));print('64 steps',d['value'],d['ms_per_step'])"
