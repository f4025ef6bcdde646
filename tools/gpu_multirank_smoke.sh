#!/bin/bash
# two ranks on ONE GPU over gloo: the multi-rank orchestration of bench.py (weak headline, strips + all_reduce(min), tile partition, voxel slabs) and the gpu-marked distributed tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-mr}
mkdir -p $OUT
cd $ROOT
TERRA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 2 --size 4096 --no-cpu-baseline > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "bench2 rc $?"
python - <<PY
import json
d=json.load(open("$OUT/bench_2rank_gloo.json"))
print({k:d[k] for k in ("value","n_gpus","scaling","ms_per_step")})
print(d["detail"]["strips"]); print(d["detail"]["tiles"]["erosion_0"], d["detail"]["tiles"]["tiles_per_rank"]); print(d["detail"]["voxels"])
PY
for w in strips tiles; do TERRA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 8 --warmup 2 --workload $w --no-cpu-baseline --no-extras > $OUT/bench_2rank_$w.json 2> $OUT/bench_2rank_$w.err; echo "$w rc $?"; python -c "import json;d=json.load(open('$OUT/bench_2rank_$w.json'));print(d['metric'],d['value'],d['scaling'],d['n_gpus'])"; done
