#!/bin/bash
# tools/gpu_round4m.sh: where the pipeline's ~0.2 ms per step beyond the bare noise kernel come from: the same loop WITHOUT erosion (--droplets 0: apply_erosion returns at once)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04m; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for rep in 1 2; do for D in 1000 0; do for S in 1 0; do for P in 1 4; do
	timeout 60 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 --droplets $D --noise-slots $S --pipelines $P 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('droplets $D slots $S P $P K20', d['value'], d['ms_per_step'])"
done; done; done; done 2>&1 | tee "$OUT/no_erosion.txt"
