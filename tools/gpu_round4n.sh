#!/bin/bash
# tools/gpu_round4n.sh: bench.py --schedule producer (ONE thread issues every map's noise in sequence, eroder threads take the maps over) against the threads schedule, same box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04n; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
python - <<'PY'
import importlib, sys, numpy as np
sys.path.insert(0, "."); pkg = importlib.import_module("3dworld_amd")
a, b = pkg.Terra(0), pkg.Terra(0)
N = 2048
for t in (a, b):
    st = t.init_scene(pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
z1, z2 = a.alloc(N * N * 4), a.alloc(N * N * 4)
mn, _ = a.gen_grid_minmax_dev(z1.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE); a.apply_erosion_dev(z1.ptr, N, N, mn, 3000, pkg.ERODE_MINZ_IS_MIN); a.synchronize()
mn, _ = b.gen_grid_minmax_dev(z2.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE); a.apply_erosion_dev(z2.ptr, N, N, mn, 3000, pkg.ERODE_MINZ_IS_MIN); a.synchronize()
x, y = z1.download(np.uint32, (N, N)), z2.download(np.uint32, (N, N))
print("noise on one context, erosion on another: identical to one context:", bool((x == y).all()))
PY
for rep in 1 2 3; do
	timeout 60 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('threads slots 1 P 4 K20', d['value'], d['ms_per_step'])"
	for P in 2 3 4 6; do
		timeout 60 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 --schedule producer --pipelines $P 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('producer P $P K20', d['value'], d['ms_per_step'])"
	done
done 2>&1 | tee "$OUT/ab_producer.txt"
for v in "--schedule threads" "--schedule producer --pipelines 3" "--schedule producer --pipelines 4"; do
	timeout 60 python bench.py --steps 64 --warmup 8 --headline-only --no-cpu-baseline --no-rccl-world1 $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v K64', d['value'], d['ms_per_step'])"
done 2>&1 | tee -a "$OUT/ab_producer.txt"
