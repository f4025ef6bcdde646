#!/bin/bash
# tools/gpu_round4h.sh <tag>: bench.py --build-ahead 0 / 1 (tables of the next map built while the pipeline waits for its noise turn), same box, alternating
TAG=${1:-r04h}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], "Gcells/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3 4; do for ba in 0 1; do
	timeout 40 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 --build-ahead $ba > "$OUT/b_${ba}_$rep.json" 2> "$OUT/b.err"; line "$OUT/b_${ba}_$rep.json" "build-ahead $ba K20"
done; done | tee "$OUT/ab_build_ahead.txt"
for ba in 0 1 0 1; do
	timeout 40 python bench.py --steps 64 --warmup 8 --headline-only --no-cpu-baseline --no-rccl-world1 --build-ahead $ba > "$OUT/b64_${ba}.json" 2> "$OUT/b.err"; line "$OUT/b64_${ba}.json" "build-ahead $ba K64"
done | tee -a "$OUT/ab_build_ahead.txt"
echo "== timeline"; tools/gpu_job.sh timeline $TAG/tl --steps 20 --warmup 5 > "$OUT/timeline.log" 2>&1; head -12 "$OUT/tl/timeline.txt"
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "build_arrays or gen_grid_minmax or grid_vs or knobs" 2>&1 | tail -3
find "$OUT" -name "*.csv" -size +1M -delete
echo "== done"
