#!/bin/bash
# tools/gpu_round4k.sh <tag>: bench.py --noise-split F (a map's noise as two row strips, the second outside the noise turn), same box, alternating
TAG=${1:-r04k}
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
for rep in 1 2 3; do for F in 0 0.75 0.85 0.92; do
	timeout 40 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 --noise-split $F > "$OUT/b_${F}_$rep.json" 2> "$OUT/b.err"; line "$OUT/b_${F}_$rep.json" "split $F K20"
done; done | tee "$OUT/ab_split.txt"
for F in 0 0.85 0 0.85; do
	timeout 40 python bench.py --steps 64 --warmup 8 --headline-only --no-cpu-baseline --no-rccl-world1 --noise-split $F > "$OUT/b64_${F}.json" 2> "$OUT/b.err"; line "$OUT/b64_${F}.json" "split $F K64"
done | tee -a "$OUT/ab_split.txt"
echo "== timeline split 0.85"; tools/gpu_job.sh timeline $TAG/tl --steps 20 --warmup 5 --noise-split 0.85 > "$OUT/timeline.log" 2>&1; head -12 "$OUT/tl/timeline.txt"
find "$OUT" -name "*.csv" -size +1M -delete
echo "== done"
