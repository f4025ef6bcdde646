#!/bin/bash
# tools/gpu_round4d.sh <tag>: bench.py --noise-slots S (how many heightmaps may be in their noise phase at once) x pipelines P, same box, alternating; timeline of the default
TAG=${1:-r04d}
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
for rep in 1 2 3; do for S in 0 1 2; do for P in 4 5 6; do
	timeout 60 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 --pipelines $P --noise-slots $S > "$OUT/b_${S}_${P}_$rep.json" 2> "$OUT/b.err"; line "$OUT/b_${S}_${P}_$rep.json" "slots $S P $P K20"
done; done; done | tee "$OUT/ab_slots.txt"
for S in 0 1 2; do for P in 4 6; do
	timeout 60 python bench.py --steps 64 --warmup 8 --headline-only --no-cpu-baseline --no-rccl-world1 --pipelines $P --noise-slots $S > "$OUT/b64_${S}_${P}.json" 2> "$OUT/b.err"; line "$OUT/b64_${S}_${P}.json" "slots $S P $P K64"
done; done | tee -a "$OUT/ab_slots.txt"
echo "== done"
