#!/bin/bash
# tools/gpu_round4g.sh <tag>: the streamed (enqueue-only) schedule again, now that two noise waves fit beside a droplet wave on a SIMD
TAG=${1:-r04g}
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
for rep in 1 2; do
	timeout 40 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 > "$OUT/t_$rep.json" 2> "$OUT/b.err"; line "$OUT/t_$rep.json" "threads slots 1 P 4 K20"
	for pr in none erosion-high noise-low; do for P in 3 4; do
		timeout 40 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 --schedule streamed --priorities $pr --pipelines $P > "$OUT/s_${pr}_${P}_$rep.json" 2> "$OUT/b.err"; line "$OUT/s_${pr}_${P}_$rep.json" "streamed $pr P $P K20"
	done; done
done | tee "$OUT/ab_streamed.txt"
echo "== timeline streamed erosion-high P 4"; tools/gpu_job.sh timeline $TAG/tl --steps 20 --warmup 5 --schedule streamed --priorities erosion-high --pipelines 4 > "$OUT/timeline.log" 2>&1; head -14 "$OUT/tl/timeline.txt"
find "$OUT" -name "*.csv" -size +1M -delete
echo "== done"
