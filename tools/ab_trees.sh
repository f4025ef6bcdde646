#!/bin/bash
# tools/ab_trees.sh <tag> <other tree> <reps>: the headline of THIS tree against another checkout of the repository (e.g. `git worktree add tmp_r04 <round-4 commit>`, library
# built there) on the same box, alternating: bench.py --headline-only at the driver's flags, over 64 steps, and with one heightmap in flight.
TAG=${1:-ab}; OTHER=${2:-tmp_r04}; REPS=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/ab_trees.txt"
one() { # one <name> <dir> <args...>
	local name=$1 dir=$2; shift; shift
	local line; line=$(cd "$dir" && timeout 90 python bench.py --headline-only --no-cpu-baseline --no-rccl-world1 "$@" 2>> "$OUT/ab_trees.err" | tail -1)
	echo "$name | $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['value'], 'Gcells/s', d['ms_per_step'], 'ms/step')" "$line" 2>/dev/null || echo FAILED)" | tee -a "$OUT/ab_trees.txt"
}
for rep in $(seq 1 "$REPS"); do
	one "this  K20" "$ROOT" --steps 20 --warmup 5;        one "other K20" "$ROOT/$OTHER" --steps 20 --warmup 5
	one "this  K64" "$ROOT" --steps 64 --warmup 8;        one "other K64" "$ROOT/$OTHER" --steps 64 --warmup 8
	one "this  P1 " "$ROOT" --steps 16 --warmup 4 --pipelines 1; one "other P1 " "$ROOT/$OTHER" --steps 16 --warmup 4 --pipelines 1
done
