#!/bin/bash
# tools/collect_if_typical.sh <tag> <min Gcells/s>: boxes differ by +-5 %; take the round's evidence set only on a box whose headline (driver's flags) reaches the threshold
TAG=${1:-r05y}; MIN=${2:-288}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/$TAG
timeout 300 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 > gpurun_out/$TAG/probe.json 2> gpurun_out/$TAG/probe.err
V=$(python -c "import json; print(json.load(open('gpurun_out/$TAG/probe.json'))['value'])" 2>/dev/null || echo 0)
echo "probe: $V Gcells/s (threshold $MIN)"
if python -c "import sys; sys.exit(0 if float('$V') >= float('$MIN') else 1)"; then
	bash tools/gpu_job.sh check $TAG > gpurun_out/${TAG}_check.log 2>&1; tail -4 gpurun_out/${TAG}_check.log | cut -c1-200
	bash tools/collect_round.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -3 gpurun_out/${TAG}_collect.log | cut -c1-200
	echo "collected"
else
	echo "slow box: nothing collected"
fi
