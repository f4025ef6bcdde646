#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-benchonce}
mkdir -p $OUT
cd $ROOT
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "rc $?"; echo "bench took $SECONDS s"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["latency_ms_single"]); print(d["detail"]["dense_erosion"]); print(d["detail"]["ms_erosion"])
PY
