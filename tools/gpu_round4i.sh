#!/bin/bash
# tools/gpu_round4i.sh <tag>: k_sine_grid's staging (all loads of a table's chunk in flight, stride addressing) against the build before it (tools/_ab/prev), same box
TAG=${1:-r04i}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], "Gcells/s", d["ms_per_step"], "ms/step; grid kernel", d["detail"].get("ms_grid_kernel"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do for v in prev new; do
	if [ $v = prev ]; then export TERRA_LIB=$ROOT/tools/_ab/prev/libterra_hip.so; else unset TERRA_LIB; fi
	timeout 60 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/b_${v}_$rep.json" 2> "$OUT/b.err"; line "$OUT/b_${v}_$rep.json" "$v K20"
done; done | tee "$OUT/ab_staging.txt"
unset TERRA_LIB
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py -m gpu -q -k "sine or grid_vs or minmax or strips or random_configs or knobs or bench_step or build_arrays or golden" 2>&1 | tail -3
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_p1" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --no-rccl-world1 --pipelines 1 > "$OUT/stats_p1.log" 2>&1)
python tools/summarize_rocprof.py "$OUT/stats_p1" > "$OUT/bench_pipelines1_kernel_stats.txt" 2>&1; find "$OUT/stats_p1" -name "*kernel_trace.csv" -size +2M -delete; head -4 "$OUT/bench_pipelines1_kernel_stats.txt"
echo "== done"
