#!/bin/bash
# tools/gpu_round4c.sh <tag>: k_sine_grid with 24 operand registers / smaller LDS chunks (two of its waves beside a droplet wave on a SIMD) against the build before it
# (tools/_ab/base, built by tools/ab_build.sh from the previous commit), same box, alternating: headline at the driver's flags, kernel alone, pipelines sweep, timeline; then the GPU suite.
TAG=${1:-r04c}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], "Gcells/s", d["ms_per_step"], "ms/step; grid kernel", d["detail"].get("ms_grid_kernel"), "noise kernels", d["detail"].get("ms_noise_kernels"), "erosion", d["detail"].get("ms_erosion"))
PY
}
for rep in 1 2; do
	for v in base kc27 kc20 kc45; do
		case $v in base) export TERRA_LIB=$ROOT/tools/_ab/base/libterra_hip.so; unset TERRA_SG_KC ;; kc27) unset TERRA_LIB; export TERRA_SG_KC=27 ;; kc20) unset TERRA_LIB; export TERRA_SG_KC=20 ;; kc45) unset TERRA_LIB; export TERRA_SG_KC=45 ;; esac
		timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/bench_${v}_$rep.json" 2> "$OUT/bench_${v}_$rep.err"; line "$OUT/bench_${v}_$rep.json" "$v K20"
	done
done | tee "$OUT/ab_headline.txt"
for v in base kc27; do
	case $v in base) export TERRA_LIB=$ROOT/tools/_ab/base/libterra_hip.so; unset TERRA_SG_KC ;; kc27) unset TERRA_LIB; export TERRA_SG_KC=27 ;; esac
	timeout 300 python bench.py --steps 64 --warmup 8 --no-extras --no-cpu-baseline > "$OUT/bench_${v}_k64.json" 2> "$OUT/bench_${v}_k64.err"; line "$OUT/bench_${v}_k64.json" "$v K64"
done | tee -a "$OUT/ab_headline.txt"
unset TERRA_LIB; unset TERRA_SG_KC
for P in 3 5 6; do timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pipelines $P > "$OUT/bench_kc27_P$P.json" 2> "$OUT/bench_P$P.err"; line "$OUT/bench_kc27_P$P.json" "kc27 P$P K20"; done | tee -a "$OUT/ab_headline.txt"
echo "== p1 kernel stats (new default)"
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_p1" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --no-rccl-world1 --pipelines 1 > "$OUT/stats_p1.log" 2>&1)
python tools/summarize_rocprof.py "$OUT/stats_p1" > "$OUT/bench_pipelines1_kernel_stats.txt" 2>&1; find "$OUT/stats_p1" -name "*kernel_trace.csv" -size +2M -delete; head -8 "$OUT/bench_pipelines1_kernel_stats.txt"
echo "== timeline (new default)"; tools/gpu_job.sh timeline $TAG/tl > "$OUT/timeline.log" 2>&1; head -12 "$OUT/tl/timeline.txt" 2>/dev/null || head -12 "$ROOT/gpurun_out/$TAG/tl/timeline.txt"
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"; tail -6 "$OUT/pytest_gpu.log"
TERRA_SG_KC=45 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sine or grid_vs or minmax or strips or random_configs" > "$OUT/pytest_gpu_kc45.log" 2>&1; echo "pytest kc45 rc $?"; tail -2 "$OUT/pytest_gpu_kc45.log"
TERRA_SG_KC=20 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sine or grid_vs or minmax or strips or random_configs" > "$OUT/pytest_gpu_kc20.log" 2>&1; echo "pytest kc20 rc $?"; tail -2 "$OUT/pytest_gpu_kc20.log"
find "$OUT" -name "*.csv" -size +1M -delete
echo "== done"
