#!/bin/bash
# (historical: ran against the commit that carried k_sine_grid_mx; results in profiles/r04_sine_matrix_pipe.txt)
# tools/gpu_round4b.sh <tag>: the second half of round 4 in one GPU call -- (1) what the matrix pipe can be for the sine sum (tools/mfma_products.hip), (2) the GPU suite
# on the default build and again with TERRA_SINE_PIPES=both (products of k_sine_grid from the matrix pipe), (3) the headline with either choice on the same box,
# kernel stats of both, (4) the dense-erosion timings of the final scheduler.
TAG=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
echo "== mfma_products"; timeout 120 tools/_bin/mfma_products 2>&1 | tee "$OUT/mfma_products.txt"
echo "== pytest default"; timeout 900 python -m pytest tests -m gpu -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"; tail -12 "$OUT/pytest_gpu.log"
echo "== pytest TERRA_SINE_PIPES=both"; TERRA_SINE_PIPES=both timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py tests/test_gpu_timed_sizes.py -m gpu -q -k "sine or grid or bench_step or minmax or strips or random_configs or proc_gen or golden or multi_contexts or streamed" > "$OUT/pytest_gpu_both.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu_both.log"; tail -12 "$OUT/pytest_gpu_both.log"
for pipes in valu both valu both; do
	echo "== bench $pipes"; TERRA_SINE_PIPES=$pipes timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/bench_${pipes}_line.json" 2> "$OUT/bench_${pipes}.err"; echo "rc $?"
	python - <<PY
import json
d = json.load(open("$OUT/bench_${pipes}_line.json")); print("$pipes", d["value"], d["ms_per_step"], d.get("latency_ms_single"), "grid kernel ms", d["detail"].get("ms_grid_kernel"), "noise kernels ms", d["detail"].get("ms_noise_kernels"), "frac", d["roofline"].get("frac"))
PY
	cp "$OUT/bench_${pipes}_line.json" "$OUT/bench_${pipes}_line_$(date +%s).json"
done
for pipes in valu both; do
	echo "== p1 kernel stats $pipes"
	(cd /tmp && TERRA_SINE_PIPES=$pipes timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_p1_$pipes" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --no-rccl-world1 --pipelines 1 > "$OUT/stats_p1_$pipes.log" 2>&1)
	python tools/summarize_rocprof.py "$OUT/stats_p1_$pipes" > "$OUT/p1_${pipes}_kernel_stats.txt" 2>&1; find "$OUT/stats_p1_$pipes" -name "*kernel_trace.csv" -size +2M -delete; head -8 "$OUT/p1_${pipes}_kernel_stats.txt"
done
echo "== clock"; TERRA_SINE_PIPES=both tools/gpu_job.sh clock $TAG/clock_both prof_driver.py 16384 3 2>&1 | head -6
echo "== erosion"; tools/gpu_job.sh erosion $TAG "16384 1000 0:0" "4096 100000 0:0" "4096 1000000 0:0" "8192 1000000 0:0" "16384 1000000 0:0" "1024 30000 0:0" "2048 1000000 0:0"
find "$OUT" -name "*.csv" -size +1M -delete
echo "== done"
