#!/bin/bash
# tools/gpu_job.sh -- the one GPU-box job script (run through gpurun).  Everything lands in gpurun_out/<tag>/; summaries worth keeping are copied to profiles/ by hand.
#   gpu_job.sh check   <tag>                         what the driver does at round end: pytest -m gpu, smoke(), the bench line with the driver's flags
#   gpu_job.sh tests   <tag> [pytest args]           pytest -m gpu with extra arguments (e.g. "tests/test_gpu_at_size.py -k tile")
#   gpu_job.sh bench   <tag> [bench.py args]         one bench.py line (+ a second one with --pipelines 1 --no-extras when no args are given)
#   gpu_job.sh profile <tag> [what ...]              rocprofv3 --kernel-trace --stats summaries; what = bench | p1 | tiles | tile_erosion | weights | ao | voxels | voxels64 | noise | erosion | shadows | fused | fast (default: the first nine)
#   gpu_job.sh pmc     <tag> <driver.py args> -- <kernel substr ...>   FETCH_SIZE, WRITE_SIZE and the SQ set, one --pmc pass each, over tools/<driver>
#   gpu_job.sh erosion <tag> "<grid> <droplets> <W:slice[,W:slice..]>" ...   dense-erosion timings (tools/ero_sweep.py); TERRA_ERO_* knobs pass through the environment
#   gpu_job.sh multirank <tag>                       2 ranks on one GPU over gloo: bench.py self-launch + the sharded workloads
#   gpu_job.sh stepcost <tag>                        ns per droplet step on one wave (tools/step_cost.hip) + the 64x64x1000 tile batch
#   gpu_job.sh native  <tag> [args]                  tools/bench_native.c and tools/bench_native_multi.c (--same-device) built and run
#   gpu_job.sh timeline <tag> [bench.py args]        rocprofv3 kernel trace of the headline run with the driver's flags -> tools/timeline.py (who ran when, idle time, overlap)
#   gpu_job.sh clock   <tag> <driver.py args>        GRBM_GUI_ACTIVE per dispatch: the clock a kernel actually ran at (cycles / duration)
#   gpu_job.sh ab      <tag> <reps> "name|ENV=..|bench args" ...   alternating same-box A/B of the headline under environment / argument variants
set -u
CMD=${1:-check}; TAG=${2:-job}; shift; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() { # prof <name> <command...>: kernel trace + stats, summarised
	local name=$1; shift
	(cd /tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -- "$@" > "$OUT/stats_$name.log" 2>&1)
	python "$ROOT/tools/summarize_rocprof.py" "$OUT/stats_$name" > "$OUT/${name}_kernel_stats.txt" 2>&1
	find "$OUT/stats_$name" -name "*kernel_trace.csv" -size +2M -delete
	head -14 "$OUT/${name}_kernel_stats.txt"
}
cd "$ROOT"
case $CMD in
check)
	timeout 600 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"; tail -14 "$OUT/pytest_gpu.log"
	timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc $?"; tail -2 "$OUT/smoke.log"
	timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags_line.json" 2> "$OUT/bench.err"; echo "bench rc $?"; tail -c 1500 "$OUT/bench_driver_flags_line.json" | head -c 1500; echo
	;;
tests)
	timeout ${TESTS_TIMEOUT:-500} python -m pytest -m gpu -x -q --durations=8 "$@" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest_gpu.log"; tail -25 "$OUT/pytest_gpu.log"
	;;
bench)
	if [ $# -eq 0 ]; then
		timeout 900 python bench.py > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"; echo "bench rc $?"
		timeout 600 python bench.py --pipelines 1 --no-extras --no-cpu-baseline > "$OUT/bench_pipelines1_line.json" 2> "$OUT/bench_p1.err"; echo "bench p1 rc $?"
		python - <<PY
import json
for f in ("bench_default_line.json", "bench_pipelines1_line.json"):
    d = json.load(open("$OUT/" + f)); print(f, {k: d[k] for k in ("value", "ms_per_step", "latency_ms_single")}, d["roofline"]["frac"], d["detail"].get("tiles", {}).get("erosion_1000"), d["detail"].get("dense_erosion"))
PY
	else
		timeout 900 python bench.py "$@" > "$OUT/bench_line.json" 2> "$OUT/bench.err"; echo "bench rc $?"; head -c 1200 "$OUT/bench_line.json"; echo
	fi
	;;
profile)
	WHAT=${*:-bench p1 tiles tile_erosion weights ao voxels noise erosion}
	for w in $WHAT; do
		case $w in
		bench) prof bench_default python "$ROOT/bench.py" --no-cpu-baseline --no-extras --no-rccl-world1 ;;
		p1) prof bench_pipelines1 python "$ROOT/bench.py" --no-cpu-baseline --no-extras --no-rccl-world1 --pipelines 1 ;;
		tiles) prof tiles python "$ROOT/tools/prof_tiles.py" ;;
		tile_erosion) prof tile_erosion python "$ROOT/tools/prof_tile_erosion.py" 1000 2 ;;
		weights) prof weights python "$ROOT/tools/prof_weights.py" 3 ;;
		ao) prof ao python "$ROOT/tools/prof_ao.py" ;;
		voxels) prof voxels python "$ROOT/tools/prof_voxels.py" ;;
		noise) prof noise16384 python "$ROOT/tools/prof_noise.py" 16384 2 1,2,4 ;;
		erosion) prof erosion_dense python "$ROOT/tools/prof_erosion.py" 4096 1000000 ;;
		shadows) prof shadows python "$ROOT/tools/prof_shadows.py" 3 ;;
		fused) TERRA_GEN_FUSED=1 prof noise_fused python "$ROOT/tools/prof_noise.py" 16384 6 0,1,2 8; TERRA_GEN_FUSED=1 prof voxels_fused python "$ROOT/tools/prof_voxels.py" 512 512; TERRA_GEN_FUSED=1 prof voxels64_fused python "$ROOT/tools/prof_voxels.py" 512 64 ;;
		fast) TERRA_GEN_FUSED=2 prof noise_fast python "$ROOT/tools/prof_noise.py" 16384 6 0 8; TERRA_GEN_FUSED=2 prof voxels_fast python "$ROOT/tools/prof_voxels.py" 512 512; TERRA_GEN_FUSED=2 prof voxels64_fast python "$ROOT/tools/prof_voxels.py" 512 64 ;;
		voxels64) prof voxels64 python "$ROOT/tools/prof_voxels.py" 512 64 0,1,2 ;;
		esac
	done
	;;
pmc)
	DRV=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do DRV+=("$1"); shift; done; shift || true
	for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python "$ROOT/tools/${DRV[0]}" "${DRV[@]:1}" > "$OUT/pmc_$c.log" 2>&1); done
	(cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python "$ROOT/tools/${DRV[0]}" "${DRV[@]:1}" > "$OUT/pmc_sq.log" 2>&1)
	(cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH --output-format csv -d "$OUT/pmc_sq2" -- python "$ROOT/tools/${DRV[0]}" "${DRV[@]:1}" > "$OUT/pmc_sq2.log" 2>&1)
	for k in "$@"; do for p in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_sq pmc_sq2; do echo "== $p $k"; python tools/pmc_summary.py "$OUT/$p" "$k" 2>&1 | tail -3; done; done > "$OUT/pmc_summary.txt"
	find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
	cat "$OUT/pmc_summary.txt"
	;;
erosion)
	for cfg in "$@"; do set -- $cfg; echo "== $1 x $1, $2 droplets, rings $3"; timeout 300 python tools/ero_sweep.py "$1" "$2" "$3" 2>&1 | grep -v "^    " ; done > "$OUT/erosion_timings.txt" 2>&1
	cut -c1-150 "$OUT/erosion_timings.txt"
	;;
multirank)
	TERRA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --size 4096 --no-cpu-baseline > "$OUT/bench_2rank_gloo.json" 2> "$OUT/bench_2rank_gloo.err"; echo "bench2 rc $?"
	for w in strips tiles; do TERRA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --workload $w --no-cpu-baseline --no-extras > "$OUT/bench_2rank_$w.json" 2> "$OUT/bench_2rank_$w.err"; echo "$w rc $?"; done
	python - <<PY
import json
for f in ("gloo", "strips", "tiles"):
    d = json.load(open("$OUT/bench_2rank_%s.json" % f)); print(f, d["metric"], d["value"], d["scaling"], d["n_gpus"])
PY
	;;
native)
	mkdir -p tools/_bin
	gcc -O2 -std=c99 -Iinclude tools/bench_native.c -L3dworld_amd -lterra_hip -lpthread -Wl,-rpath,"$ROOT/3dworld_amd" -o tools/_bin/bench_native && timeout 300 tools/_bin/bench_native 64 4 | tee "$OUT/bench_native.json"
	gcc -O2 -std=c99 -Iinclude tools/bench_native_multi.c -L3dworld_amd -lterra_hip -lpthread -Wl,-rpath,"$ROOT/3dworld_amd" -o tools/_bin/bench_native_multi || exit 1
	if [ $# -eq 0 ]; then set -- 2 16 16384 --same-device; fi
	timeout 600 tools/_bin/bench_native_multi "$@" | tee "$OUT/bench_native_multi.jsonl"
	# one heightmap per step on all ranks, one C process per GPU (terra.h + rccl.h): one rank through RCCL, what one rank of eight does per step, two ranks sharing this GPU (checked)
	gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/bench_native_onegrid.c -L3dworld_amd -lterra_hip -L/opt/rocm/lib -lrccl -lamdhip64 -lpthread -lm -Wl,-rpath,"$ROOT/3dworld_amd" -Wl,-rpath,/opt/rocm/lib -o tools/_bin/bench_native_onegrid || exit 1
	B=tools/_bin/bench_native_onegrid
	(timeout 120 $B 1 32 16384 1000 --warmup 8; timeout 120 $B 1 64 16384 1000 --warmup 8 --simulate-world 8; timeout 120 $B 1 64 16384 1000 --warmup 8 --simulate-world 8 --eroders 1; timeout 200 $B 2 16 4096 1000 --same-device --warmup 4 --check; timeout 120 $B 1 32 16384 1000 --warmup 8 --shard-traces --check; timeout 200 $B 2 16 4096 300 --same-device --warmup 4 --shard-traces --check) 2>&1 | grep "^{" | tee "$OUT/bench_native_onegrid.jsonl"
	;;
stepcost)
	mkdir -p tools/_bin
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I3dworld_amd/csrc tools/step_cost.hip -o tools/_bin/step_cost 2> "$OUT/step_cost_build.log" && tools/_bin/step_cost | tee "$OUT/step_cost.txt"
	timeout 300 python tools/prof_tile_erosion.py 1000 3 | tee "$OUT/tile_erosion.txt"
	;;
timeline)
	if [ $# -eq 0 ]; then set -- --steps 20 --warmup 5; fi
	(cd /tmp && timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --no-rccl-world1 --headline-only "$@" > "$OUT/trace_bench_line.json" 2> "$OUT/trace.err")
	python tools/timeline.py "$OUT/trace" --timed-steps "${TL_STEPS:-20}" > "$OUT/timeline.txt" 2>&1; cat "$OUT/timeline.txt"; cat "$OUT/trace_bench_line.json"
	python tools/summarize_rocprof.py "$OUT/trace" > "$OUT/trace_kernel_stats.txt" 2>&1
	find "$OUT/trace" -name "*kernel_trace.csv" -size +2M -delete
	;;
clock)
	(cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_clock" -- python "$ROOT/tools/$1" "${@:2}" > "$OUT/pmc_clock.log" 2>&1)
	python - <<PY
import csv, glob, collections
d = "$OUT/pmc_clock"
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:48])
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur and dur[r["Dispatch_Id"]][0] > 50000:
            a = agg[dur[r["Dispatch_Id"]][1]]; a[0] += dur[r["Dispatch_Id"]][0]; a[1] += float(r["Counter_Value"]); a[2] += 1
for k, (ns, cyc, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:50s} {n:4d} dispatches  avg {ns / n / 1e3:9.1f} us  GRBM_GUI_ACTIVE / duration = {cyc / ns:6.3f} GHz (per-XCD counter summed? divide by 8 if ~16-20)")
PY
	;;
ab)
	# gpu_job.sh ab <tag> <reps> "<name>|<ENV=.. ENV=..>|<bench.py args>" ...   the headline (--headline-only) under each variant, alternating, <reps> times: same-box A/B
	REPS=$1; shift
	: > "$OUT/ab.txt"
	for rep in $(seq 1 "$REPS"); do
		for v in "$@"; do
			IFS='|' read -r NAME ENVS ARGS <<< "$v"
			line=$(env $ENVS timeout 60 python bench.py --headline-only --no-cpu-baseline --no-rccl-world1 $ARGS 2>> "$OUT/ab.err" | tail -1)
			echo "$NAME | $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['value'], 'Gcells/s', d['ms_per_step'], 'ms/step')" "$line" 2>/dev/null || echo "FAILED: $line")" | tee -a "$OUT/ab.txt"
		done
	done
	;;
*) echo "unknown command $CMD"; exit 2 ;;
esac
