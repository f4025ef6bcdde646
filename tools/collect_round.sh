#!/bin/bash
# tools/collect_round.sh <tag>: the evidence set of a round in one GPU call (bench lines, rocprofv3 kernel stats, PMC passes, secondary workloads, erosion timings).
# Everything goes to gpurun_out/<tag>/; the summaries are then copied to profiles/<tag>_*.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
tools/gpu_job.sh bench $TAG
tools/gpu_job.sh bench ${TAG}/driver --steps 20 --warmup 5
tools/gpu_job.sh profile $TAG bench p1 tiles tile_erosion weights ao voxels noise erosion
tools/gpu_job.sh pmc ${TAG}/pmc_sine prof_driver.py 16384 2 -- "k_sine_grid" "k_minmax" "quantize16" "sparse_erosion"
tools/gpu_job.sh pmc ${TAG}/pmc_tile_erosion prof_tile_erosion.py 1000 1 -- "k_tile_erosion" "k_sine_grid"
tools/gpu_job.sh pmc ${TAG}/pmc_noise prof_noise.py 16384 1 1,2,4 -- "k_noise_grid<1>" "k_noise_grid<2>" "k_noise_grid<4>"
tools/gpu_job.sh pmc ${TAG}/pmc_tiles prof_tiles.py 0 1 -- "k_tile_post(" "k_tile_ao" "k_tile_shadows"
tools/gpu_job.sh pmc ${TAG}/pmc_voxels prof_voxels.py 512 -- "k_voxel_sines"
TERRA_GRAPHS=0 tools/gpu_job.sh pmc ${TAG}/pmc_erosion_dense ero_sweep.py 4096 200000 0:0 -- "speculative_erosion lambda#1" "k_waves_nolds<speculative_erosion" # (graphs off: rocprofv3 collects no counters for kernels replayed from a hipGraph -- round 5's summary came back empty)
tools/gpu_job.sh profile $TAG shadows fused fast voxels64
TERRA_GEN_FUSED=1 tools/gpu_job.sh pmc ${TAG}/pmc_fused prof_noise.py 16384 2 0 8 -- "k_sine_grid_mx" "k_minmax" "quantize16"
TERRA_GEN_FUSED=2 tools/gpu_job.sh pmc ${TAG}/pmc_fast prof_noise.py 16384 2 0 8 -- "k_sine_grid_h3" "k_split_table" "k_minmax" "quantize16"
TERRA_GEN_FUSED=2 tools/gpu_job.sh pmc ${TAG}/pmc_voxels_fast prof_voxels.py 512 512 -- "k_sine_grid_h3" "k_split_voxel_table"
tools/gpu_job.sh pmc ${TAG}/pmc_voxel_noise prof_voxels.py 512 64 1,2 -- "k_voxel_noise<true>" "k_voxel_noise<false>"
tools/gpu_job.sh pmc ${TAG}/pmc_weights prof_weights.py 2 -- "k_tile_weights" "k_sine_grid<true"
tools/gpu_job.sh pmc ${TAG}/pmc_shadows prof_shadows.py 2 -- "k_tile_shadows_flow"
tools/gpu_job.sh clock ${TAG}/clock prof_driver.py 16384 3
python tools/make_pmc_traffic.py gpurun_out/${TAG}/pmc_sine 16384 > gpurun_out/${TAG}/pmc_traffic.json 2>/dev/null
tools/gpu_job.sh erosion $TAG "16384 1000 0:0" "4096 1000 0:0" "4096 100000 0:0" "4096 1000000 0:0" "8192 1000000 0:0" "16384 1000000 0:0" "1024 30000 0:0" "2048 1000000 0:0"
tools/gpu_job.sh stepcost $TAG
[ -n "${SKIP_BENCH_EXTRA:-}" ] || { timeout 600 python tools/bench_extra.py > gpurun_out/${TAG}/bench_extra.json 2> gpurun_out/${TAG}/bench_extra.err; echo "bench_extra rc $?"; }
tools/gpu_job.sh native $TAG
find gpurun_out/${TAG} -name "*.csv" -size +1M -delete
