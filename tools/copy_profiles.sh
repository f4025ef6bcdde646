#!/bin/bash
# tools/copy_profiles.sh <gpurun_out tag> <round prefix>: copy the summaries of tools/collect_round.sh (+ multirank, prof_tile2) into profiles/<prefix>_*
S=gpurun_out/${1:-r03c}; R=${2:-r03}; P=profiles
cp $S/bench_default_line.json $P/${R}_bench_default_line.json; cp $S/driver/bench_line.json $P/${R}_bench_driver_flags_line.json; cp $S/bench_pipelines1_line.json $P/${R}_bench_pipelines1_line.json
for k in bench_default bench_pipelines1 tiles tile_erosion weights ao voxels noise16384 erosion_dense; do cp $S/${k}_kernel_stats.txt $P/${R}_${k}_kernel_stats.txt; done
for k in bench_default bench_pipelines1; do f=$(find $S/stats_$k -name "*kernel_stats.csv" | head -1); cp "$f" $P/${R}_${k}_rocprofv3_kernel_stats.csv; done
cp $S/pmc_sine/pmc_summary.txt $P/${R}_pmc_summary.txt; cp $S/pmc_tile_erosion/pmc_summary.txt $P/${R}_pmc_tile_erosion_summary.txt; cp $S/pmc_traffic.json $P/${R}_pmc_traffic.json
for k in noise_fused noise_fast voxels_fused voxels_fast voxels64 voxels64_fused voxels64_fast shadows; do [ -f $S/${k}_kernel_stats.txt ] && cp $S/${k}_kernel_stats.txt $P/${R}_${k}_kernel_stats.txt; done
for k in noise tiles voxels erosion_dense fused fast voxels_fast voxel_noise weights shadows; do [ -f $S/pmc_$k/pmc_summary.txt ] && cp $S/pmc_$k/pmc_summary.txt $P/${R}_pmc_${k}_summary.txt; done
cp $S/erosion_timings.txt $P/${R}_erosion_timings.txt; cat $S/step_cost.txt $S/tile_erosion.txt > $P/${R}_step_cost.txt
cp $S/bench_extra.json $P/${R}_bench_extra.json; cat $S/bench_native.json $S/bench_native_multi.jsonl > $P/${R}_bench_native.jsonl; [ -f $S/bench_native_onegrid.jsonl ] && cp $S/bench_native_onegrid.jsonl $P/${R}_onegrid_native.jsonl
for w in gloo strips tiles; do [ -f $S/bench_2rank_$w.json ] && cp $S/bench_2rank_$w.json $P/${R}_bench_2rank_$w.json; done
[ -f $S/tile_two_waves.txt ] && cp $S/tile_two_waves.txt $P/${R}_tile_two_waves.txt
ls $P | grep -c "^${R}_"
