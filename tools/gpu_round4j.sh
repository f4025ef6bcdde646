#!/bin/bash
# tools/gpu_round4j.sh: SQ counters of the dense erosion's kernels summed over all launches (TERRA_GRAPHS=0: rocprofv3's counter collection crashes inside hipGraphLaunch on this
# many replays -- the same kernels, launched one by one)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04y; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp TERRA_GRAPHS=0
(cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python "$ROOT/tools/ero_sweep.py" 4096 200000 0:0 > "$OUT/pmc_sq.log" 2>&1); echo "rc $?"
(cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH --output-format csv -d "$OUT/pmc_sq2" -- python "$ROOT/tools/ero_sweep.py" 4096 200000 0:0 > "$OUT/pmc_sq2.log" 2>&1); echo "rc $?"
grep "^W" "$OUT/pmc_sq.log" | cut -c1-150
for p in pmc_sq pmc_sq2; do python tools/pmc_sum.py "$OUT/$p" "trace waves=k_waves<" "post + commit waves=k_waves_nolds" "thread passes=k_generic<"; done | tee "$OUT/pmc_erosion_dense_summary.txt"
find "$OUT" -name "*.csv" -size +1M -delete
