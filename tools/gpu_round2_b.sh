#!/bin/bash
# round 2: parity suite + timings of the table-driven fBm kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02b}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $ROOT/tools/prof_noise.py 4096 5 1,2,4 > $OUT/noise_4096.txt 2>&1
timeout 300 python $ROOT/tools/prof_noise.py 16384 2 1,2,4,0 > $OUT/noise_16384.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq_noise4096 -- python $ROOT/tools/prof_noise.py 4096 2 1,2,4 > $OUT/pmc_sq_noise4096.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds_noise4096 -- python $ROOT/tools/prof_noise.py 4096 2 1,2,4 > $OUT/pmc_lds_noise4096.log 2>&1
cd $ROOT
for k in "k_noise_grid<1>" "k_noise_grid<2>" "k_noise_grid<4>"; do echo "== sq $k"; python tools/pmc_summary.py $OUT/pmc_sq_noise4096 "$k" 2>&1 | tail -3; echo "== lds $k"; python tools/pmc_summary.py $OUT/pmc_lds_noise4096 "$k" 2>&1 | tail -3 | head -2; done > $OUT/pmc_noise_summary.txt
find $OUT -name "*kernel_trace.csv" -size +3M -delete
cat $OUT/noise_4096.txt $OUT/noise_16384.txt; cat $OUT/pmc_noise_summary.txt
