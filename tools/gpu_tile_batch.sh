#!/bin/bash
# batched tile erosion: parity of the tile cases, then timings for 1 / 4 / 8 droplets of a tile in flight.  Every step under its own timeout.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-tilebatch}
mkdir -p $OUT
cd $ROOT
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_tiles or tile_erosion_large or tile_golden" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
for w in 1 4 8; do echo "== waves $w"; TERRA_TILE_WAVES=$w TERRA_TILE_STATS=1 timeout 60 python tools/prof_tile_erosion.py 1000 2 2>&1 | tail -3; done
