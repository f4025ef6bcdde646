#!/bin/bash
# tools/gpu_round4l.sh: the plain-C driver (tools/bench_native.c, noise turns) against bench.py on the same box: does the Python driver cost anything?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04l; mkdir -p "$OUT" tools/_bin; cd "$ROOT"; export TMPDIR=/tmp
gcc -O2 -std=c99 -Iinclude tools/bench_native.c -L3dworld_amd -lterra_hip -lpthread -Wl,-rpath,"$ROOT/3dworld_amd" -o tools/_bin/bench_native || exit 1
for rep in 1 2 3; do
	timeout 60 tools/_bin/bench_native 20 4
	timeout 60 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --no-rccl-world1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('python K20', d['value'], d['ms_per_step'])"
	timeout 60 tools/_bin/bench_native 64 4
	timeout 60 python bench.py --steps 64 --warmup 8 --headline-only --no-cpu-baseline --no-rccl-world1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('python K64', d['value'], d['ms_per_step'])"
done 2>&1 | tee "$OUT/native_vs_python.txt"
