#!/bin/bash
# tools/pmc_probe.sh <outdir> <binary> [args...]: SQ + clock counters of a stand-alone probe binary, one rocprofv3 pass each (run on the GPU box through gpurun)
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
BIN=$(realpath "$1"); shift
run() { (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc "${@:2}" --output-format csv -d "$OUT/$1" -- "$BIN" $ARGS > "$OUT/$1.log" 2>&1); }
ARGS="$*"
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run clk GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(f.split("/")[-3] if False else f[len(d):].split("/")[1], r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3
for p in ("sq", "clk", "sq2"):
    disp = collections.OrderedDict()
    for f in glob.glob(d + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"][:50]})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in list(disp.items())[-2:]:
        v["us"] = dur.get((p, k), 0)
        print(p, k, {n: (f"{x:.5g}" if isinstance(x, float) else x) for n, x in v.items()})
PY
