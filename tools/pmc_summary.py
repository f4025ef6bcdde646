#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel-name-substring, per dispatch, counters + derived per-wave ratios.
usage: pmc_summary.py <dir> <kernel substring>"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
disp = collections.OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = r["Dispatch_Id"]
        disp.setdefault(k, {"name": r["Kernel_Name"][:60], "grid": r.get("Grid_Size")})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in list(disp.items())[-3:]:
    if k in dur:
        v["duration_us"] = dur[k]
    w = v.get("SQ_WAVES", 0) or 1
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(k, v["name"], "grid", v["grid"])
    print("   ", {n: f"{x:.4g}" for n, x in v.items() if isinstance(x, float)})
    print(f"    per wave: VALU {v.get('SQ_INSTS_VALU', 0)/w:.0f} SALU {v.get('SQ_INSTS_SALU', 0)/w:.0f} LDS {v.get('SQ_INSTS_LDS', 0)/w:.0f} VMEM {(v.get('SQ_INSTS_VMEM_RD', 0)+v.get('SQ_INSTS_VMEM_WR', 0))/w:.0f}"
          f" | of wave cycles: active_valu {v.get('SQ_ACTIVE_INST_VALU', 0)/wc:.2f} wait_any {v.get('SQ_WAIT_ANY', 0)/wc:.2f} wait_inst {v.get('SQ_WAIT_INST_ANY', 0)/wc:.2f} ifetch {v.get('SQ_IFETCH', 0)/w:.0f}/wave")
