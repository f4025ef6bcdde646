#!/usr/bin/env python3
"""Sum the counters of a rocprofv3 --pmc pass over ALL dispatches of a kernel (a dense erosion is a thousand launches of the same few kernels).
usage: pmc_sum.py <dir> <label>=<kernel substring> [...]"""
import csv, glob, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = [r for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
for spec in sys.argv[2:]:
    label, pat = spec.split("=", 1)
    tot, ids = collections.defaultdict(float), set()
    for r in rows:
        if pat in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
    us = sum(dur.get(i, 0.0) for i in ids)
    print(f"== {d.rstrip('/').split('/')[-1]}: {label}  ({pat}): {len(ids)} dispatches, {us / 1e3:.1f} ms of kernel time")
    print("    ", {k: f"{v:.4g}" for k, v in sorted(tot.items())})
    wc = tot.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        valu_us = tot.get("SQ_INSTS_VALU", 0.0) * 4 / 1024 / 2400.0
        print(f"    of wave cycles: active_valu {tot.get('SQ_ACTIVE_INST_VALU', 0) / wc:.2f}  wait_any {tot.get('SQ_WAIT_ANY', 0) / wc:.2f}  wait_inst {tot.get('SQ_WAIT_INST_ANY', 0) / wc:.2f};"
              f"  VALU instructions x 4 cycles / 1024 SIMDs at 2.4 GHz = {valu_us:.0f} us of the {us:.0f} us: the vector ALUs are {100 * valu_us / max(us, 1e-9):.1f} % busy;  {tot.get('SQ_INSTS_VALU', 0) / max(tot.get('SQ_WAVES', 1), 1):.0f} VALU instructions per wave")
