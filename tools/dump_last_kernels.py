#!/usr/bin/env python3
"""the last T ms of a rocprofv3 kernel trace as a list: start (us from the window's start), duration, queue, kernel.  usage: dump_last_kernels.py <dir> [T=3]"""
import csv, glob, os, re, sys
d = sys.argv[1]; T = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
end = max(r[1] for r in rows); t0 = end - int(T * 1e6)
for s, e, q, n in rows:
    if e < t0:
        continue
    m = re.search(r"k_(generic|waves\w*)<.*?(\w+)\(.*?\)::\{lambda.*?#(\d+)\}", n)
    nm = f"k_{m.group(1)}<{m.group(2)} #{m.group(3)}>" if m else re.sub(r"\(.*", "", n)[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q:>3s} {nm}")
