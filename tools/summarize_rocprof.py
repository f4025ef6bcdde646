#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace output (csv kernel trace or rocpd .db) into a per-kernel table (calls, total, avg, min, max)."""
import csv
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"terra::terra_engine<hip_backend_t>::", "", name)
    name = re.sub(r"terra::simple_paths<hip_backend_t>::", "simple::", name)
    m = re.search(r"k_(generic|waves)<(\w+)\(.*?\)::\{lambda\(.*?\)#(\d+)\}", name)
    if m:
        return f"k_{m.group(1)}<{m.group(2)} lambda#{m.group(3)}>"
    return re.sub(r"\(.*", "", name)[:90]


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(r[0], r[1] - r[2]) for r in cur.execute("select name, end, start from kernels")]


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += rows_from_csv(p)
    if not rows:
        for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            rows += rows_from_db(p)
    agg = {}
    for n, dt in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dt; a[2] = min(a[2], dt); a[3] = max(a[3], dt)
    tot = sum(a[1] for a in agg.values()) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>11s} {'pct':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:70s} {a[0]:6d} {a[1] / 1e3:11.1f} {100 * a[1] / tot:6.2f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:10.2f} {a[3] / 1e3:10.2f}")
    print(f"{'TOTAL':70s} {sum(a[0] for a in agg.values()):6d} {tot / 1e3:11.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
