#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run: who ran when.  For the pipelined headline (4 heightmaps in flight) this answers what the per-kernel
statistics cannot: how much of the wall time the noise kernel is on the chip, how much of an erosion hides under another map's noise, where the chip idles.

  python tools/timeline.py <rocprof output dir> [--last-ms T]     (default: the last 25 ms of kernel activity = the timed region of a 20-step run)

Prints: the window, per kernel class the busy time (union of its intervals) and the time it runs ALONE, the chip's idle time (no kernel at all), and a coarse
strip chart (one character per 50 us: N noise only, E erosion only, B both, . idle, o other)."""
import csv
import glob
import os
import sys


def load(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    return rows


def klass(name):
    if "k_sine_grid" in name or "k_noise_grid" in name:
        return "noise"
    if "speculative_erosion" in name or "sparse_erosion" in name or "apply_erosion" in name or "direct_droplet" in name or "k_erosion" in name:
        return "erosion"
    if "gen_grid_dev" in name:
        return "tables"
    return "other"


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    d = sys.argv[1]
    last_ms = 25.0
    if "--last-ms" in sys.argv:
        last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1])
    rows = load(d)
    if not rows:
        print("no kernel trace under", d)
        return
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(last_ms * 1e6)
    if "--timed-steps" in sys.argv:  # a --headline-only run: the timed region starts with the K-th last noise kernel (one per step)
        k = int(sys.argv[sys.argv.index("--timed-steps") + 1])
        ns = sorted(r[0] for r in rows if klass(r[2]) == "noise" and r[1] - r[0] > 100_000)
        if len(ns) >= k:
            t0 = ns[-k] - 20_000
    rows = [r for r in rows if r[1] > t0]
    t0 = max(t0, min(r[0] for r in rows))
    span = t_end - t0
    by = {}
    for a, b, n, q in rows:
        by.setdefault(klass(n), []).append((max(a, t0), b))
    u = {k: union(v) for k, v in by.items()}
    allu = union([iv for v in by.values() for iv in v])
    print(f"window {span / 1e6:.3f} ms, {len(rows)} kernels, queues {sorted(set(r[3] for r in rows))}")
    print(f"chip busy (any kernel) {total(allu) / 1e6:.3f} ms = {100 * total(allu) / span:.1f} %, idle {(span - total(allu)) / 1e6:.3f} ms")
    for k in ("noise", "erosion", "tables", "other"):
        if k not in u:
            continue
        others = union([iv for kk, v in by.items() if kk != k for iv in v])
        alone = total(u[k]) - total(intersect(u[k], others))
        n = len(by[k])
        print(f"  {k:8s} {n:5d} launches, busy {total(u[k]) / 1e6:8.3f} ms ({100 * total(u[k]) / span:5.1f} %), alone {alone / 1e6:8.3f} ms, sum of durations {sum(b - a for a, b in by[k]) / 1e6:8.3f} ms")
    if "noise" in by:
        ds = sorted(b - a for a, b in by["noise"])
        print(f"  noise kernel durations: min {ds[0] / 1e3:.1f} median {ds[len(ds) // 2] / 1e3:.1f} max {ds[-1] / 1e3:.1f} us; concurrent noise kernels (time with >= 2): "
              f"{sum(max(0, min(b1, b2) - max(a1, a2)) for i, (a1, b1) in enumerate(by['noise']) for (a2, b2) in by['noise'][i + 1:]) / 1e6:.3f} ms")
    cell = 50_000
    chart = []
    for c in range(int(span // cell) + 1):
        a, b = t0 + c * cell, t0 + (c + 1) * cell
        def on(k):
            return k in u and total(intersect(u[k], [[a, b]])) > cell // 4
        n, e = on("noise"), on("erosion")
        chart.append("B" if n and e else "N" if n else "E" if e else ("o" if total(intersect(allu, [[a, b]])) > cell // 4 else "."))
    s = "".join(chart)
    for i in range(0, len(s), 100):
        print(f"  {i * cell / 1e6:6.2f} ms  {s[i:i + 100]}")


if __name__ == "__main__":
    main()
