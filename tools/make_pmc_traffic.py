#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_driver.py (one counter per rocprofv3 pass):
HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports 1/2 of
a wide coalesced read; checked here on k_minmax, which reads a known byte count in the same trace; WRITE_SIZE checked on k_quantize16).
usage: make_pmc_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <N> > out.json"""
import csv, glob, json, sys
d, N = sys.argv[1], int(sys.argv[2])

def per_kernel(counter):
    rows = {}
    for f in glob.glob(f"{d}/pmc_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows

def pick(rows, pat, big_only=True):
    best = None
    for k, v in rows.items():
        if pat in k:
            vals = [x for x in v if (not big_only) or x > 0.25 * max(v)]  # skip the small start-up launches of the same kernel
            best = sum(vals) / len(vals)
    return best

fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python tools/prof_driver.py %d 2   (one counter per pass; tools/collect_profiles_r02.sh)" % N,
       "unit": "KB as reported by rocprofv3; fetch bytes = 2 x FETCH_SIZE on gfx950 (guide, HBM section), checked on k_minmax's known read below; WRITE_SIZE checked on k_quantize16's known write",
       "kernels": {}}
cells = N * N
for name, pat in (("k_sine_grid", "k_sine_grid<false, false"), ("k_minmax", "k_minmax"), ("k_quantize16", "k_quantize16")):
    f, w = pick(fetch, pat), pick(write, pat)
    if f is None or w is None:
        continue
    e = {"FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1), "hbm_bytes_per_launch": int(round((2 * f + w) * 1024))}
    if name == "k_sine_grid":
        e["algorithmic_bytes_per_launch"] = 4 * cells
    if name == "k_minmax":
        e["known_read_KB"] = 4 * cells // 1024
    if name == "k_quantize16":
        e["known_read_KB"] = 4 * cells // 1024; e["known_write_KB"] = 2 * cells // 1024
    out["kernels"][name] = e
print(json.dumps(out, indent=1))
