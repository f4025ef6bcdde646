#!/usr/bin/env python3
"""How often do consecutive droplets of ONE tile's erosion overlap?  (CPU, oracle with its access trace.)  For a few tiles of BASELINE config 4 (130^2 zvals, 1000 droplets
in serial order): coarse 4x4-cell footprints of every droplet, the fraction of neighbouring pairs (i, i+1) whose footprints intersect, and the serial chain that is left when
disjoint neighbours run as a pair (steps of a pair = max instead of sum) -- the gain a two-droplets-per-tile kernel could have.
usage: tile_pair_overlap.py [tx ty ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
orclib.build_oracle()
o = orclib.Checker("orc")
s = o.init(orclib.make_config(mesh_gen_mode=0))
fn = o.lib.orc_apply_erosion_trace
fn.restype = C.c_uint64
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p]
args = [int(a) for a in sys.argv[1:]] or [0, 0, 3, 5, -7, 2, 10, -12, 1, 1, 20, 20]
D, n = 1000, 130
for tx, ty in zip(args[0::2], args[1::2]):
    g = o.gen_grid(tx * 128 - 64 - 64, ty * 128 - 64 - 64, s.DX_VAL, s.DY_VAL, n, n, 1)
    cap = 4000 * D
    cells = np.zeros(cap, np.uint32); off = np.zeros(D + 1, np.uint64)
    tot = fn(g.ctypes.data, n, n, float(s.zmin), D, cells.ctypes.data, cap, off.ctypes.data)
    off = off.astype(np.int64)
    NX = n + 8
    foot = []; steps = []
    for j in range(D):
        a = cells[off[j]:off[j + 1]]
        c = (a >> 1).astype(np.int64); w = (a & 1).astype(bool)
        steps.append(max(1, (len(a) - int(w.sum())) // 4) if len(a) else 0)
        foot.append(set(((c // NX) >> 2) * 64 + ((c % NX) >> 2)) if len(a) else set())
    steps = np.array(steps)
    serial = steps.sum(); chain = 0; j = 0; pairs = 0; ok = 0
    while j < D:
        if j + 1 < D:
            pairs += 1
            if not (foot[j] & foot[j + 1]):
                ok += 1; chain += max(steps[j], steps[j + 1]); j += 2; continue
            chain += steps[j] + steps[j + 1]; j += 2; continue   # overlap: the higher droplet re-runs after the lower one
        chain += steps[j]; j += 1
    print(f"tile ({tx},{ty}): {serial} steps in serial order, land droplets {(steps > 2).sum()}, neighbouring pairs disjoint at 4x4 granularity {ok}/{pairs} = {ok / max(1, pairs):.2f}, chain with pairs {chain} = {serial / max(1, chain):.2f}x shorter")
    for lag in (1, 2, 3, 4, 5, 6, 7, 8, 16, 37, 64):
        ok = sum(1 for j in range(D - lag) if not (foot[j] & foot[j + lag]))
        print(f"    droplets (i, i+{lag}) disjoint: {ok / (D - lag):.3f}", end=";")
    print()
    for K in (2, 4, 8):
        clean = 0; nb = 0
        for b in range(0, D - K + 1, K):
            nb += 1
            clean += all(not (foot[b + i] & foot[b + j]) for i in range(K) for j in range(i + 1, K))
        print(f"    batches of {K} consecutive droplets that are pairwise disjoint: {clean}/{nb}")
    xs0 = [min(foot[j]) % 64 if foot[j] else -1 for j in range(12)]
    print("    first coarse column of droplets 0..11:", xs0)
