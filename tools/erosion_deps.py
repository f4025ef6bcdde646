#!/usr/bin/env python3
"""Dependency structure between droplets of apply_erosion in serial order (CPU, oracle with its access trace): for every droplet the latest lower droplet that
wrote a cell it reads (true read-after-write at cell level) and at 8x8-block level, how far back it lies, and the depth of the dependency DAG through those
latest writers -- the number of rounds no round-synchronous scheduler can go below.   usage: erosion_deps.py [N=4096] [droplets=200000]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
orclib.build_oracle()
o = orclib.Checker("orc")
s = o.init(orclib.make_config(mesh_gen_mode=0))
g = o.gen_grid(-N / 2, -N / 2, s.DX_VAL, s.DY_VAL, N, N, 1)
fn = o.lib.orc_apply_erosion_trace
fn.restype = C.c_uint64
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p]
cap = 400 * D + 1000000
cells = np.zeros(cap, np.uint32); off = np.zeros(D + 1, np.uint64)
t0 = time.time()
n = fn(g.ctypes.data, N, N, float(g.min()), D, cells.ctypes.data, cap, off.ctypes.data)
print(f"traced {D} droplets on {N}^2: {n} accesses ({time.time() - t0:.1f}s), {n / D:.1f} per droplet", flush=True)
assert n <= cap
NX = N + 8
off = off.astype(np.int64)
# per droplet: dep_cell[j] = latest lower droplet whose WRITE a READ (or read-modify-write) of j hits, at cell level; dep_blk[j]: same with 8x8 blocks
# (a reader's block vs blocks a lower droplet wrote).  -1 = none.
last_w_cell = np.full(NX * NX, -1, np.int64)
nb = (NX >> 3) + 1
last_w_blk = np.full(nb * nb, -1, np.int64)
dep_cell = np.full(D, -1, np.int64); dep_blk = np.full(D, -1, np.int64); writes = np.zeros(D, bool); nacc = np.zeros(D, np.int64)
for j in range(D):
    a = cells[off[j]:off[j + 1]]
    if len(a) == 0:
        continue
    c = (a >> 1).astype(np.int64); w = (a & 1).astype(bool)
    nacc[j] = len(a)
    uc = np.unique(c)
    dep_cell[j] = last_w_cell[uc].max()
    bl = np.unique((uc // NX >> 3) * nb + (uc % NX >> 3))
    dep_blk[j] = last_w_blk[bl].max()
    if w.any():
        wc = np.unique(c[w]); writes[j] = True
        last_w_cell[wc] = j
        last_w_blk[np.unique((wc // NX >> 3) * nb + (wc % NX >> 3))] = j
print(f"writers: {writes.sum()} ({100 * writes.mean():.1f}%), droplets with a lower writer under their reads: cell level {(dep_cell >= 0).sum()}, block level {(dep_blk >= 0).sum()}")
for name, dep in (("cell", dep_cell), ("8x8 block", dep_blk)):
    level = np.zeros(D, np.int64)
    for j in range(D):
        if dep[j] >= 0:
            level[j] = level[dep[j]] + 1   # only the LATEST conflicting writer: a lower bound of the true depth, exact for chains through last writers
    dist = (np.arange(D) - dep)[dep >= 0]
    print(f"{name}: DAG depth (through latest writers) {level.max()}, distance to the latest conflicting lower droplet: median {np.median(dist):.0f}, 10% {np.percentile(dist, 10):.0f}, within 2048: {(dist < 2048).mean() * 100:.1f}%, within 16384: {(dist < 16384).mean() * 100:.1f}%")

# ---- the same through ALL conflicting lower writers (a droplet waits for every one of them), and weighted with the droplets' lengths: the serial chain
# in droplet STEPS that an exact scheduler which re-runs a droplet from its spawn once its inputs are final cannot go below
steps = np.zeros(D, np.int64)
for name, shift in (("cell", 0), ("8x8 block", 3)):
    nbb = (NX >> shift) + 1
    lvl = np.zeros(nbb * nbb, np.int64); tw = np.zeros(nbb * nbb, np.int64)
    depth = 0; chain = 0
    for j in range(D):
        a = cells[off[j]:off[j + 1]]
        if len(a) == 0:
            continue
        c = (a >> 1).astype(np.int64); w = (a & 1).astype(bool)
        steps[j] = max(1, (len(a) - int(w.sum())) // 4)
        u = np.unique((c // NX >> shift) * nbb + (c % NX >> shift))
        lj = lvl[u].max() + 1; tj = tw[u].max() + steps[j]
        if w.any():
            wu = np.unique((c[w] // NX >> shift) * nbb + (c[w] % NX >> shift))
            lvl[wu] = np.maximum(lvl[wu], lj); tw[wu] = np.maximum(tw[wu], tj)   # a later reader waits for every writer so far, not only the last
        depth = max(depth, lj); chain = max(chain, tj)
    print(f"{name}: depth through all lower writers {depth}; longest dependency chain weighted with droplet lengths: {chain} steps (total steps {steps.sum()}, longest droplet {steps.max()})")
