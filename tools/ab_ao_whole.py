#!/usr/bin/env python3
"""tools/ab_ao_whole.py: the AO row of a 64 x 64 tile batch with the rays from one workgroup per tile (k_tile_ao_tile, "ao.whole" 1) and from four band workgroups per tile
(k_tile_ao, "ao.whole" 0), alternating on the same box; the bytes of both forms are compared"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("3dworld_amd")
t = pkg.Terra(0)
t.init_scene(pkg.make_config(mesh_gen_mode=0))
tiles = np.array([(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)], np.int32)
n = len(tiles)
zt = t.alloc(n * 130 * 130 * 4); stt = t.alloc(n * 160); nm = t.alloc(n * 129 * 129 * 4); mz = t.alloc(n * 4); ao = t.alloc(n * 129 * 129)
def timed(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    t.synchronize(); t.timer_start()
    for _ in range(reps): fn()
    return t.timer_stop() / reps
t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr)
res = {}
for rep in range(3):
    for whole in ("0", "1"):
        t.set_option("ao.whole", whole)
        ms = timed(lambda: t.tiles_ao_lighting_dev(tiles, zt.ptr, ao.ptr))
        res.setdefault(whole, []).append(round(ms, 3))
        if rep == 0: res["bytes" + whole] = ao.download(np.uint8, (n, 129, 129)).copy()
print("ao.whole 0 ms", res["0"]); print("ao.whole 1 ms", res["1"]); print("equal", bool((res["bytes0"] == res["bytes1"]).all()))
