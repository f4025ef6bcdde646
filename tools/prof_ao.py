#!/usr/bin/env python3
"""HIP-event timing of the tile rows for a 64x64 batch: create_zvals (+stats + normals), AO lighting, weights texture, mesh shadows"""
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
def timed(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    t.synchronize(); t.timer_start()
    for _ in range(reps): fn()
    return t.timer_stop() / reps
print("create_zvals ms", round(timed(lambda: t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr)), 3))
print("ao ms", round(timed(lambda: t.tiles_ao_lighting_dev(tiles, zt.ptr, ao.ptr)), 3))
