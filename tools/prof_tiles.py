#!/usr/bin/env python3
"""Tile-batch driver for rocprofv3: 64x64 tiles of 128^2 (create_zvals + stats + normals), then AO lighting.  usage: prof_tiles.py [mode=0] [reps=3]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("3dworld_amd")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = pkg.Terra(0)
t.init_scene(pkg.make_config(mesh_gen_mode=mode))
tiles = np.array([(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)], np.int32)
n = len(tiles)
zt = t.alloc(n * 130 * 130 * 4); stt = t.alloc(n * 160); nm = t.alloc(n * 129 * 129 * 4); mz = t.alloc(n * 4); ao = t.alloc(n * 129 * 129); sm = t.alloc(n * 130 * 130)
for _ in range(reps):
    t.synchronize(); t0 = time.perf_counter()
    t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr); t.synchronize()
    t1 = time.perf_counter()
    t.tiles_ao_lighting_dev(tiles, zt.ptr, ao.ptr); t.synchronize()
    t2 = time.perf_counter()
    t.tiles_mesh_shadows_dev(tiles, zt.ptr, (0.6, 0.5, 0.4), sm.ptr); t.synchronize()
    t3 = time.perf_counter()
    print(f"create_zvals {1e3*(t1-t0):.2f} ms, ao {1e3*(t2-t1):.2f} ms, shadows {1e3*(t3-t2):.2f} ms")
