#!/usr/bin/env python3
"""64x64 tiles of 128^2 with `iters` droplets each (BASELINE config 4 with erosion): time of the batch.  usage: prof_tile_erosion.py [iters=1000] [reps=2]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("3dworld_amd")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t = pkg.Terra(0)
t.init_scene(pkg.make_config(mesh_gen_mode=0))
tiles = np.array([(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)], np.int32)
n = len(tiles)
zt = t.alloc(n * 130 * 130 * 4)
for _ in range(reps):
    t.synchronize(); t0 = time.perf_counter()
    t.tiles_create_zvals_dev(tiles, 0, zt.ptr); t.synchronize()
    t1 = time.perf_counter()
    t.tiles_create_zvals_dev(tiles, iters, zt.ptr); t.synchronize()
    t2 = time.perf_counter()
    print(f"zvals {1e3*(t1-t0):.2f} ms, zvals + {iters} droplets per tile {1e3*(t2-t1):.2f} ms => erosion {1e3*((t2-t1)-(t1-t0)):.2f} ms")
