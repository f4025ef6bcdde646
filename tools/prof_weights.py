#!/usr/bin/env python3
"""Weights-texture driver for rocprofv3: 64x64 tiles, tiles_create_weights_dev.  usage: prof_weights.py [reps=3]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("3dworld_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t = pkg.Terra(0)
t.init_scene(pkg.make_config(mesh_gen_mode=0))
t.set_landscape(pkg.make_landscape(grass_density=100))
tiles = np.array([(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)], np.int32)
n = len(tiles)
zt = t.alloc(n * 130 * 130 * 4); wt = t.alloc(n * 129 * 129 * 4); gb = t.alloc(n * 32 * 32 * 12); hg = t.alloc(n)
t.tiles_create_zvals_dev(tiles, 0, zt.ptr)
for _ in range(reps):
    t.synchronize(); t0 = time.perf_counter()
    t.tiles_create_weights_dev(tiles, zt.ptr, wt.ptr, gb.ptr, hg.ptr); t.synchronize()
    print(f"weights {1e3*(time.perf_counter()-t0):.2f} ms")
