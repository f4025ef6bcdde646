#!/usr/bin/env python3
"""64 x 64 tile batch (zvals + stats + normals) in the fBm modes: ms per batch.  usage: prof_tiles_fbm.py [modes=1,2,4]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
t = pkg.Terra(0)
tiles = np.array([(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)], np.int32)
n = len(tiles)
zt = t.alloc(n * 130 * 130 * 4); stt = t.alloc(n * 160); nm = t.alloc(n * 129 * 129 * 4); mz = t.alloc(n * 4)
for mode in [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4").split(",")]:
    t.init_scene(pkg.make_config(mesh_gen_mode=mode))
    for _ in range(3):
        t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr)
    t.synchronize(); t.timer_start()
    for _ in range(5):
        t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr)
    ms = t.timer_stop() / 5
    print(f"tiles 64x64 mode {mode}: {ms:.3f} ms per batch  {n * 130 * 130 / ms / 1e6:.1f} Gcells/s", flush=True)
