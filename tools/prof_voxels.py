#!/usr/bin/env python3
"""Voxel sine field (BASELINE config 5) a few times, for rocprofv3 --kernel-trace --stats."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
t = pkg.Terra(0)
t.init_scene(pkg.make_config(mesh_gen_mode=0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
v = t.alloc(n * n * n * 4)
lo, vsz, off = (-1.0, -1.0, -1.0), (2.0 / n, 2.0 / n, 2.0 / n), (0.0, 0.0, 0.0)
for mode in (0,):
    for _ in range(6):
        t.voxel_fill_dev(v.ptr, n, n, n, lo, vsz, off, 1.0, 1.0, 123, 456, mode, 0.01, 1)
    t.synchronize()
    t.timer_start()
    for _ in range(10):
        t.voxel_fill_dev(v.ptr, n, n, n, lo, vsz, off, 1.0, 1.0, 123, 456, mode, 0.01, 1)
    print("mode", mode, "ms/call", t.timer_stop() / 10)
