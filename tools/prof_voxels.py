#!/usr/bin/env python3
"""Voxel sine field (BASELINE config 5) a few times, for rocprofv3 --kernel-trace --stats."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
t = pkg.Terra(0)
t.init_scene(pkg.make_config(mesh_gen_mode=0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nz = int(sys.argv[2]) if len(sys.argv) > 2 else n  # prof_voxels.py 512 64: the reference's own field (config_voxel_params.txt:1-3)
modes = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
v = t.alloc(n * n * nz * 4)
lo, vsz, off = (-1.0, -1.0, -1.0), (2.0 / n, 2.0 / n, 2.0 / nz), (0.0, 0.0, 0.0)
for mode in modes:
    for _ in range(6):
        t.voxel_fill_dev(v.ptr, n, n, nz, lo, vsz, off, 1.0, 1.0, 123, 456, mode, 0.01, 1)
    t.synchronize()
    t.timer_start()
    for _ in range(10):
        t.voxel_fill_dev(v.ptr, n, n, nz, lo, vsz, off, 1.0, 1.0, 123, 456, mode, 0.01, 1)
    ms = t.timer_stop() / 10
    print(f"voxels {n}x{n}x{nz} mode {mode} fused {os.environ.get('TERRA_GEN_FUSED', '0')}: {ms:.4f} ms/call  {n * n * nz / ms / 1e6:.1f} Gvoxels/s", flush=True)
