#!/usr/bin/env python3
"""Erosion-only driver for rocprofv3 (torch-free): one N^2 sine heightmap, then apply_erosion with D droplets.
usage: prof_erosion.py [N=4096] [D=100000] [reps=1] [ring slots] [slice steps]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
t = pkg.Terra(0)
if len(sys.argv) > 4 and int(sys.argv[4]): t.set_erosion_tuning(window=int(sys.argv[4]))
if len(sys.argv) > 5 and int(sys.argv[5]): t.set_erosion_slice_steps(int(sys.argv[5]))
st = t.init_scene(pkg.make_config(mesh_gen_mode=0))
z = t.alloc(N * N * 4)
for _ in range(reps):
    mn, mx = t.gen_grid_minmax_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    t.synchronize(); t0 = time.perf_counter()
    t.apply_erosion_dev(z.ptr, N, N, mn, D, pkg.ERODE_MINZ_IS_MIN); t.synchronize()
    dt = time.perf_counter() - t0
    r = t.erosion_report().as_dict()
    print(json.dumps({"N": N, "droplets": D, "args": sys.argv[4:], "ms": round(dt * 1e3, 2), **r}))
