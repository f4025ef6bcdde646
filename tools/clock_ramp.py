#!/usr/bin/env python3
"""How long does the chip take to reach its sustained clock under the noise kernel, and how long an idle gap makes it fall back?
   clock_ramp.py [N=16384]
 1. from a cold process: 40 batches of 4 launches (k_sine_grid + its table kernels), ms per launch of each batch -> the ramp
 2. after 300 warm launches: synchronize, sleep for g in (0, 0.2, 0.5, 1, 2, 5, 20, 100) ms, then 6 batches of 4 -> what a gap costs"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
a = pkg.Terra(0)
st = a.init_scene(pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
z = a.alloc(N * N * 4)
def batch(k=4):
    a.timer_start()
    for _ in range(k):
        a.gen_grid_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    return a.timer_stop() / k
a.gen_grid_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE); a.synchronize()  # first-use allocations
time.sleep(0.5)
print("cold ramp, ms per launch in batches of 4:", " ".join(f"{batch():.3f}" for _ in range(40)), flush=True)
for g in (0.0, 0.0002, 0.0005, 0.001, 0.002, 0.005, 0.02, 0.1):
    for _ in range(75):
        batch()
    a.synchronize()
    if g:
        time.sleep(g)
    print(f"gap {g * 1e3:6.1f} ms:", " ".join(f"{batch():.3f}" for _ in range(6)), flush=True)
