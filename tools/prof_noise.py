#!/usr/bin/env python3
"""torch-free driver for rocprofv3 passes over the fBm grid kernels (k_noise_grid<simplex|perlin|dwarp>) and the streaming helpers:
   prof_noise.py <N> <reps> <mode,mode,...> [octaves]      modes: 1 simplex, 2 Perlin, 4 domain warp, 0 sine
prints HIP-event times per mode so the same run also gives Gcells/s."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
modes = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4").split(",")]
octaves = int(sys.argv[4]) if len(sys.argv) > 4 else 8
t = pkg.Terra(0)
z = t.alloc(N * N * 4)
pix = t.alloc(N * N * 2)
for mode in modes:
    st = t.init_scene(pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=9 - octaves))
    t.gen_grid_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    t.synchronize()
    t.timer_start()
    for _ in range(reps):
        t.gen_grid_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    ms = t.timer_stop() / reps
    print(f"mode {mode} N {N} octaves {octaves}: {ms:.4f} ms  {N * N / ms / 1e6:.2f} Gcells/s", flush=True)
mn, mx = t.minmax_dev(z.ptr, N * N)
t.timer_start()
for _ in range(reps):
    t.minmax_dev(z.ptr, N * N)
ms_mm = t.timer_stop() / reps
t.timer_start()
for _ in range(reps):
    t.quantize16_dev(z.ptr, N * N, mn, max(mx - mn, 1e-12), pix.ptr)
ms_q = t.timer_stop() / reps
print(f"minmax {ms_mm:.4f} ms = {N * N * 4 / ms_mm / 1e6:.1f} GB/s ; quantize16 {ms_q:.4f} ms = {N * N * 6 / ms_q / 1e6:.1f} GB/s", flush=True)
t.close()
