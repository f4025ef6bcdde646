#!/usr/bin/env python3
"""torch-free driver of the bench workload for rocprofv3 PMC passes (same kernels, same sizes as bench.py)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = pkg.Terra(0)
st = t.init_scene(pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
z = t.alloc(N * N * 4)
for _ in range(reps):
    mn, mx = t.gen_grid_minmax_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    t.apply_erosion_dev(z.ptr, N, N, mn, 1000, pkg.ERODE_MINZ_IS_MIN)
# calibration points for the TCC counters in the same trace: a pure float4 read of the grid (k_minmax) and a 4 B/lane read + 2x1 B/lane write (quantise)
pix = t.alloc(N * N * 2)
mn2, mx2 = t.minmax_dev(z.ptr, N * N)
t.quantize16_dev(z.ptr, N * N, mn2, max(mx2 - mn2, 1e-12), pix.ptr)
t.synchronize()
print("done", mn, mx, t.erosion_report().as_dict())
