"""Round-synchronous scheduler model on the oracle's access trace (CPU only): rounds and the sum over rounds of the longest wave (steps) for the speculative
multi-version scheduler, with versions that appear when their trace has finished vs first traces that are visible while they grow, for slice / near / checkpoint settings.
model2 reproduces the measured run (4096^2, 10^6 droplets: 1488 rounds / 238 905 steps against 1500 / 238 600 on the GPU) and predicted the gain of the live
versions (1340 rounds / 183 781 steps; measured 1351 rounds).   usage: ero_sched_model.py N droplets [W]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, ""+os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")+"")
import orclib
N = int(sys.argv[1]); D = int(sys.argv[2])
o = orclib.Checker("orc")
s = o.init(orclib.make_config(mesh_gen_mode=0))
g = o.gen_grid(-N / 2, -N / 2, s.DX_VAL, s.DY_VAL, N, N, 1)
fn = o.lib.orc_apply_erosion_trace
fn.restype = C.c_uint64
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p]
cap = 200 * D + 1000000
cells = np.zeros(cap, np.uint32); off = np.zeros(D + 1, np.uint64)
o.set_num_threads(1)
gg = g.copy()
n = fn(gg.ctypes.data, N, N, float(g.min()), D, cells.ctypes.data, cap, off.ctypes.data); assert n <= cap
off = off.astype(np.int64)
import subprocess
_here = os.path.dirname(os.path.abspath(__file__))
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(_here, "_libsched.so"), os.path.join(_here, "ero_sched_model.c")], check=True)
L = C.CDLL(""+os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libsched.so")+"")
L.build.restype = C.c_int64
L.build.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]
L.simulate.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
L.simulate2.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
t0 = time.time()
print("steps (groups):", L.build(cells.ctypes.data, off.ctypes.data, D, N + 8, 3), f"{time.time()-t0:.1f}s")
out = np.zeros(3)
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
for (pip, sl, near, ck) in [(0, 128, 512, 32), (1, 128, 512, 32), (1, 128, 0, 32), (1, 64, 512, 32), (1, 64, 0, 32), (1, 64, 0, 16), (1, 32, 0, 16), (1, 32, 0, 8), (0, 64, 512, 32), (0, 1<<30, 0, 32)]:
    t0 = time.time()
    L.simulate2(W, pip, sl, near, ck, out.ctypes.data)
    print(f"model2 W {W} {'live partial' if pip else 'on completion'} slice {sl if sl < 1 << 30 else 'inf'} near {near} ck {ck}: rounds {out[0]:.0f}  sum of longest waves {out[1]:.0f} steps  traced {out[2]:.0f}  ({time.time()-t0:.1f}s)", flush=True)
