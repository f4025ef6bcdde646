"""Dataflow bounds of the droplet dependency structure on the serial-order access trace of the oracle (CPU only): the longest dependency chain in droplet STEPS when a
droplet may start only after every lower writer of what it touches has finished (whole-droplet) and when each step waits only for the writes it reads (step-level), at cell
and at 8x8-block granularity.  4096^2 / 10^6 droplets: 284-291 K steps whole-droplet, 51-65 K step-level.   usage: ero_dataflow.py N droplets"""
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
t0 = time.time()
o.set_num_threads(1)
gg = g.copy()
n = fn(gg.ctypes.data, N, N, float(g.min()), D, cells.ctypes.data, cap, off.ctypes.data)
print(f"traced: {n} accesses {time.time()-t0:.1f}s"); assert n <= cap
st, steps = o.apply_erosion_stats(g.copy(), float(g.min()), D)
steps = np.maximum(steps, 1).astype(np.float64)
print("total steps", steps.sum())
off = off.astype(np.int64)
import subprocess
_here = os.path.dirname(os.path.abspath(__file__))
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(_here, "_libdataflow.so"), os.path.join(_here, "ero_dataflow.c")], check=True)
L = C.CDLL(""+os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libdataflow.so")+"")
L.dataflow.restype = C.c_double
L.dataflow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]
for W in (0, 4096, 16384):
    for shift in (0, 3):
        for whole in (1, 0):
            r = L.dataflow(cells.ctypes.data, off.ctypes.data, steps.ctypes.data, D, N + 8, shift, whole, W, None)
            print(f"W {W} {'block' if shift else 'cell'} {'whole-droplet' if whole else 'step-level'}: chain {r:.0f} steps")
