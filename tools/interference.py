#!/usr/bin/env python3
"""What slows the noise kernel of one heightmap when another heightmap's erosion runs beside it?  (rocprofv3 timeline of the pipelined headline: a k_sine_grid launch
takes ~1.03 ms beside erosions, 0.84 ms alone.)  Context A times back-to-back noise kernels (HIP events on its stream) while context B, on its own stream and host
thread, runs one of:
   idle       nothing
   tiny       a stream of tiny dependent kernels (terra_quantize16_dev of 64 values): kernel boundaries (cache write-back / invalidate, dispatch) without any work
   erosion    1000-droplet erosions of its own 16384^2 map, back to back (what a pipeline does between its noise kernels)
   erosion_g0 the same with TERRA_GRAPHS=0
   trace_only (with TERRA_ERO_CUS etc. from the environment)
usage: interference.py [N=16384] [reps=12]"""
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
warm = int(os.environ.get("WARM", "300"))
cfg = pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1)


def run(kind):
    if kind == "erosion_g0":
        os.environ["TERRA_GRAPHS"] = "0"
    a, b = pkg.Terra(0), pkg.Terra(0)
    os.environ.pop("TERRA_GRAPHS", None)
    st = a.init_scene(cfg); b.init_scene(cfg)
    za, zb = a.alloc(N * N * 4), b.alloc(N * N * 4)
    pix = b.alloc(4096)
    mn, _ = b.gen_grid_minmax_dev(zb.ptr, -N / 2 + N, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    b.apply_erosion_dev(zb.ptr, N, N, mn, 1000, pkg.ERODE_MINZ_IS_MIN)  # warm: scratch, graph capture
    b.synchronize()
    a.gen_grid_dev(za.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE); a.synchronize()
    stop = threading.Event()
    count = [0]

    def side():
        while not stop.is_set():
            if kind == "tiny":
                for _ in range(64):
                    b.quantize16_dev(zb.ptr, 64, 0.0, 1.0, pix.ptr)
                b.synchronize()
            elif kind.startswith("erosion"):
                b.apply_erosion_dev(zb.ptr, N, N, mn, 1000, pkg.ERODE_MINZ_IS_MIN)
            else:
                time.sleep(0.001)
            count[0] += 1

    for _ in range(warm):  # the chip's clocks settle over hundreds of milliseconds of load: warm up before anything is timed
        a.gen_grid_dev(za.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    a.synchronize()
    th = threading.Thread(target=side)
    th.start()
    time.sleep(0.02)
    out = []
    for _batch in range(4):
        c0 = count[0]
        a.timer_start()
        for _ in range(reps):
            a.gen_grid_dev(za.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        out.append((a.timer_stop() / reps, count[0] - c0))
    stop.set(); th.join()
    b.synchronize()
    print(f"{kind:12s} noise kernel (+ tables), 4 batches of {reps}: " + "  ".join(f"{ms:.4f} ms ({c} side iterations)" for ms, c in out), flush=True)
    for x in (za, zb, pix):
        x.free()
    a.close(); b.close()


for kind in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["idle", "tiny", "erosion", "erosion_g0", "idle"]):
    run(kind)
