#!/usr/bin/env python3
"""What one rank of the one-grid pipeline does per step at a simulated world size, with the sparse erosion's traces sharded by strip owner (terra_erosion_shard_*), torch-free
(no collectives: their place in the stream order is kept by the events): strip noise on the noise context; this rank's traces on a tracer context behind it; every
world-th step the eroder's gather + check + commit.  For rocprofv3 --kernel-trace (tools/timeline.py / the dump below).  usage: prof_shard_floor.py [world=8] [steps=64] [tracers=1] [shard=1]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 1
SHARD = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0
N, D = 16384, 1000
cfg = pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1)
t = pkg.Terra(0); st = t.init_scene(cfg)
e = pkg.Terra(0); e.init_scene(cfg)
tcs = [pkg.Terra(0) for _ in range(NT)]
for c in tcs:
    c.init_scene(cfg)
rows = -(-N // W)
z = t.alloc(N * N * 4); ez = t.alloc(N * N * 4); mm = t.alloc(8)
full_min, _ = e.gen_grid_minmax_dev(ez.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
import numpy as np
mmz = t.alloc(8).upload(np.array([full_min, 0.0], np.float32))
stride = -(-e.erosion_shard_arena_bytes(D) // 4096) * 4096
arena = t.alloc(stride * W)
row_end = [min((r + 1) * rows, N) for r in range(W)]
for r in range(W):
    r0 = min(r * rows, N)
    tcs[0].erosion_shard_trace_dev(ez.ptr, N, N, D, r0, row_end[r] - r0, arena.ptr + r * stride)
tcs[0].synchronize()
ev, ev2 = t.event_create(), t.event_create()


def steps(k):
    th = None
    for s in range(k):
        t.gen_grid_rows_minmax_async_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 0, rows, mm.ptr, pkg.GEN_GLACIATE)
        t.event_record(ev)
        if SHARD:
            tc = tcs[s % NT]
            tc.event_wait(ev)
            tc.erosion_shard_trace_dev(ez.ptr, N, N, D, 0, rows, arena.ptr)
            tc.event_record(ev2)
        if s % W == 0:
            if th is not None:
                th.join()

            def job():
                if SHARD:
                    e.event_wait(ev2)
                    e.erosion_shard_finish_dev(ez.ptr, N, N, mmz.ptr, D, pkg.ERODE_MINZ_IS_MIN, W, 0, row_end, arena.ptr, stride)
                else:
                    e.event_wait(ev)
                    e.apply_erosion_devmin_dev(ez.ptr, N, N, mmz.ptr, D, pkg.ERODE_MINZ_IS_MIN)
            th = threading.Thread(target=job)
            th.start()
    if th is not None:
        th.join()


def sync():
    for c in [t, e] + tcs:
        c.synchronize()
steps(16); sync()
t0 = time.perf_counter(); steps(K); sync(); dt = (time.perf_counter() - t0) / K
print(f"world {W} steps {K} tracers {NT} shard {int(SHARD)}: {dt * 1e3:.4f} ms per step", flush=True)
