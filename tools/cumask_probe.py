#!/usr/bin/env python3
"""CU partitioning instead of time sharing: the noise context on a stream whose CU mask leaves R compute units out, the eroder contexts on streams that may use ONLY those
(hipExtStreamCreateWithCUMask).  One 16384^2 heightmap per step, noise enqueued back to back, the erosion of map i beside the noise of map i + 1 ...  ms per step for several R.
usage: cumask_probe.py [steps=40] [R list, e.g. 0,8,16,32] [pattern: low|spread]"""
import ctypes as C, importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
RS = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,8,16,32").split(",")]
PAT = sys.argv[3] if len(sys.argv) > 3 else "low"
hip = C.CDLL("libamdhip64.so")
N, D, G, E = 16384, 1000, 4, 2
cfg = pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1)


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


for R in RS:
    if PAT == "low":
        reserved = (1 << R) - 1
    else:  # spread: every (256 / R)-th CU
        reserved = 0
        for i in range(R):
            reserved |= 1 << (i * (256 // max(R, 1)))
    allb = (1 << 256) - 1
    t = pkg.Terra(0); st = t.init_scene(cfg)
    es = [pkg.Terra(0) for _ in range(E)]
    for e in es:
        e.init_scene(cfg)
    streams = []
    if R:
        s_n = masked_stream(allb & ~reserved); t.set_stream(s_n.value); streams.append(s_n)
        for e in es:
            s_e = masked_stream(reserved); e.set_stream(s_e.value); streams.append(s_e)
    zs = [t.alloc(N * N * 4) for _ in range(G)]
    mms = [t.alloc(8) for _ in range(G)]
    evs = [t.event_create() for _ in range(G)]
    done = [threading.Event() for _ in range(G)]
    for d in done:
        d.set()

    def steps(k):
        ths = []
        for s in range(k):
            g = s % G
            done[g].wait(); done[g].clear()
            t.gen_grid_minmax_async_dev(zs[g].ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, mms[g].ptr, pkg.GEN_GLACIATE)
            t.event_record(evs[g])
            e = es[s % E]

            def job(e=e, g=g):
                e.event_wait(evs[g])
                e.apply_erosion_devmin_dev(zs[g].ptr, N, N, mms[g].ptr, D, pkg.ERODE_MINZ_IS_MIN)
                e.synchronize()
                done[g].set()
            th = threading.Thread(target=job); th.start(); ths.append(th)
        for th in ths:
            th.join()
        t.synchronize()
    steps(12)
    t0 = time.perf_counter(); steps(K); dt = (time.perf_counter() - t0) / K
    print(f"R {R} ({PAT}): {dt * 1e3:.4f} ms per step  {N * N / dt / 1e9:.1f} Gcells/s", flush=True)
    for c in [t] + es:
        c.synchronize(); c.set_stream(None); c.close()
