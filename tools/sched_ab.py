#!/usr/bin/env python3
"""How the heightmaps in flight are driven -- same-process A/B of host schedules for the headline step (16384^2 sine noise + min + 1000-droplet erosion).
Measured again in round 5 because the erosion's footprint changed (lean trace waves: 128 registers / 9.8 KB of LDS instead of 264 / 22.7 KB).

  threads    bench.py's schedule: P pipelines (context + host thread each) run noise + erosion of their maps, one noise call at a time (semaphore)
  producer   ONE thread runs every map's noise on its own context (terra_gen_grid_minmax_dev: min read back), P eroder threads erode
  streamed   the producer only enqueues (terra_gen_grid_minmax_async_dev: min stays in HBM, events order the streams), P eroder threads erode with terra_apply_erosion_devmin_dev
  noise      no erosion at all: the producer's loop alone (the floor of any schedule)

usage: sched_ab.py [--steps 20] [--reps 3] [--pipelines 3,4] [--schedules threads,producer,streamed,noise] [--size 16384] [--droplets 1000]"""
import argparse
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--pipelines", default="3,4")
    ap.add_argument("--schedules", default="threads,producer,streamed,noise")
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--droplets", type=int, default=1000)
    ap.add_argument("--warm-ms", type=float, default=150.0)
    a = ap.parse_args()
    N, D = a.size, a.droplets
    Pmax = max(int(x) for x in a.pipelines.split(","))
    cfg = pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1)
    nctx = pkg.Terra(0); st = nctx.init_scene(cfg)
    ctxs = [pkg.Terra(0) for _ in range(Pmax)]
    for c in ctxs:
        c.init_scene(cfg)
    zs = [nctx.alloc(N * N * 4) for _ in range(Pmax)]
    mms = [nctx.alloc(8) for _ in range(Pmax)]
    ev_noise = [nctx.event_create() for _ in range(Pmax)]
    ev_free = [ctxs[p].event_create() for p in range(Pmax)]
    G = (-N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N)

    def threads(k, P):
        turn = threading.Semaphore(1)
        lock = threading.Lock(); left = [k]
        started = [threading.Event() for _ in range(P)]

        def worker(p):
            if p:
                started[p - 1].wait()
            first = True
            while True:
                with lock:
                    if left[0] <= 0:
                        break
                    left[0] -= 1
                with turn:
                    mn, _ = ctxs[p].gen_grid_minmax_dev(zs[p].ptr, *G, pkg.GEN_GLACIATE)
                if first:
                    started[p].set(); first = False
                ctxs[p].apply_erosion_dev(zs[p].ptr, N, N, mn, D, pkg.ERODE_MINZ_IS_MIN)
            started[p].set()
        th = [threading.Thread(target=worker, args=(p,)) for p in range(P)]
        [x.start() for x in th]; [x.join() for x in th]

    def produced(k, P, streamed):
        ready = [threading.Semaphore(0) for _ in range(P)]
        free = [threading.Semaphore(1) for _ in range(P)]
        mins = [0.0] * P

        def eroder(p):
            for _ in range(p, k, P):
                ready[p].acquire()
                if streamed:
                    ctxs[p].event_wait(ev_noise[p])
                    ctxs[p].apply_erosion_devmin_dev(zs[p].ptr, N, N, mms[p].ptr, D, pkg.ERODE_MINZ_IS_MIN)
                    ctxs[p].event_record(ev_free[p])
                else:
                    ctxs[p].apply_erosion_dev(zs[p].ptr, N, N, mins[p], D, pkg.ERODE_MINZ_IS_MIN)
                free[p].release()
        th = [threading.Thread(target=eroder, args=(p,)) for p in range(min(P, k))]
        [x.start() for x in th]
        for i in range(k):
            p = i % P
            free[p].acquire()
            if streamed:
                if i >= P:
                    nctx.event_wait(ev_free[p])
                nctx.gen_grid_minmax_async_dev(zs[p].ptr, *G, mms[p].ptr, pkg.GEN_GLACIATE)
                nctx.event_record(ev_noise[p])
            else:
                mins[p], _ = nctx.gen_grid_minmax_dev(zs[p].ptr, *G, pkg.GEN_GLACIATE)
            ready[p].release()
        [x.join() for x in th]

    def noise_only(k, P):
        for i in range(k):
            nctx.gen_grid_minmax_dev(zs[i % P].ptr, *G, pkg.GEN_GLACIATE)

    fns = {"threads": threads, "producer": lambda k, P: produced(k, P, False), "streamed": lambda k, P: produced(k, P, True), "noise": noise_only}

    def sync():
        nctx.synchronize()
        for c in ctxs:
            c.synchronize()

    for rep in range(a.reps):
        for name in a.schedules.split(","):
            for P in ([int(x) for x in a.pipelines.split(",")] if name != "noise" else [Pmax]):
                fn = fns[name]
                t0 = time.perf_counter()
                fn(2 * P, P)
                while (time.perf_counter() - t0) * 1e3 < a.warm_ms:  # the chip's clock ramp (profiles/r04_clock_ramp.txt)
                    fn(P, P)
                sync()
                t0 = time.perf_counter()
                fn(a.steps, P)
                sync()
                dt = time.perf_counter() - t0
                print(f"{name:9s} P {P} K{a.steps}  {N * N * a.steps / dt / 1e9:8.2f} Gcells/s  {dt / a.steps * 1e3:7.4f} ms/step  (sparse={os.environ.get('TERRA_ERO_SPARSE', 'auto')})", flush=True)


if __name__ == "__main__":
    main()
