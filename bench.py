#!/usr/bin/env python3
"""bench.py -- heightmap Gcells/s (noise + erosion) on MI355X, BASELINE.json's metric.

A "step" = one pass of the hot path over one synthetic 16384 x 16384 heightmap per GPU, with heightmap_t::proc_gen
semantics (src/heightmap.cpp:130-187): build_arrays + enable_glaciate + eval of every cell (8-octave noise =
mesh_freq_filter 1 -> start_eval_sin 10), min(vals), apply_erosion(vals, N, N, min, 1000 droplets), all device resident.
The z grid never leaves HBM inside the timed region (there is no input grid; parameters are a few hundred bytes).

  python bench.py [--gpus N --steps K --warmup W --size 16384 --mode sine --droplets 1000]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (one rank per GPU)

Multi-GPU: every rank owns one independent N x N region of the world (origin shifted by rank*N cells in x), generated and
eroded exactly like the reference erodes each tile alone on its clamp-padded copy (src/tiled_mesh.cpp:515): no data-path
collective, weak scaling; torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.

Prints ONE JSON line on rank 0.  `roofline` = dominant kernel, measured with HIP events on the library's stream;
`cpu_baseline` = the reference's own CPU code (oracle/_ref) or the C restatement (oracle/) timed on this host, rank 0, N=1.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MODES = {"sine": 0, "simplex": 1, "perlin": 2, "dwarp": 4}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
VALU_PEAK_TOPS = 78.6      # fp32 VALU without FMA: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (mul and add are separate instructions here)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=64)
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--size", type=int, default=16384)
    p.add_argument("--mode", default="sine", choices=sorted(MODES))
    p.add_argument("--droplets", type=int, default=1000)
    p.add_argument("--octaves", type=int, default=8)
    p.add_argument("--pipelines", type=int, default=4, help="heightmaps in flight per GPU (each on its own HIP stream, like the reference's height_gens[8])")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-size", type=int, default=0, help="grid edge of the CPU sample (default: min(size, 8192))")
    return p.parse_args()


def cpu_baseline(args, mode):
    """The reference CPU path on this host's cores: same noise + erosion on a bounded sample (one grid)."""
    import numpy as np
    import orclib
    orclib.build_oracle()
    kind = "reference" if orclib.ref_available() else "port"
    ck = orclib.Checker("ref" if kind == "reference" else "orc")
    cores = ck.num_threads()
    n = args.cpu_size or min(args.size, 8192)
    s = ck.init(orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=9 - args.octaves))
    t0 = time.perf_counter()
    g = ck.gen_grid(-n / 2, -n / 2, s.DX_VAL, s.DY_VAL, n, n, 1)   # build_arrays + enable_glaciate + eval_index loop, OpenMP over all cores
    t1 = time.perf_counter()
    mn = float(g.min())
    t2 = time.perf_counter()
    ck.apply_erosion(g, mn, args.droplets)                          # reference apply_erosion incl. its pad / unpad copies
    t3 = time.perf_counter()
    total = (t1 - t0) + (t3 - t2)
    return {"value": round(n * n / total / 1e9, 6), "unit": "Gcells/s", "cores": cores, "kind": kind,
            "sample": f"one {n}x{n} grid, same seed/params: noise {t1 - t0:.3f}s + apply_erosion({args.droplets}) {t3 - t2:.3f}s, OMP threads={cores}",
            "noise_gcells_s": round(n * n / (t1 - t0) / 1e9, 6)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fall-back)")
    backend = os.environ.get("TERRA_BENCH_BACKEND", "nccl")  # "gloo" only to smoke-test the multi-rank orchestration on a box with fewer GPUs than ranks
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs: one rank per GPU")
    local_rank %= ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus or world == 1, "launch one rank per GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module("3dworld_amd")
    if not os.path.exists(pkg.default_lib_path()):
        raise SystemExit("libterra_hip.so missing: run __graft_entry__.build() (no CPU fall-back)")
    import threading
    mode = MODES[args.mode]
    N = args.size
    cells = N * N
    P = max(1, min(args.pipelines, args.steps))
    # P independent heightmaps in flight per GPU, each with its own context (HIP stream, scratch) and its own z grid in HBM.
    # A heightmap's erosion is a latency-bound chain of dependent droplet steps on ~1000 waves; the next heightmap's noise kernel
    # (VALU-bound, whole chip) runs beside it.  The reference keeps 8 generator objects in flight for the same reason (src/tiled_mesh.h:418).
    ctxs = [pkg.Terra(local_rank) for _ in range(P)]
    sts = [c.init_scene(pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=9 - args.octaves)) for c in ctxs]
    st = sts[0]
    t = ctxs[0]
    zs = [torch.empty(cells, dtype=torch.float32, device=dev) for _ in range(P)]
    z = zs[0]
    x0 = -N / 2 + rank * N  # each rank owns its own N x N region of the world
    y0 = -N / 2

    def step(p=0):
        # heightmap_t::proc_gen on the device: noise + glaciate (+ fused min) -> erosion (in place)
        c, zz = ctxs[p], zs[p]
        mn, _ = c.gen_grid_minmax_dev(zz.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)  # min(vals) is folded into the grid kernel
        c.apply_erosion_dev(zz.data_ptr(), N, N, mn, args.droplets, pkg.ERODE_MINZ_IS_MIN)                   # run_erosion passes min(vals): only written cells can need the clamp

    def run_steps(k):
        """k steps in total, dealt round-robin to the P pipelines (one host thread each: the library calls release the GIL)."""
        if P == 1:
            for _ in range(k):
                step(0)
            return
        def worker(p):
            for _ in range(p, k, P):
                step(p)
        th = [threading.Thread(target=worker, args=(p,)) for p in range(P)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    if True:
        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        barrier()
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        rep = t.erosion_report().as_dict()

        # ---- per-kernel times, live, HIP events on the same stream (rank 0 only)
        detail = {}
        if rank == 0:
            reps = max(3, args.steps)
            t.timer_start()
            for _ in range(reps):
                t.gen_grid_minmax_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
            ms_gen = t.timer_stop() / reps
            t.timer_start()
            for _ in range(reps):  # the grid kernel with its two small table kernels, no min/max read-back: what rocprofv3 reports as k_sine_grid / k_noise_grid (+ ~20 us)
                t.gen_grid_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
            ms_grid = t.timer_stop() / reps
            mn, _ = t.minmax_dev(z.data_ptr(), cells)
            t.timer_start()
            for _ in range(reps):
                t.minmax_dev(z.data_ptr(), cells)
            ms_minmax = t.timer_stop() / reps
            zc = z.clone()
            ms_ero = 0.0
            for _ in range(reps):
                z.copy_(zc)
                torch.cuda.synchronize(dev)
                t.timer_start()
                t.apply_erosion_dev(z.data_ptr(), N, N, mn, args.droplets, pkg.ERODE_MINZ_IS_MIN)
                ms_ero += t.timer_stop() / reps
            detail = {"ms_noise_kernels": round(ms_gen, 4), "ms_grid_kernel": round(ms_grid, 4), "ms_minmax_unfused": round(ms_minmax, 4), "ms_erosion": round(ms_ero, 4)}

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * cells * args.steps / dt / 1e9
        terms = 10 * args.octaves if mode == 0 else None
        # dominant kernel by time: the noise grid kernel (k_sine_grid / k_noise_grid). Algorithmic bytes: 4 B written per cell.
        ms_k = detail["ms_grid_kernel"]
        achieved = 4.0 * cells / (ms_k * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the PMC passes (profiles/r01_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE), only for the configuration they were taken on
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]["k_sine_grid"]
            if mode == 0 and N == 16384 and args.octaves == 8:
                traffic = pm["hbm_bytes_per_launch"]
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": "k_sine_grid (+table kernels)" if mode == 0 else f"k_noise_grid<{args.mode}>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes": 4 * cells,
                "note": "kernel is fp32-VALU bound, not HBM bound: see valu_frac (mul and add issue separately because the CPU reference has no FMA)"}
        if terms:
            ops = 2.0 * terms * cells / (ms_k * 1e-3) / 1e12
            roof["valu_tops"] = round(ops, 2); roof["valu_peak_tops"] = VALU_PEAK_TOPS; roof["valu_frac"] = round(ops / VALU_PEAK_TOPS, 4)
            # the HBM fraction this kernel could reach at 100 % of the (nominal) non-FMA VALU peak: 4 B per 2*terms ops
            roof["hbm_frac_ceiling_when_valu_bound"] = round(4.0 / (2.0 * terms) * VALU_PEAK_TOPS * 1e12 / (HBM_PEAK_GBS * 1e9), 4)
        out = {"metric": "heightmap Gcells/sec (noise+erosion), 16384^2 grid", "value": round(value, 4), "unit": "Gcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{N}x{N} heightmap per GPU, {args.mode} noise {args.octaves} octaves + glaciate/islands, min(vals), {args.droplets}-droplet erosion (heightmap_t::proc_gen semantics), device resident",
                          "mesh_gen_mode": mode, "octaves": args.octaves, "droplets": args.droplets, "grid": N, "pipelines_per_gpu": P, "parallelism": f"{world} independent regions (one per GPU), no collective; {P} heightmaps in flight per GPU"},
               "roofline": roof, "detail": dict(detail, erosion=rep)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, mode)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
