#!/usr/bin/env python3
"""bench.py -- heightmap Gcells/s (noise + erosion) on MI355X, BASELINE.json's metric.

A "step" = one pass of the hot path over one synthetic 16384 x 16384 heightmap per GPU, with heightmap_t::proc_gen
semantics (src/heightmap.cpp:130-187): build_arrays + enable_glaciate + eval of every cell (8-octave noise =
mesh_freq_filter 1 -> start_eval_sin 10), min(vals), apply_erosion(vals, N, N, min, 1000 droplets), all device resident.
The z grid never leaves HBM inside the timed region (there is no input grid; parameters are a few hundred bytes).

  python bench.py [--gpus N --steps K --warmup W --size 16384 --mode sine --droplets 1000]   (N > 1: starts N ranks itself through torch.distributed.run)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (one rank per GPU; WORLD_SIZE must equal --gpus)

`value` (the headline):
  N = 1   one 16384^2 heightmap per step on the GPU, `pipelines` heightmaps in flight (own context / stream / host thread each).
  N > 1   ONE 16384^2 heightmap per step on all N GPUs TOGETHER, erosion included -- STRONG scaling (`value_strong`): rank r evaluates its row strip into its own HBM, min(vals)
          is one float through all_reduce(min) over RCCL, and the step's grid is a terra_dgrid (every rank's strip mapped back to back on every rank, HIP virtual memory
          management): rank (step mod N) erodes the whole grid in serial droplet order over the mapped pointer, reaching the other ranks' rows over xGMI, while the next steps'
          noise runs everywhere (3dworld_amd/dist.py::OneHeightmapPipeline).  Bit-identical to the single-GPU heightmap.  The independent-regions number of round 1-3 (every
          rank its own N x N region, no collective, weak scaling) is measured in the same run and printed as `value_weak` / detail.regions.

The same run also measures, under `detail` (incl. `dense_erosion`: 10^6 droplets on the bench grid and on BASELINE config 3's 4096^2 map) (all ranks take part, rank 0 reports; switch off with --no-extras):
  single   one heightmap in flight (no overlap of a map's erosion with the next map's noise): the latency of one map
  strips   the noise + min part of the one-grid line alone: rank r evaluates rows [r*N/W, (r+1)*N/W) (terra_gen_grid_rows_minmax_dev, bit-identical to the
           full grid), min(vals) = one float through all_reduce(min) over RCCL; no erosion
  tiles    STRONG scaling of BASELINE config 4: the 64 x 64 tiles of 128^2 block-partitioned over the ranks
           (tile_t::create_zvals + stats + normals; with 0 and with 1000 droplets per tile), no collective
  voxels   STRONG scaling of BASELINE config 5: one 512^3 voxel noise field as y slabs (terra_voxel_fill_slab_dev), no collective
  modes    the same 16384^2 step with simplex / Perlin / domain-warp noise (rank 0)
`--workload onegrid|regions|strips|tiles` makes one of those the headline `value` instead (for a scaling sweep of that mode alone; onegrid also at N = 1).

Prints ONE JSON line on rank 0.  `roofline` = dominant kernel, measured with HIP events on the library's stream;
`cpu_baseline` = the reference's own CPU code (oracle/_ref) or the C restatement (oracle/) timed on this host, rank 0, N=1.
"""
import argparse
import importlib
import json
import os
import sys
import threading
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_detail  # noqa: E402 -- the secondary measurements of the same run (detail.*)
from bench_detail import flops_per_cell  # noqa: E402

MODES = {"sine": 0, "simplex": 1, "perlin": 2, "dwarp": 4}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3   # fp32 vector peak (packed FMA): 256 CU x 4 SIMD x 16 lanes x 2 (packed) x 2 (FMA) x 2.4 GHz
VALU_NOFMA_TOPS = 78.6     # the same without fusing: the CPU reference rounds the product before adding, so mul and add are separate instructions here


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
    p.add_argument("--headline-only", action="store_true", help="stop after the timed headline run (for kernel traces of exactly that region): no per-kernel section, no roofline in the line")
    p.add_argument("--handoff", default="event", choices=["event", "semaphore", "none"], help="how the heightmaps in flight take turns in their noise phase: event = the next map's thread waits on the host for the "
                   "GPU event behind the previous map's noise kernel, min(vals) stays in HBM (3dworld_amd/pipeline.py); semaphore = a host semaphore around a synchronous noise call (rounds 1-4); none = no turns")
    p.add_argument("--workload", default="heightmap", choices=["heightmap", "onegrid", "regions", "strips", "tiles"],
                   help="which measurement is the headline `value`.  heightmap (default): at N = 1 one 16384^2 heightmap per step on the GPU; at N > 1 ONE 16384^2 heightmap per step on all "
                        "GPUs together, erosion included (= onegrid, strong scaling), with the independent-regions number (= regions, weak scaling) beside it as value_weak")
    p.add_argument("--shard-ab", action="store_true", help="onegrid at N > 1: also time the line with the sparse erosion's traces made by the strip owners and report the faster form as `value` (default at N > 1: off)")
    p.add_argument("--no-shard-ab", action="store_true", help="onegrid: do not time the second form of the line (the sparse erosion's traces made by the strip owners, terra_erosion_shard_*)")
    p.add_argument("--grids-in-flight", type=int, default=8, help="onegrid: distributed grids in flight (a grid is reused this many steps later)")
    p.add_argument("--tile-droplets", type=int, default=0, help="--workload tiles: erosion_iters_tt of the headline tile batch")
    p.add_argument("--no-extras", action="store_true", help="skip the single / strips / tiles / modes measurements under `detail`")
    p.add_argument("--preflight", action="store_true", help="N > 1 smoke of every cross-device path in < 30 s: process group, one collective, the one-grid mapping (HIP VMM export / import / "
                   "peer access), a peer copy, one sharded step; rank 0 prints {\"preflight\": ...} naming the first failing call, then the process exits (no bench line)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-rccl-world1", action="store_true", help="N = 1: do not create the one-rank RCCL group (the collectives of the sharded paths are then skipped)")
    p.add_argument("--clock-warmup-ms", type=float, default=150.0, help="untimed steps run for this long right before every timed region so that it runs at the chip's sustained clock (a cold MI355X needs ~30 ms of load to get there, an idle gap of 5 ms already costs 12 %%: profiles/r04_clock_ramp.txt); 0 = only the W warm-up steps")
    p.add_argument("--cpu-size", type=int, default=0, help="grid edge of the CPU sample (default: the benchmark's own size)")
    p.add_argument("--simulate-world", type=int, default=8, help="detail.onegrid_rank_floor: what one rank of the one-grid line does per step at this world size (its 1/W row strip of noise + the "
                   "all_reduce(min) + every W-th step an erosion), measured on this GPU: the floor of the N = W step and the prediction that follows from it")
    return p.parse_args()


def cpu_baseline(args, mode):
    """The reference CPU path on this host's cores: the SAME grid (size, octaves, droplets), all OpenMP threads; then one thread (the only
    deterministic erosion order) on a bounded sample: the noise on the first 1/16 of the rows, the erosion on the whole grid."""
    import numpy as np
    import orclib
    orclib.build_oracle()
    kind = "reference" if orclib.ref_available() else "port"
    ck = orclib.Checker("ref" if kind == "reference" else "orc")
    cores = ck.num_threads()
    n = args.cpu_size or args.size
    s = ck.init(orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=9 - args.octaves))
    t0 = time.perf_counter()
    g = ck.gen_grid(-n / 2, -n / 2, s.DX_VAL, s.DY_VAL, n, n, 1)   # build_arrays + enable_glaciate + eval_index loop, OpenMP over all cores
    t1 = time.perf_counter()
    mn = float(g.min())
    g1 = g.copy()
    t2 = time.perf_counter()
    ck.apply_erosion(g, mn, args.droplets)                          # reference apply_erosion incl. its pad / unpad copies
    t3 = time.perf_counter()
    total = (t1 - t0) + (t3 - t2)
    out = {"value": round(n * n / total / 1e9, 6), "unit": "Gcells/s", "cores": cores, "kind": kind,
           "sample": f"one {n}x{n} grid, same seed/params: noise {t1 - t0:.3f}s + apply_erosion({args.droplets}) {t3 - t2:.3f}s, OMP threads={cores}",
           "noise_gcells_s": round(n * n / (t1 - t0) / 1e9, 6)}
    try:
        ck.set_num_threads(1)
        rows = max(1, n // 16)
        t4 = time.perf_counter()
        ck.gen_grid(-n / 2, -n / 2, s.DX_VAL, s.DY_VAL, n, rows, 1)
        t5 = time.perf_counter()
        ck.apply_erosion(g1, mn, args.droplets)
        t6 = time.perf_counter()
        noise_1 = (t5 - t4) * (n / rows)
        out["threads_1"] = {"value": round(n * n / (noise_1 + (t6 - t5)) / 1e9, 6), "unit": "Gcells/s", "cores": 1,
                            "sample": f"noise on the first {rows} of {n} rows ({t5 - t4:.3f}s, scaled x{n // rows}) + apply_erosion({args.droplets}) on the whole {n}x{n} grid {t6 - t5:.3f}s, OMP threads=1",
                            "noise_gcells_s": round(n * n / noise_1 / 1e9, 6)}
    finally:
        ck.set_num_threads(cores)
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_cmd(n, argv):
    """the launcher line for N ranks on this node: the same one the driver uses for N > 1"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
            os.path.abspath(__file__)] + list(argv)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU"""
    cmd = launch_cmd(n, sys.argv[1:])
    os.execv(sys.executable, cmd)


class c_stdout_to_stderr:
    """RCCL prints a version banner on the C stdout when its first communicator comes up (buffered, so it would surface after the JSON line at exit): while this
    is active the process's fd 1 is stderr, and the C buffers are flushed before fd 1 is put back -- stdout stays exactly ONE JSON line"""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


def preflight(args, torch, dist, pkg, dmod, rank, world, local_rank, dev, coll_dev, backend):
    """Every path a multi-GPU run crosses a device boundary on, once, small, each stage named: what first contact with an 8-GPU node should run before the scaling bench.
    A stage that raises on any rank is reported by rank 0 with the exception text (the ranks agree on the outcome through one all_reduce per stage, so nobody hangs)."""
    import traceback
    stages, failed = [], [None]

    def agree(ok):
        if dist is None:
            return ok
        f = torch.tensor([0 if ok else 1], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(f)
        return int(f.item()) == 0

    def stage(name, fn):
        if failed[0]:
            return
        t0, err = time.perf_counter(), None
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
            traceback.print_exc(file=sys.stderr)
        ok = agree(err is None)
        stages.append({"stage": name, "ok": ok, "ms": round((time.perf_counter() - t0) * 1e3, 1), **({"error": err} if err else {})})
        if not ok:
            failed[0] = name

    st_box, t_box, grid_box = {}, {}, {}

    def s_context():
        t_box["t"] = pkg.Terra(local_rank)
        st_box["st"] = t_box["t"].init_scene(pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1))

    def s_collective():
        if dist is None:
            return
        v = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=coll_dev)
        dist.all_reduce(v)
        assert float(v[0]) == world * (world + 1) / 2, f"all_reduce gave {float(v[0])}"

    def s_onegrid():
        n = 1024
        g, rows = dmod.create_distributed_grid(pkg.terra, t_box["t"], dist, n, n, "preflight", coll_dev)  # HIP VMM: export / import / map / peer access
        grid_box["g"], grid_box["rows"], grid_box["n"] = g, rows, n

    def s_strip_fill_and_peer_read():
        g, rows, n, t, st = grid_box["g"], grid_box["rows"], grid_box["n"], t_box["t"], st_box["st"]
        r0, r1 = rows[rank]
        t.gen_grid_rows_minmax_dev(g.ptr + r0 * n * 4, -n / 2, -n / 2, st.DX_VAL, st.DY_VAL, n, n, r0, r1 - r0)  # this rank's rows into its own HBM
        t.synchronize()
        if dist is not None:
            dist.barrier()
        peer = (rank + 1) % world
        p0, p1 = rows[peer]
        np = __import__("numpy")
        got = np.empty(n, np.float32)
        t._ck(t.lib.terra_memcpy_d2h(t.ctx, got.ctypes.data, g.ptr + p0 * n * 4, n * 4))  # one row of the neighbour's strip, read through the mapping
        tmp = t.alloc(n * 4)  # the same row computed here (the rows entry point is bit-identical to the full grid's rows: tests/test_gpu_parity.py)
        t.gen_grid_rows_minmax_dev(tmp.ptr, -n / 2, -n / 2, st.DX_VAL, st.DY_VAL, n, n, p0, 1)
        t.synchronize()
        want = tmp.download(np.float32, (n,))
        tmp.free()
        bad = int((got != want).sum())
        assert bad == 0, f"a row read through the peer mapping differs from the row computed locally (rank {rank}, row {p0} of rank {peer}: {bad} of {n} cells, first got {got[:3].tolist()} want {want[:3].tolist()})"

    def s_erode_across_strips():
        g, n, t = grid_box["g"], grid_box["n"], t_box["t"]
        if dist is not None:
            dist.barrier()
        if rank == 0:
            t.apply_erosion_dev(g.ptr, n, n, -1.0e9, 200)  # droplets walk rows that live in every rank's HBM
            t.synchronize()
        if dist is not None:
            dist.barrier()

    def s_sharded_traces():
        # terra_erosion_shard_*: every rank traces the droplets that start in its rows into its arena (one more mapped array), rank 0 gathers them through the mapping and
        # commits; compared with the same erosion by one context on a private copy of the grid
        g, rows, n, t = grid_box["g"], grid_box["rows"], grid_box["n"], t_box["t"]
        np = __import__("numpy")
        D = 40
        t.gen_grid_rows_minmax_dev(g.ptr + rows[rank][0] * n * 4, -n / 2, -n / 2, st_box["st"].DX_VAL, st_box["st"].DY_VAL, n, n, rows[rank][0], rows[rank][1] - rows[rank][0])
        t.synchronize()
        if dist is not None:
            dist.barrier()
        ag, _ = dmod.create_distributed_grid(pkg.terra, t, dist, 0, 0, "preflight_arena", coll_dev, strip_bytes=t.erosion_shard_arena_bytes(D))  # (raises on every rank together)
        err = None  # (whatever happens on one rank, every rank passes the same two barriers: the stage's outcome is agreed on afterwards)
        try:
            ref = t.alloc(n * n * 4)
            whole = np.empty((n, n), np.float32)
            t._ck(t.lib.terra_memcpy_d2h(t.ctx, whole.ctypes.data, g.ptr, whole.nbytes))
            ref.upload(whole)
            dmin = t.alloc(8).upload(np.array([-1.0e9, 0.0], np.float32))
            t.erosion_shard_trace_dev(g.ptr, n, n, D, rows[rank][0], rows[rank][1] - rows[rank][0], ag.strip_ptr(rank))
            t.synchronize()
        except Exception as e:  # noqa: BLE001
            err = e
        if dist is not None:
            dist.barrier()
        if rank == 0 and err is None:
            try:
                t.erosion_shard_finish_dev(g.ptr, n, n, dmin.ptr, D, 0, world, 0, [r1 for (_, r1) in rows], ag.strip_ptr(0), ag.strip_bytes[0])
                t.apply_erosion_devmin_dev(ref.ptr, n, n, dmin.ptr, D, 0)
                t.synchronize()
                got = np.empty((n, n), np.float32)
                t._ck(t.lib.terra_memcpy_d2h(t.ctx, got.ctypes.data, g.ptr, got.nbytes))
                want = ref.download(np.float32, (n, n))
                assert (got.view(np.uint32) == want.view(np.uint32)).all(), "the sharded erosion's grid differs from one context's"
            except Exception as e:  # noqa: BLE001
                err = e
        if dist is not None:
            dist.barrier()
        ag.destroy()
        if err is not None:
            raise err

    stage("context", s_context)
    stage("collective", s_collective)
    stage("onegrid_vmm_mapping", s_onegrid)
    stage("strip_fill_and_peer_read", s_strip_fill_and_peer_read)
    stage("erode_across_strips", s_erode_across_strips)
    stage("sharded_traces", s_sharded_traces)
    try:
        if "g" in grid_box:
            grid_box["g"].destroy()
        if "t" in t_box:
            t_box["t"].close()
    except Exception:  # noqa: BLE001
        pass
    if rank == 0:
        print(json.dumps({"preflight": "ok" if not failed[0] else "failed", "failed_stage": failed[0], "n_gpus": world, "backend": backend, "stages": stages}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if failed[0]:
        raise SystemExit(3)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)  # does not return
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: one rank per GPU, the line's n_gpus must be the N that was asked for")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fall-back)")
    backend = os.environ.get("TERRA_BENCH_BACKEND", "nccl")  # "gloo" only to smoke-test the multi-rank orchestration on a box with fewer GPUs than ranks
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs: one rank per GPU")
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_note = None
    # Rendezvous without name resolution: the ranks of one node meet in a FileStore under /tmp (N = 1: an in-process HashStore), RCCL / gloo bootstrap over the loopback
    # interface.  torch's TCPStore looks the client's host name up for every connection; on a box whose resolver does not answer that alone took 105 s for the one-rank
    # group (BENCH_r04: "hostname of the client socket cannot be retrieved").  Bounded: a group that is not up after the timeout is reported (N = 1) or fatal (N > 1).
    import datetime
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")

    def init_group(timeout_s):
        kw = {"device_id": dev} if backend == "nccl" else {}
        if world == 1:
            store = dist.HashStore()
        else:  # every local rank has the same launcher as its parent: one file per launch, never a stale one
            store = dist.FileStore(os.path.join("/tmp", f"terra_bench_store_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{world}"), world)
        dist.init_process_group(backend=backend, store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
        probe = torch.tensor([3.5 + rank], dtype=torch.float32, device=dev if backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(probe, op=dist.ReduceOp.MIN)  # the communicator (and RCCL's banner) comes up here, not inside the timed region
        if backend == "nccl":
            torch.cuda.synchronize(dev)
        assert float(probe.item()) == 3.5

    if world > 1:
        with c_stdout_to_stderr():
            try:
                init_group(180)
            except Exception as e:  # noqa: BLE001 -- e.g. /tmp not shared by the ranks: the launcher's own store (env://) is the fall-back
                print(f"[bench] FileStore rendezvous failed ({e!r}); falling back to env://", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))
    elif backend == "nccl" and not args.no_rccl_world1:
        # N = 1: a one-rank RCCL group, so that the device-tensor collectives of the sharded paths (all_reduce(min) of the strips, the max over ranks of the
        # step time, barriers) run through RCCL on the 1-GPU box too instead of being skipped; a failure to set it up is reported, not fatal
        try:
            t0 = time.perf_counter()
            with c_stdout_to_stderr():
                init_group(10)
                dist.barrier()
            rccl_note = {"world1_group": "ok", "store": "HashStore (no sockets, no name resolution)", "init_plus_first_all_reduce_ms": round((time.perf_counter() - t0) * 1e3, 1)}
        except Exception as e:  # noqa: BLE001
            rccl_note = {"world1_group": f"unavailable: {e!r}"[:300]}
            if dist.is_initialized():
                dist.destroy_process_group()
    have_group = dist.is_initialized()
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    pkg = importlib.import_module("3dworld_amd")
    if not os.path.exists(pkg.default_lib_path()):
        raise SystemExit("libterra_hip.so missing: run __graft_entry__.build() (no CPU fall-back)")
    dmod = importlib.import_module("3dworld_amd.dist")
    if args.preflight:
        preflight(args, torch, dist if have_group else None, pkg, dmod, rank, world, local_rank, dev, coll_dev, backend)
        return
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

    pmod = importlib.import_module("3dworld_amd.pipeline")
    noise_turn = threading.Semaphore(1) if args.handoff == "semaphore" else None
    turns = pmod.NoiseTurns() if args.handoff == "event" else None
    evs = [c.event_create() for c in ctxs]                                           # "the noise kernel of pipeline p's current map has finished"
    mms = [torch.zeros(2, dtype=torch.float32, device=dev) for _ in range(P)]         # {min, max} of the map in pipeline p: written by its noise kernel, read by its erosion's clamp

    def step(p=0, noise_done=None):
        # heightmap_t::proc_gen on the device: noise + glaciate (+ fused min) -> erosion (in place); one map's noise phase at a time (3dworld_amd/pipeline.py)
        c, zz = ctxs[p], zs[p]
        if args.handoff == "event":
            pmod.proc_gen_step(pkg, c, turns if P > 1 else None, evs[p], zz.data_ptr(), mms[p].data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, args.droplets,
                               on_noise_enqueued=noise_done.set if noise_done is not None else None)
            return
        # --handoff semaphore / none (round 1-4's schedules, kept for the A/B): min(vals) read back by the host, a host semaphore around the noise call / nothing
        if noise_turn is not None:
            with noise_turn:
                mn, _ = c.gen_grid_minmax_dev(zz.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        else:
            mn, _ = c.gen_grid_minmax_dev(zz.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        if noise_done is not None:
            noise_done.set()
        c.apply_erosion_dev(zz.data_ptr(), N, N, mn, args.droplets, pkg.ERODE_MINZ_IS_MIN)

    def run_steps(k, npipe):
        """k steps in total on npipe pipelines (one host thread each: the library calls release the GIL)."""
        if npipe == 1:
            for _ in range(k):
                step(0)
            return
        # the pipelines start one noise phase apart (worker p issues its first step when worker p-1's first noise call has returned): started together
        # they run in lockstep -- four noise kernels sharing the chip, then four erosions leaving it idle -- and only drift into an overlapping
        # pattern after dozens of steps
        # The k steps are a shared queue: a pipeline takes the next one when it is free (the GPU does not serve the four streams evenly -- a host-clock trace shows one stream's
        # noise kernel waiting behind a dozen of the others' -- so fixed shares would leave that pipeline's steps for the end of the run; measured equal on average).
        first_noise_done = [threading.Event() for _ in range(npipe)]
        lock = threading.Lock()
        left = [k]
        def take():
            with lock:
                if left[0] <= 0:
                    return False
                left[0] -= 1
                return True
        errs = []

        def worker(p):
            first = True
            try:
                if p > 0:
                    first_noise_done[p - 1].wait()
                while not errs and take():
                    step(p, first_noise_done[p] if first else None)
                    first = False
            except Exception as e:  # noqa: BLE001 -- reported by the caller's thread; nobody is left waiting for this worker
                errs.append(repr(e))
            finally:
                first_noise_done[p].set()  # also when this worker had no step at all
        th = [threading.Thread(target=worker, args=(p,)) for p in range(npipe)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if errs:
            raise RuntimeError("pipeline worker failed: " + "; ".join(errs))

    def barrier():
        torch.cuda.synchronize(dev)
        for c in ctxs:
            c.synchronize()
        if have_group:
            dist.barrier()

    spin_log = {}

    def spin_up(fn, warm, what):
        """untimed steps until --clock-warmup-ms have passed (at least `warm`): the chip's clocks ramp up over ~30 ms of load and fall back within a few ms of idling, and
        everything timed here lasts only 5-20 ms -- without this the timed region of a 20-step run IS the ramp (1.05 instead of 0.9 ms per step)"""
        t_s, n = time.perf_counter(), 0
        fn(warm); n += warm
        while True:
            el = (time.perf_counter() - t_s) * 1e3
            if world > 1:  # the steps of some workloads contain collectives: every rank must run the same number of them, so the ranks agree on "enough" (4 bytes, untimed)
                e = torch.tensor([el], dtype=torch.float32, device=coll_dev)
                dist.all_reduce(e, op=dist.ReduceOp.MIN)
                el = float(e.item())
            if el >= args.clock_warmup_ms:
                break
            fn(max(1, warm)); n += max(1, warm)
        spin_log[what] = n

    def timed(fn, k, warm, what="headline"):
        """warm untimed calls (+ the clock spin-up), then EXACTLY k timed ones between barrier + synchronize on both sides; max over ranks (seconds)."""
        spin_up(fn, warm, what)
        barrier()
        t0 = time.perf_counter()
        fn(k)
        torch.cuda.synchronize(dev)
        for c in ctxs:
            c.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        if have_group:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    all_tiles = [(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)]
    env = types.SimpleNamespace(args=args, pkg=pkg, dmod=dmod, dist=dist, torch=torch, dev=dev, rank=rank, world=world, have_group=have_group, coll_dev=coll_dev, ctxs=ctxs, zs=zs,
                                st=st, t=t, z=z, N=N, cells=cells, mode=mode, MODES=MODES, P=P, x0=x0, y0=y0, timed=timed, backend_name=("RCCL" if backend == "nccl" else backend),
                                all_tiles=all_tiles, my_tiles=dmod.partition_tiles(all_tiles, rank, world), tile_bufs={}, local_rank=local_rank)
    strips_steps, _ = bench_detail.strips_steps_fn(env)
    nt = len(env.my_tiles)

    K, W = args.steps, args.warmup
    detail = {}
    value_weak = value_strong = None
    # ---- ONE heightmap per step on all ranks together, erosion included: the strips live in their owners' HBM and are mapped back to back on every rank
    # (terra_dgrid), min(vals) is one float through all_reduce(min), rank s % world erodes step s's grid over the mapped pointer (remote rows over xGMI) while the next
    # steps' noise runs; 3dworld_amd/dist.py::OneHeightmapPipeline.  Bit-identical to the single-GPU heightmap (tests/test_distributed.py).
    want_onegrid = args.workload == "onegrid" or (args.workload == "heightmap" and (world > 1 or not args.no_extras))  # at N = 1 it is measured too (value_strong), so that 1 -> N compares like with like
    pipe = None
    if want_onegrid:
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getpid() if world == 1 else 'job'}"
        try:
            pipe = dmod.OneHeightmapPipeline(pkg, lambda: pkg.Terra(local_rank), pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=9 - args.octaves), dist if have_group and world > 1 else None,
                                             N, N, args.droplets, tag=tag, grids=max(2, args.grids_in_flight), eroders=2, coll_device=coll_dev)
        except RuntimeError as e:  # raised on every rank together (dist.py::create_distributed_grid): e.g. a runtime without virtual memory management between these devices
            if args.workload == "onegrid":
                raise
            detail["onegrid_unavailable"] = str(e)[:400]  # the line then carries the independent regions as `value` and says so
    # ---- the headline
    if args.workload in ("heightmap", "regions", "onegrid"):
        workload_w = f"{N}x{N} heightmap per GPU, {args.mode} noise {args.octaves} octaves + glaciate/islands, min(vals), {args.droplets}-droplet erosion (heightmap_t::proc_gen semantics), device resident"
        noise_turn_note = f", one in its noise phase at a time (hand-over: {args.handoff})" if (args.handoff != "none" and P > 1) else ""
        par_w = f"{world} independent regions (one per GPU), no collective; {P} heightmaps in flight per GPU (one host thread each{noise_turn_note})"
        workload_s = (f"ONE {N}x{N} heightmap per step on {world} GPU(s) together: {args.mode} noise {args.octaves} octaves + glaciate/islands as row strips in their owners' HBM, min(vals) by "
                      f"all_reduce(min), {args.droplets}-droplet erosion of the whole grid in serial droplet order by rank (step mod {world}) over the mapped strips (heightmap_t::proc_gen semantics)")
        par_s = (f"{world} row strips of {N // world} rows mapped back to back on every rank (terra_dgrid, HIP virtual memory management; remote rows over xGMI), one 4-byte all_reduce(min) per step over "
                 + (("RCCL" if backend == "nccl" else backend) if have_group else "nothing (one rank)") + f", {max(2, args.grids_in_flight)} grids in flight, eroders rotate over the ranks")
        if args.workload != "onegrid":
            dt = timed(lambda k: run_steps(k, P), K, max(W, 2 * P))  # at least two untimed steps per pipeline: first-use allocations, graph captures and clocks settle
            value_weak = world * cells * K / dt / 1e9
            value, scaling, workload, par = value_weak, "weak", workload_w, par_w
        if pipe is not None:
            dts = timed(lambda k: pipe.run(k), K, max(W, 2), "onegrid")
            value_strong = cells * K / dts / 1e9
            if value_weak is not None:
                detail["regions"] = {"value_weak": round(value_weak, 4), "ms_per_step": round(dt / K * 1e3, 4), "scaling": "weak", "workload": workload_w, "parallelism": par_w}
            # the same line with the sparse erosion scheduler's read-only phases made by the strip owners (terra_erosion_shard_*, dist.py shard_traces): same grids, bit for
            # bit; which form is faster at N > 1 is a question for the hardware (remote window traffic of the traces vs a second collective per step), so both are timed and
            # the line carries the faster one as `value` and both in detail
            # (timed by default on ONE GPU, where every part of it runs in this repository's tests; at N > 1 only with --shard-ab: its cross-device parts -- a second
            # communicator, arenas read through peer mappings -- have never met a multi-GPU node, and a run that measures the scaling curve must not depend on them)
            if (world == 1 and not args.no_shard_ab) or args.shard_ab:
                pipe2 = None
                try:
                    pipe2 = dmod.OneHeightmapPipeline(pkg, lambda: pkg.Terra(local_rank), pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=9 - args.octaves), dist if have_group and world > 1 else None,
                                                      N, N, args.droplets, tag=tag + "_sh", grids=max(2, args.grids_in_flight), eroders=2, coll_device=coll_dev, shard_traces=True)
                    dts2 = timed(lambda k: pipe2.run(k), K, max(W, 2), "onegrid")
                    detail["onegrid_sharded_traces"] = {"value_strong": round(cells * K / dts2 / 1e9, 4), "ms_per_step": round(dts2 / K * 1e3, 4), "plain_value_strong": round(value_strong, 4),
                                                        "what": "every rank probes / traces the droplets that start in its rows into its own arena behind the step's all_reduce, a second collective, the eroding rank gathers the traces and checks / commits"}
                    if dts2 < dts and (world > 1 or args.workload == "onegrid"):
                        dts, value_strong = dts2, cells * K / dts2 / 1e9
                        par_s += "; the sparse erosion scheduler's probe / trace phases made by the strip owners (terra_erosion_shard_*), one more 4-byte collective per step"
                except RuntimeError as e:  # (raised on every rank together)
                    detail["onegrid_sharded_traces"] = {"error": str(e)[:400]}
                finally:
                    if pipe2 is not None:
                        pipe2.close()
            if world > 1 or args.workload == "onegrid":
                dt, value, scaling, workload, par = dts, value_strong, "strong", workload_s, par_s
            else:  # N = 1: the region and the one grid are the same heightmap; the headline stays the 4-pipeline form, the one-grid pipeline (1 rank) is printed beside it
                detail["onegrid"] = {"value_strong": round(value_strong, 4), "ms_per_step": round(dts / K * 1e3, 4), "scaling": "strong", "workload": workload_s, "parallelism": par_s}
    elif args.workload == "strips":
        dt = timed(strips_steps, K, max(W, 2))
        value = cells * K / dt / 1e9
        scaling = "strong"
        workload = f"ONE {N}x{N} heightmap as {world} row strips, {args.mode} noise {args.octaves} octaves + glaciate/islands, min(vals) by all_reduce(min); erosion excluded (does not shard: replicas only)"
        par = f"{world} row strips of {N // world} rows, one 4-byte all_reduce(min) per step over " + (("RCCL" if backend == "nccl" else backend) if have_group else "nothing (no process group)")
    else:
        dt = timed(bench_detail.tiles_steps_fn(env, args.tile_droplets), K, max(W, 2))
        value = len(all_tiles) * 130 * 130 * K / dt / 1e9
        scaling = "strong"
        workload = f"64x64 tiles of 128^2 (tile_t::create_zvals + sub-block stats + normals, {args.tile_droplets} droplets per tile), block-partitioned over {world} GPUs"
        par = f"{nt} tiles on rank 0 of {len(all_tiles)}, no collective"
    rep = t.erosion_report().as_dict()
    if args.headline_only:
        barrier()
        if rank == 0:
            print(json.dumps({"metric": "heightmap Gcells/sec (noise+erosion), 16384^2 grid", "value": round(value, 4), "unit": "Gcells/s", "n_gpus": world, "steps": K, "warmup": W,
                              "value_strong": None if value_strong is None else round(value_strong, 4), "value_weak": None if value_weak is None else round(value_weak, 4),
                              "ms_per_step": round(dt / K * 1e3, 4), "scaling": scaling, "config": {"workload": workload, "parallelism": par}, "headline_only": True}), flush=True)
        if pipe is not None:
            pipe.close()
        for c in ctxs:
            c.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---- the other measurements of the same run (bench_detail.py): all ranks take part
    if not args.no_extras:
        bench_detail.sharded(env, detail, run_steps)

    # ---- per-kernel times, live, HIP events on the library's stream (rank 0 only; the other ranks wait at the barrier below)
    if rank == 0:
        reps = max(3, min(K, 16))
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.clock_warmup_ms:  # the kernel times below are quoted at the sustained clock, like the headline
            t.gen_grid_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
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
        del zc
        detail.update({"ms_noise_kernels": round(ms_gen, 4), "ms_grid_kernel": round(ms_grid, 4), "ms_minmax_unfused": round(ms_minmax, 4), "ms_erosion": round(ms_ero, 4)})
        if not args.no_extras:
            bench_detail.modes_and_dense(env, detail, ms_gen, ms_ero)
            bench_detail.fused_modes(env, detail)
            if world == 1:  # (rank 0 alone runs these: no collective with another rank inside)
                bench_detail.end_to_end(env, detail)
                bench_detail.onegrid_rank_floor(env, detail, args.simulate_world)
    barrier()

    if rank == 0:
        ms_step = dt / K * 1e3
        # dominant kernel by time: the noise grid kernel (k_sine_grid / k_noise_grid).  It is fp32-VALU bound (45 flop per byte written), so the
        # roofline is the flop one: achieved = SURVEY 8(d)'s flops per cell x cells / kernel time, peak = the chip's fp32 vector peak.  The HBM side
        # (4 B written per cell) is reported beside it.
        ms_k = detail["ms_grid_kernel"]
        fl = flops_per_cell(mode, args.octaves)
        tflops = fl * cells / (ms_k * 1e-3) / 1e12
        hbm = 4.0 * cells / (ms_k * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, profiles/*_pmc_traffic.json), only for the configuration they were taken on
        traffic_source = None
        for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", fn)))["kernels"]["k_sine_grid"]
                if mode == 0 and N == 16384 and args.octaves == 8:
                    traffic = pm["hbm_bytes_per_launch"]
                    traffic_source = f"profiles/{fn} (2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over tools/prof_driver.py: a recorded constant of this configuration, not measured in this run)"
                    break
            except Exception:
                pass
        roof = {"bound": "valu", "kernel": "k_sine_grid (+table kernels)" if mode == 0 else f"k_noise_grid<{args.mode}>", "achieved": round(tflops, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tflops / FP32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source, "algorithmic_flops": fl * cells, "algorithmic_bytes": 4 * cells,
                "hbm_achieved_gbs": round(hbm, 2), "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": round(hbm / HBM_PEAK_GBS, 4),
                "nofma_peak_tops": VALU_NOFMA_TOPS, "nofma_frac": round(tflops / VALU_NOFMA_TOPS, 4), "operative_ceiling": "nofma_frac (bit parity forbids the fused multiply-add the fp32 peak counts)",
                "note": "fp32 VALU bound; the peak counts fused multiply-adds at 2.4 GHz.  Bit-parity with the FMA-free CPU reference forbids fusing (mul and add issue separately): nofma_frac is the "
                        "fraction of that rate; and the chip sustains ~1.94 GHz under this kernel (GRBM_GUI_ACTIVE / duration, profiles/r04_clock_ramp.txt): at the clock it gets, its ~6000 "
                        "VALU instructions per wave (5120 of them the sum) keep the vector ALUs issuing ~90 % of the time (profiles/r04_pmc_summary.txt); the f32 matrix instructions share that datapath "
                        "(profiles/r04_sine_matrix_pipe.txt)"}
        out = {"metric": "heightmap Gcells/sec (noise+erosion), 16384^2 grid", "value": round(value, 4), "unit": "Gcells/s", "n_gpus": world, "steps": K, "warmup": W,
               "value_strong": None if value_strong is None else round(value_strong, 4), "value_weak": None if value_weak is None else round(value_weak, 4),
               "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "mesh_gen_mode": mode, "octaves": args.octaves, "droplets": args.droplets, "grid": N, "pipelines_per_gpu": P if (args.workload in ("heightmap", "regions") and scaling == "weak") else 1, "handoff": args.handoff, "parallelism": par},
               "latency_ms_single": detail.get("single", {}).get("latency_ms_single", round(ms_step, 4) if P == 1 else None),
               "roofline": roof, "detail": dict(detail, erosion=rep, rccl=rccl_note,
                                                  clock_warmup={"ms": args.clock_warmup_ms, "untimed_steps_run": spin_log,
                                                                "why": "a cold MI355X reaches its sustained clock after ~30 ms of load and loses it again within ~5 ms of idling (profiles/r04_clock_ramp.txt); every timed region is preceded by untimed steps of the same kind for this long, with < 1 ms between them and the timed steps"})}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, mode)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    if pipe is not None:
        pipe.close()
    for c in ctxs:
        c.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
