"""Multi-GPU piece of the tile path: mesh shadows over tile strips owned by different ranks (one process per GPU, torch.distributed).

Everything else in the tile path is embarrassingly parallel (ranks own disjoint tiles, no communication); the shadow sweeps are the one
place where data crosses a tile border -- the outgoing edge heights (sh_out, 130 floats per tile side) of a tile are the starting
heights (sh_in) of its neighbour away from the light (tile_t::calc_shadows_for_light, src/tiled_mesh.cpp:664-692).  With the tile
columns dealt out in contiguous strips, a strip needs the sh_out_y column of the strip toward the light: a point-to-point send/recv of
(rows x 130) floats per strip border -- on MI355X nodes that is one xGMI hop between neighbouring ranks (backend "nccl" = RCCL), on the
CPU test it is gloo.  The strips form a pipeline in the light direction; this helper keeps it simple (a strip waits for the whole
border column of its neighbour)."""
import numpy as np


def partition_tiles(tiles, rank, world):
    """BASELINE config 4 on several GPUs: contiguous block partition of the tile list (tiles are independent units: tile_t::create_zvals needs nothing
    from a neighbour, the reference erodes each tile alone on its clamp-padded copy, src/tiled_mesh.cpp:515) -- no collective, rank r computes tiles[lo:hi]"""
    n = len(tiles)
    per = (n + world - 1) // world
    return list(tiles[min(rank * per, n):min((rank + 1) * per, n)])


def strip_rows(ny, rank, world):
    """rows [r0, r1) of an ny-row heightmap owned by `rank` when the grid is cut into `world` contiguous row strips (heightmap_t::proc_gen's loop is
    independent per row, src/heightmap.cpp:139-143)"""
    per = (ny + world - 1) // world
    return min(rank * per, ny), min((rank + 1) * per, ny)


def sharded_heightmap_strips(terra, dist, d_out_ptr, x0, y0, dx, dy, nx, ny, flags, min_start_sin=0):
    """ONE nx x ny heightmap on all ranks of `dist`: this rank evaluates its row strip into d_out_ptr (strip-local layout, (r1 - r0) x nx floats) with
    terra_gen_grid_rows_minmax_dev -- bit-identical to the same rows of a single-GPU grid -- and min / max of the WHOLE map (what heightmap_t::run_erosion
    and from_floats need next) come from one all_reduce each of a single float (backend "nccl" = RCCL over xGMI on the MI355X node, gloo in the CPU test).
    Returns (r0, r1, min, max).  Erosion is not part of this: one shared grid in serial droplet order does not shard (replicas only)."""
    import torch
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
    r0, r1 = strip_rows(ny, rank, world)
    mn, mx = float("inf"), float("-inf")
    if r1 > r0:
        mn, mx = terra.gen_grid_rows_minmax_dev(d_out_ptr, x0, y0, dx, dy, nx, ny, r0, r1 - r0, flags, min_start_sin)
    if dist is not None and dist.is_initialized():  # also in a one-rank group: the collective then runs through the backend (RCCL on a 1-GPU box) instead of being skipped
        on_gpu = str(dist.get_backend()).lower() == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        a = torch.tensor([mn], dtype=torch.float32, device=dev); b = torch.tensor([mx], dtype=torch.float32, device=dev)
        dist.all_reduce(a, op=dist.ReduceOp.MIN); dist.all_reduce(b, op=dist.ReduceOp.MAX)
        mn, mx = float(a.item()), float(b.item())
    return r0, r1, mn, mx


def strip_of(tile_x, x_min, x_max, world):
    """rank owning tile column tile_x when columns [x_min, x_max] are dealt out in `world` contiguous strips"""
    width = x_max - x_min + 1
    per = -(-width // world)
    return min((tile_x - x_min) // per, world - 1)


def sharded_tile_mesh_shadows(terra, dist, tiles, light_pos, make_zvals, alloc_smask):
    """terra: a Terra context on this rank's device.  tiles: the FULL tile list (same on every rank).  make_zvals(my_tiles) -> device pointer of
    their zvals; alloc_smask(n) -> device pointer for n*130*130 bytes.  Returns (my_tiles, smask_ptr).  Communication: torch.distributed send/recv
    (the edge arrays are tiny: 520 B per tile border); under the "nccl" (= RCCL) backend the message buffers are staged on this rank's GPU, because
    RCCL moves device memory only, under gloo they stay on the host."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = str(dist.get_backend()).lower() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    tiles = [tuple(int(v) for v in t) for t in tiles]
    xs = [t[0] for t in tiles]
    x_min, x_max = min(xs), max(xs)
    owner = {t: strip_of(t[0], x_min, x_max, world) for t in tiles}
    mine = [t for t in tiles if owner[t] == rank]
    sx = -1 if light_pos[0] < 0.0 else 1  # toward the light in x
    # tiles of mine whose x-neighbour toward the light exists but belongs to another rank: their sh_in_y arrives from that rank
    need = [(i, (t[0] + sx, t[1])) for i, t in enumerate(mine) if (t[0] + sx, t[1]) in owner and owner[(t[0] + sx, t[1])] != rank]
    edge_in = np.full((len(mine), 2, 130), -1.0e6, np.float32)
    present = np.zeros((len(mine), 2), np.uint8)
    src_ranks = sorted({owner[nb] for _, nb in need})
    for src in src_ranks:  # one message per neighbouring strip: the sh_out_y rows of its border tiles, in the order of `need`
        rows = [(i, nb) for i, nb in need if owner[nb] == src]
        buf = torch.empty((len(rows), 130), dtype=torch.float32, device=dev)
        dist.recv(buf, src=src)
        host = buf.cpu().numpy()
        for k, (i, _) in enumerate(rows):
            edge_in[i, 1] = host[k]
            present[i, 1] = 1
    z_ptr = make_zvals(mine)
    sm_ptr = alloc_smask(len(mine))
    edge_out = terra.tiles_mesh_shadows_halo_dev(mine, z_ptr, light_pos, sm_ptr, edge_in if len(need) else None, present if len(need) else None, True) if mine else None
    # ship my border tiles' sh_out_y to the strips away from the light
    index = {t: i for i, t in enumerate(mine)}
    for dst in range(world):
        if dst == rank:
            continue
        dst_tiles = [t for t in tiles if owner[t] == dst]
        rows = [index[(t[0] + sx, t[1])] for t in dst_tiles if (t[0] + sx, t[1]) in index]  # same order as the receiver's `need`
        if rows:
            dist.send(torch.from_numpy(np.ascontiguousarray(edge_out[rows, 1])).to(dev), dst=dst)
    return mine, sm_ptr


# ---------------------------------------------------------------- ONE heightmap on several GPUs, erosion included (terra_dgrid)

def exchange_fds(rank, world, fd, tag, sockdir=None):
    """every rank hands the file descriptor of its strip to every other rank of the node over unix sockets (SCM_RIGHTS): returns {rank: fd} of the peers.
    `tag` names the rendezvous (unique per grid and job, e.g. f"{MASTER_PORT}_{k}"); `sockdir` is a directory only this user can enter (create_distributed_grid makes one
    per job with mkdtemp and tells the other ranks through the process group).  A rank starts listening, then connects to the others with retries, so no barrier is
    needed in between.  A connection from another user (SO_PEERCRED), a message without exactly one descriptor or with a rank out of range is refused; on any failure
    the descriptors received so far are closed."""
    import os
    import socket
    import struct
    import time
    base = sockdir or os.environ.get("TERRA_DGRID_SOCKDIR", "/tmp")
    def path(r):
        return os.path.join(base, f"terra_dgrid_{tag}_{r}.sock")
    got = {}
    if world == 1:
        return got
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        os.unlink(path(rank))
    except FileNotFoundError:
        pass
    srv.bind(path(rank))
    os.chmod(path(rank), 0o600)
    srv.listen(world)
    ok = False
    try:
        for peer in range(world):  # send mine to everybody else
            if peer == rank:
                continue
            deadline = time.time() + 120.0
            while True:
                c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    c.connect(path(peer))
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    c.close()
                    if time.time() > deadline:
                        raise TimeoutError(f"terra_dgrid: rank {peer} never opened its socket")
                    time.sleep(0.01)
            try:
                socket.send_fds(c, [rank.to_bytes(4, "little")], [fd])
            finally:
                c.close()
        srv.settimeout(120.0)
        while len(got) < world - 1:  # and take theirs
            conn, _ = srv.accept()
            try:
                uid = struct.unpack("3i", conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))[1]
                msg, fds, _, _ = socket.recv_fds(conn, 4, 4)
            finally:
                conn.close()
            peer = int.from_bytes(msg, "little") if len(msg) == 4 else -1
            if uid != os.getuid() or len(fds) != 1 or not (0 <= peer < world) or peer == rank or peer in got:
                for f in fds:
                    os.close(f)
                raise RuntimeError(f"terra_dgrid: refused a descriptor message (uid {uid}, {len(fds)} descriptors, rank {peer})")
            got[peer] = fds[0]
        ok = True
    finally:
        srv.close()
        try:
            os.unlink(path(rank))
        except FileNotFoundError:
            pass
        if not ok:
            for f in got.values():
                os.close(f)
            got.clear()
    return got


def strip_rows_aligned(terra_mod, terra, nx, ny, world):
    """row strips of an nx x ny float grid whose byte sizes are multiples of the mapping granularity (terra_dgrid_granularity: 2 MiB on MI355X): rows per strip rounded
    up to the next multiple that satisfies it; the last strips may be short or empty.  Returns [(r0, r1)] * world and the padded strip size in bytes."""
    import math
    gran = terra_mod.DistributedGrid.granularity(terra)
    row_bytes = nx * 4
    unit = gran // math.gcd(gran, row_bytes)           # rows per granule-aligned block
    per = -(-ny // world)
    per = -(-per // unit) * unit
    rows = [(min(r * per, ny), min((r + 1) * per, ny)) for r in range(world)]
    return rows, per * row_bytes


def _all_ok(dist, ok, coll_device="cpu"):
    """every rank learns whether EVERY rank succeeded (one all_reduce(min) of a flag): a failure is raised on all ranks together, never on one rank while the others wait"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return ok
    import torch
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=coll_device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item()) > 0.5


def create_distributed_grid(terra_mod, terra, dist, nx, ny, tag, coll_device="cpu", strip_bytes=None):
    """ONE nx x ny float grid over all ranks of `dist`: this rank's rows live in its HBM, everybody maps the whole grid (see include/terra.h, terra_dgrid_*).
    Returns (grid, rows) with grid.ptr the mapped device pointer and rows[r] = (r0, r1) of rank r.  Raises RuntimeError on EVERY rank when any rank failed (a runtime
    without virtual memory management, devices that cannot map each other ...): the two steps end in an agreement, so nobody is left waiting in a collective."""
    import os
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
    g, fd, err = None, -1, None
    try:  # step 1: the local strip and its descriptor
        if strip_bytes is None:
            rows, strip_bytes = strip_rows_aligned(terra_mod, terra, nx, ny, world)
        else:  # a plain byte array in equal strips (the trace arenas of a sharded erosion): nx / ny unused
            gran = terra_mod.DistributedGrid.granularity(terra)
            rows, strip_bytes = None, -(-int(strip_bytes) // gran) * gran
        g = terra_mod.DistributedGrid(terra, [strip_bytes] * world, rank)
        if world > 1:
            fd = g.export_fd()
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    if not _all_ok(dist, err is None, coll_device):
        if g is not None:
            g.destroy()
        raise RuntimeError(f"terra_dgrid: allocation / export failed on some rank ({err or 'another rank'})")
    sockdir = None
    if world > 1:  # a directory only this user can enter, made by rank 0 and announced through the process group
        import tempfile
        box, dir_err = [None], None
        try:
            if rank == 0 and "TERRA_DGRID_SOCKDIR" not in os.environ:
                box[0] = tempfile.mkdtemp(prefix="terra_dgrid_")
                if len(box[0].encode()) > 255:  # the announcement below carries 255 bytes: a longer path would send the other ranks to a directory that does not exist
                    dir_err = f"socket directory path longer than 255 bytes ({box[0]!r}): set TMPDIR or TERRA_DGRID_SOCKDIR to a shorter one"
        except OSError as e:
            dir_err = f"mkdtemp failed: {e!r}"  # (no silent fall-back to a world-writable directory)
        if not _all_ok(dist, dir_err is None, coll_device):
            os.close(fd)
            g.destroy()
            raise RuntimeError(f"terra_dgrid: no private socket directory ({dir_err or 'rank 0 could not make one'})")
        # rank 0's path to everybody: one all_reduce(sum) of its bytes (the other ranks contribute zeros) -- the same kind of collective, on the same device, as every
        # other one of this module (no pickling, no object collectives)
        import torch
        raw = (box[0] or "").encode()[:255]
        buf = torch.zeros(256, dtype=torch.int32, device=coll_device)
        if rank == 0:
            buf[:len(raw)] = torch.tensor(list(raw), dtype=torch.int32)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        path = bytes(int(v) for v in buf.tolist() if int(v) != 0).decode()
        sockdir = path or None
    peers = {}
    try:  # step 2: everybody's strips, mapped
        if world > 1:
            try:
                peers = exchange_fds(rank, world, fd, tag, sockdir)
            finally:
                os.close(fd)
            for r in sorted(peers):
                g.import_fd(r, peers[r])
        g.map()
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    finally:
        for pfd in peers.values():
            try:
                os.close(pfd)
            except OSError:
                pass
    all_ok = _all_ok(dist, err is None, coll_device)  # (also: every rank is done with the sockets)
    if sockdir is not None and rank == 0:
        try:
            os.rmdir(sockdir)
        except OSError:
            pass
    if not all_ok:
        g.destroy()
        raise RuntimeError(f"terra_dgrid: import / map failed on some rank ({err or 'another rank'})")
    return g, rows


class OneHeightmapPipeline:
    """STRONG scaling of the whole hot path on ONE grid: every step produces one nx x ny heightmap with heightmap_t::proc_gen semantics (src/heightmap.cpp:130-187: eval loop,
    min(vals), apply_erosion over the whole map in serial droplet order) on all ranks together.

      noise     rank r evaluates its row strip of the step's grid into its own HBM (terra_gen_grid_rows_minmax_dev: bit-identical to those rows of a single-GPU grid)
      min       min(vals) of the whole map = all_reduce(min) of one float (RCCL over xGMI under "nccl", gloo in the CPU test); the collective is also the step's barrier:
                when it returns, every strip of the step's grid is written
      erosion   the step's grid is a terra_dgrid (all strips mapped back to back on every rank): rank s % world runs terra_apply_erosion_dev over the mapped pointer on one of
                its eroder contexts while everybody goes on with the next steps' noise; droplets that start in another rank's rows reach them over xGMI.  Serial droplet
                order, the single-GPU kernels: the result is the single-GPU result bit for bit.
      reuse     `grids` grids are in flight; before a rank contributes to the all_reduce of step s it waits (host) until the erosion of step s - grids + 1 is complete if
                it was that step's eroder -- so when the all_reduce of step s returns anywhere, that erosion is complete everywhere and step s + 1 may overwrite its grid."""

    def __init__(self, terra_mod, make_ctx, cfg, dist, nx, ny, droplets, tag, grids=8, eroders=2, coll_device="cpu", shard_traces=None, tracers=3):
        import os
        import threading
        self.pkg, self.dist, self.nx, self.ny, self.droplets = terra_mod, dist, nx, ny, droplets
        # shard_traces (opt-in; TERRA_ONEGRID_SHARD_TRACES=1): the sparse erosion scheduler's read-only phases run where the rows live (terra_erosion_shard_*): after a
        # step's all_reduce every rank probes / traces the droplets that START in its strip into its own arena (one more terra_dgrid per grid in flight, the strips are the
        # arenas), a second collective says "all traces made", the eroding rank gathers the traces over xGMI and checks / commits.  Same grid, bit for bit.
        self.shard = (os.environ.get("TERRA_ONEGRID_SHARD_TRACES", "0") == "1") if shard_traces is None else bool(shard_traces)
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
        self.nctx = make_ctx()
        self.st = self.nctx.init_scene(cfg)
        self.ectx = [make_ctx() for _ in range(eroders)]
        for c in self.ectx:
            c.init_scene(cfg)
        self.G = grids
        self.grids, self.rows = [], None
        for g in range(grids):
            try:
                dg, rows = create_distributed_grid(terra_mod, self.nctx, dist, nx, ny, f"{tag}_{g}", coll_device)
            except Exception:
                for x in self.grids:
                    x.destroy()
                self.nctx.close()
                for c in self.ectx:
                    c.close()
                raise
            self.grids.append(dg)
            self.rows = rows
        self.coll_device = coll_device
        self._threading = threading
        self.tctxs, self.arenas, self._pg2, self._hmin = [], [], None, []
        if self.shard:
            try:
                # a strip's traces are a latency chain (its longest droplet: a few hundred dependent steps), longer than a strip's noise at N = 8: consecutive steps' traces
                # overlap on `tracers` contexts (own streams), as consecutive erosions do on the eroder contexts
                for _ in range(max(1, tracers)):
                    c = make_ctx()
                    self.tctxs.append(c)
                    c.init_scene(cfg)
                self.arena_stride = None
                for g in range(grids):
                    ag, _ = create_distributed_grid(terra_mod, self.nctx, dist, 0, 0, f"{tag}_arena_{g}", coll_device, strip_bytes=self.tctxs[0].erosion_shard_arena_bytes(droplets))
                    self.arenas.append(ag)
                    self.arena_stride = ag.strip_bytes[0]
                    self._hmin.append(self.tctxs[0].alloc(8))
                self.row_end = [r1 for (_, r1) in self.rows]
                if dist is not None and dist.is_initialized() and self.world > 1:
                    # the "traces made" collectives: a communicator of their own, so that they never queue in front of the next steps' all_reduce(min).  (A new communicator
                    # prints a banner on the C stdout: a bench line must stay alone there)
                    import sys
                    sys.stdout.flush()
                    saved = os.dup(1)
                    os.dup2(2, 1)
                    try:
                        self._pg2 = dist.new_group()
                        if str(coll_device) == "cpu":
                            self._traces_made(True)  # (gloo connects at the first collective)
                    finally:
                        import ctypes
                        try:
                            ctypes.CDLL(None).fflush(None)
                        finally:
                            os.dup2(saved, 1)
                            os.close(saved)
            except Exception:
                self._free_shard()
                for x in self.grids:
                    x.destroy()
                self.nctx.close()
                for c in self.ectx:
                    c.close()
                raise
        # A collective that runs on the device (RCCL) lets a step be enqueued without a host round trip -- the form tools/bench_native_onegrid.c has in C: the strip's
        # {min, max} stay in HBM (terra_gen_grid_rows_minmax_async_dev), all_reduce(min) works on that float on the noise context's stream, the eroding context's
        # stream waits for an event behind it and the final clamp reads min(vals) from HBM (terra_apply_erosion_devmin_dev).  Measured on one GPU at a simulated world
        # of 8: 0.15-0.16 ms per rank and step against 0.21-0.24 ms with the read-back (profiles/r05_onegrid_native.jsonl, bench.py detail.onegrid_rank_floor).
        self._dev_paced = str(coll_device).startswith("cuda") and os.environ.get("TERRA_ONEGRID_PACING", "device") != "host"  # (host: the read-back step, whatever the collective runs on)
        if self._dev_paced:
            import torch
            self._torch = torch
            self._tstream = torch.cuda.Stream(device=coll_device)  # the noise context works on it, the process group orders its collective against it
            self.nctx.set_stream(self._tstream.cuda_stream)
            self._mm = torch.zeros((grids, 2), dtype=torch.float32, device=coll_device)
            torch.cuda.current_stream(coll_device).synchronize()  # (the fill ran on the current stream; everything else touches _mm on _tstream)
            self._ev = [self.nctx.event_create() for _ in range(grids)]
            if self.shard:
                self._tstream2 = [torch.cuda.Stream(device=coll_device) for _ in self.tctxs]
                for c, st2 in zip(self.tctxs, self._tstream2):
                    c.set_stream(st2.cuda_stream)
                self._ev2 = [self.tctxs[0].event_create() for _ in range(grids)]
                self._flag = torch.zeros(grids, dtype=torch.float32, device=coll_device)
                torch.cuda.current_stream(coll_device).synchronize()

    def _free_shard(self):
        for a in self.arenas:
            a.destroy()
        self.arenas = []
        for b in self._hmin:
            b.free()
        self._hmin = []
        for c in self.tctxs:
            c.close()
        self.tctxs = []

    def _traces_made(self, ok=True):
        """host-side collective of the sharded form: every rank's traces of the step are complete (and whether all of them worked)"""
        if self.dist is None or not self.dist.is_initialized() or self.world == 1:
            return ok
        import torch
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=self.coll_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self._pg2)
        return float(t.item()) > 0.5

    def close(self):
        for c in [self.nctx] + self.ectx + self.tctxs:
            c.synchronize()
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()  # nobody unmaps a strip a peer may still be reading
        for g in self.grids:
            g.destroy()
        if self._dev_paced:
            for e in self._ev:
                self.nctx.event_destroy(e)
            self.nctx.set_stream(None)
            if self.shard:
                for e in self._ev2:
                    self.tctxs[0].event_destroy(e)
                for c in self.tctxs:
                    c.set_stream(None)
        if self.shard:
            self._free_shard()
        self.nctx.close()
        for c in self.ectx:
            c.close()

    def _all_reduce_min(self, v, ok=True):
        """min over the ranks of (v, ok flag): one collective per step carries the map's min(vals) AND whether any rank has seen an error, so that every rank
        leaves the step loop on the same step (a rank that raised alone would leave the others waiting in the next all_reduce)"""
        if self.dist is None or not self.dist.is_initialized():
            return v, ok
        import torch
        t = torch.tensor([v, 1.0 if ok else 0.0], dtype=torch.float32, device=self.coll_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        h = t.tolist()
        return float(h[0]), h[1] > 0.5

    def run(self, k, origin=None, collect=None):
        """k steps (see _run_host_paced for the arguments); with a collective that runs on the device the steps are only enqueued (_run_device_paced).
        collect(s, ptr) is called from the eroder THREADS (one per eroder context, possibly at the same time): it must not use the pipeline's contexts -- a context is not
        for two threads at once -- but one of its own (the grid is final and the eroder's stream drained when it is called)."""
        return self._run_device_paced(k, origin, collect) if self._dev_paced else self._run_host_paced(k, origin, collect)

    def _run_device_paced(self, k, origin=None, collect=None):
        """The same steps with nothing read back inside a step.  A failure on one rank cannot leave the others waiting: every rank enqueues all k collectives whatever
        happens to its own work, and the closing collective (host) carries the error flag -- all ranks raise together, after the loop."""
        import queue
        pkg, N, torch = self.pkg, self.nx, self._torch
        r0, r1 = self.rows[self.rank]
        group = self.dist is not None and self.dist.is_initialized()
        done, errs = {}, []
        jobs = [queue.Queue() for _ in self.ectx]

        def eroder(i):
            c = self.ectx[i]
            while True:
                job = jobs[i].get()
                if job is None:
                    return
                s, g = job
                try:
                    if self.shard:
                        c.event_wait(self._ev2[g])  # behind every rank's traces of the step (the second collective), on the device
                        c.erosion_shard_finish_dev(self.grids[g].ptr, self.nx, self.ny, self._mm[g].data_ptr(), self.droplets, pkg.ERODE_MINZ_IS_MIN, self.world, self.rank,
                                                   self.row_end, self.arenas[g].strip_ptr(self.rank), self.arena_stride)
                    else:
                        c.event_wait(self._ev[g])  # behind the step's noise and its all_reduce, on the device
                        c.apply_erosion_devmin_dev(self.grids[g].ptr, self.nx, self.ny, self._mm[g].data_ptr(), self.droplets, pkg.ERODE_MINZ_IS_MIN)
                    c.synchronize()
                    if collect is not None:
                        collect(s, self.grids[g].ptr)
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))
                finally:
                    done[s].set()
        th = [self._threading.Thread(target=eroder, args=(i,)) for i in range(len(self.ectx))]
        for x in th:
            x.start()
        mine = 0
        try:
            with torch.cuda.stream(self._tstream):
                for s in range(k):
                    g = s % self.G
                    x0, y0 = origin(s) if origin is not None else (-self.nx / 2, -self.ny / 2)
                    j = s - self.G + 1
                    if j >= 0 and j % self.world == self.rank and j in done:
                        done[j].wait()  # my erosion of the grid that step s + 1 overwrites is complete before all_reduce(s) can complete anywhere
                    try:
                        if r1 > r0 and not errs:
                            self.nctx.gen_grid_rows_minmax_async_dev(self.grids[g].ptr + r0 * N * 4, x0, y0, self.st.DX_VAL, self.st.DY_VAL, self.nx, self.ny, r0, r1 - r0,
                                                                     self._mm[g].data_ptr(), pkg.GEN_GLACIATE)
                        else:
                            self._mm[g, 0].fill_(float("inf"))
                    except Exception as e:  # noqa: BLE001
                        errs.append(repr(e))
                    if group:
                        self.dist.all_reduce(self._mm[g, 0:1], op=self.dist.ReduceOp.MIN)  # enqueued: ordered behind the strip's kernels and in front of the record below
                    self.nctx.event_record(self._ev[g])
                    if self.shard:  # this rank's traces behind the step's all_reduce, on a tracer context's stream; then "all traces made" on that stream, in its own group
                        tc = self.tctxs[s % len(self.tctxs)]
                        with torch.cuda.stream(self._tstream2[s % len(self.tctxs)]):
                            try:
                                tc.event_wait(self._ev[g])
                                if not errs:
                                    tc.erosion_shard_trace_dev(self.grids[g].ptr, self.nx, self.ny, self.droplets, r0, r1 - r0, self.arenas[g].strip_ptr(self.rank))
                            except Exception as e:  # noqa: BLE001
                                errs.append(repr(e))
                            if group and self._pg2 is not None:
                                self.dist.all_reduce(self._flag[g:g + 1], op=self.dist.ReduceOp.MIN, group=self._pg2)
                            tc.event_record(self._ev2[g])
                    if s % self.world == self.rank and not errs:
                        done[s] = self._threading.Event()
                        jobs[mine % len(jobs)].put((s, g))
                        mine += 1
        except Exception as e:  # noqa: BLE001 -- (a failing collective: nothing left to keep in step with)
            errs.append(repr(e))
        finally:
            for q in jobs:
                q.put(None)
            for x in th:
                x.join()
        try:
            self.nctx.synchronize()
            for c in self.tctxs:
                c.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
        _, ok = self._all_reduce_min(0.0, not errs)  # every rank's erosions are complete, and whether any of them failed
        if errs or not ok:
            raise RuntimeError("; ".join(errs) if errs else "a step failed on another rank")

    def _run_host_paced(self, k, origin=None, collect=None):
        """k steps.  origin(s) -> (x0, y0) of step s's grid (default: the nx x ny grid centred on the origin); collect(s, ptr) is called on the eroding rank when step s's
        grid is final (before it can be overwritten).  Returns when the erosions of ALL ranks are complete (one more collective after the loop): run() may be
        called again at once."""
        import queue
        pkg, N = self.pkg, self.nx
        r0, r1 = self.rows[self.rank]
        done = {}
        jobs = [queue.Queue() for _ in self.ectx]
        errs = []

        def eroder(i):
            c = self.ectx[i]
            while True:
                job = jobs[i].get()
                if job is None:
                    return
                s, g, mn = job
                try:
                    if self.shard:
                        import numpy as np
                        self._hmin[g].upload(np.array([mn, 0.0], np.float32))
                        c.erosion_shard_finish_dev(self.grids[g].ptr, self.nx, self.ny, self._hmin[g].ptr, self.droplets, pkg.ERODE_MINZ_IS_MIN, self.world, self.rank,
                                                   self.row_end, self.arenas[g].strip_ptr(self.rank), self.arena_stride)
                    else:
                        c.apply_erosion_dev(self.grids[g].ptr, self.nx, self.ny, mn, self.droplets, pkg.ERODE_MINZ_IS_MIN)
                    c.synchronize()
                    if collect is not None:
                        collect(s, self.grids[g].ptr)
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))
                finally:
                    done[s].set()
        th = [self._threading.Thread(target=eroder, args=(i,)) for i in range(len(self.ectx))]
        for x in th:
            x.start()
        mine = 0
        try:
            for s in range(k):
                g = s % self.G
                x0, y0 = origin(s) if origin is not None else (-self.nx / 2, -self.ny / 2)
                mn = float("inf")
                if r1 > r0:
                    try:
                        mn, _ = self.nctx.gen_grid_rows_minmax_dev(self.grids[g].ptr + r0 * N * 4, x0, y0, self.st.DX_VAL, self.st.DY_VAL, self.nx, self.ny, r0, r1 - r0, pkg.GEN_GLACIATE)
                    except Exception as e:  # noqa: BLE001 -- reported through the step's collective like an eroder's failure
                        errs.append(repr(e))
                j = s - self.G + 1
                if j >= 0 and j % self.world == self.rank:
                    done[j].wait()  # my erosion of the grid that step s + 1 overwrites is complete before I let all_reduce(s) complete anywhere
                mn, ok = self._all_reduce_min(mn, not errs)
                if not ok:  # on EVERY rank in this step
                    if not errs:
                        errs.append("an erosion failed on another rank")
                    break
                if self.shard:  # the grid is complete everywhere: my strip's droplets, then "all traces made"
                    try:
                        self.tctxs[0].erosion_shard_trace_dev(self.grids[g].ptr, self.nx, self.ny, self.droplets, r0, r1 - r0, self.arenas[g].strip_ptr(self.rank))
                        self.tctxs[0].synchronize()
                    except Exception as e:  # noqa: BLE001
                        errs.append(repr(e))
                    if not self._traces_made(not errs):
                        if not errs:
                            errs.append("a trace failed on another rank")
                        break
                if s % self.world == self.rank:
                    done[s] = self._threading.Event()
                    jobs[mine % len(jobs)].put((s, g, mn))
                    mine += 1
        finally:
            for q in jobs:
                q.put(None)
            for x in th:
                x.join()
        # the erosions of the last steps ran on their ranks after the loop's last collective: agree that they are all complete (a peer may still have been reading and
        # writing this rank's strips over xGMI -- the next run() writes noise into grid 0 at once) and that none of them failed
        _, ok = self._all_reduce_min(0.0, not errs)
        if errs or not ok:
            raise RuntimeError("; ".join(errs) if errs else "an erosion failed on another rank")
