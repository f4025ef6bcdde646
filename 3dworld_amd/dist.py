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
