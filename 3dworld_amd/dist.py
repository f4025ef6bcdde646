"""Multi-GPU piece of the tile path: mesh shadows over tile strips owned by different ranks (one process per GPU, torch.distributed).

Everything else in the tile path is embarrassingly parallel (ranks own disjoint tiles, no communication); the shadow sweeps are the one
place where data crosses a tile border -- the outgoing edge heights (sh_out, 130 floats per tile side) of a tile are the starting
heights (sh_in) of its neighbour away from the light (tile_t::calc_shadows_for_light, src/tiled_mesh.cpp:664-692).  With the tile
columns dealt out in contiguous strips, a strip needs the sh_out_y column of the strip toward the light: a point-to-point send/recv of
(rows x 130) floats per strip border -- on MI355X nodes that is one xGMI hop between neighbouring ranks (backend "nccl" = RCCL), on the
CPU test it is gloo.  The strips form a pipeline in the light direction; this helper keeps it simple (a strip waits for the whole
border column of its neighbour)."""
import numpy as np


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
