"""Multi-GPU path on CPU: 2 ranks (gloo, 127.0.0.1).  The data path has NO collective (tiles / regions are independent units, exactly like the
reference erodes each tile alone); ranks only need a deterministic partition, a barrier and a max-reduce of the step time.  Each rank computes its
share through the host-emulation library and rank 0 checks the union against the oracle."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def partition(n_units, world, rank):
    """contiguous block partition used by bench.py --workload tiles"""
    per = (n_units + world - 1) // world
    return range(min(rank * per, n_units), min((rank + 1) * per, n_units))


def _worker(rank, world, port, emul_lib, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("3dworld_amd")
    t = pkg.Terra(0, emul_lib)
    t.init_scene(pkg.make_config(mesh_gen_mode=0))
    tiles = [(tx, ty) for ty in range(-2, 2) for tx in range(-3, 2)]  # 20 tiles
    mine = [tiles[i] for i in partition(len(tiles), world, rank)]
    z, st, nm, mnz = t.tiles_create_zvals(mine, 40)
    np.save(os.path.join(out_dir, f"z_{rank}.npy"), z)
    dist.barrier()
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)  # the only reduction bench.py performs (max step time over ranks)
    assert tt.item() == world
    dist.destroy_process_group()


def test_partition_covers_all_units_once():
    for n in (0, 1, 7, 4096):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in partition(n, world, r)]
            assert got == list(range(n))


def test_two_ranks_tile_sharding_matches_oracle(emul_lib, orc, tmp_path):
    import torch.multiprocessing as mp
    import orclib
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, emul_lib, str(tmp_path)), nprocs=2, join=True)
    z = np.concatenate([np.load(tmp_path / f"z_{r}.npy") for r in range(2)])
    tiles = [(tx, ty) for ty in range(-2, 2) for tx in range(-3, 2)]
    assert len(z) == len(tiles)
    orc.init(orclib.make_config(mesh_gen_mode=0))
    for i in (0, 7, 10, 19):
        zo, _ = orc.tile_create_zvals(*tiles[i], 40)
        orclib.assert_bit_equal(zo, z[i], f"tile {tiles[i]}")
