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



def _free_port():
    """a TCP port nobody is listening on right now, from the kernel (a port computed from the pid lies in the ephemeral range: one whole-suite run in ~10 found it taken -- EADDRINUSE)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def partition(n_units, world, rank):
    """contiguous block partition used by bench.py's tiles workload (3dworld_amd/dist.py: partition_tiles)"""
    dmod = importlib.import_module("3dworld_amd.dist")
    return dmod.partition_tiles(list(range(n_units)), rank, world)


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
    mine = importlib.import_module("3dworld_amd.dist").partition_tiles(tiles, rank, world)
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
    port = _free_port()
    mp.spawn(_worker, args=(2, port, emul_lib, str(tmp_path)), nprocs=2, join=True)
    z = np.concatenate([np.load(tmp_path / f"z_{r}.npy") for r in range(2)])
    tiles = [(tx, ty) for ty in range(-2, 2) for tx in range(-3, 2)]
    assert len(z) == len(tiles)
    orc.init(orclib.make_config(mesh_gen_mode=0))
    for i in (0, 7, 10, 19):
        zo, _ = orc.tile_create_zvals(*tiles[i], 40)
        orclib.assert_bit_equal(zo, z[i], f"tile {tiles[i]}")


def _strips_worker(rank, world, port, emul_lib, out_dir, mode, nx, ny):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("3dworld_amd")
    dmod = importlib.import_module("3dworld_amd.dist")
    t = pkg.Terra(0, emul_lib)
    st = t.init_scene(pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
    r0, r1 = dmod.strip_rows(ny, rank, world)
    buf = t.alloc(max(1, (r1 - r0) * nx * 4))
    q0, q1, mn, mx = dmod.sharded_heightmap_strips(t, dist, buf.ptr, -nx / 2, -ny / 2, st.DX_VAL, st.DY_VAL, nx, ny, pkg.GEN_GLACIATE)
    assert (q0, q1) == (r0, r1)
    np.save(os.path.join(out_dir, f"strip_{rank}.npy"), buf.download(np.float32, (r1 - r0, nx)))
    np.save(os.path.join(out_dir, f"mm_{rank}.npy"), np.array([mn, mx], np.float32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,nx,ny", [(0, 260, 131), (1, 70, 51)])
def test_two_ranks_one_heightmap_as_row_strips(emul_lib, orc, tmp_path, mode, nx, ny):
    """bench.py's strips workload: ONE heightmap, two ranks, each its row strip, min / max of the whole map by all_reduce of one float
    (gloo here, RCCL on the MI355X node).  The union equals the oracle's full grid bit for bit and every rank holds the global min / max."""
    import torch.multiprocessing as mp
    import orclib
    port = _free_port()
    mp.spawn(_strips_worker, args=(2, port, emul_lib, str(tmp_path), mode, nx, ny), nprocs=2, join=True)
    z = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(2)])
    s = orc.init(orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
    ref = orc.gen_grid(-nx / 2, -ny / 2, s.DX_VAL, s.DY_VAL, nx, ny, 1)
    orclib.assert_bit_equal(ref, z, "union of the row strips")
    for r in range(2):
        mm = np.load(tmp_path / f"mm_{r}.npy")
        assert mm[0] == ref.min() and mm[1] == ref.max()


@pytest.mark.gpu
def test_two_ranks_one_heightmap_as_row_strips_on_the_hip_library(orc, tmp_path):
    """the same through libterra_hip.so (both ranks on GPU 0 of the test box, gloo for the one-float all_reduce; under RCCL the tensor lives on the device):
    the product path of bench.py's strips workload, not the host emulation"""
    import torch.multiprocessing as mp
    import orclib
    mode, nx, ny = 0, 1030, 517
    port = _free_port()
    mp.spawn(_strips_worker, args=(2, port, None, str(tmp_path), mode, nx, ny), nprocs=2, join=True)
    z = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(2)])
    s = orc.init(orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
    ref = orc.gen_grid(-nx / 2, -ny / 2, s.DX_VAL, s.DY_VAL, nx, ny, 1)
    orclib.assert_bit_equal(ref, z, "union of the row strips (HIP)")
    for r in range(2):
        mm = np.load(tmp_path / f"mm_{r}.npy")
        assert mm[0] == ref.min() and mm[1] == ref.max()


def test_strip_rows_cover_the_grid_once():
    dmod = importlib.import_module("3dworld_amd.dist")
    for ny in (1, 5, 130, 16384):
        for world in (1, 2, 3, 8):
            rows = [y for r in range(world) for y in range(*dmod.strip_rows(ny, r, world))]
            assert rows == list(range(ny))


def _shadow_worker(rank, world, port, emul_lib, out_dir, light):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("3dworld_amd")
    dmod = importlib.import_module("3dworld_amd.dist")
    t = pkg.Terra(0, emul_lib)
    t.init_scene(pkg.make_config(mesh_gen_mode=0))
    tiles = [(tx, ty) for ty in range(-1, 2) for tx in range(-3, 3)]  # 6 columns x 3 rows: strips of 3 columns
    keep = {}
    def make_zvals(mine):
        z, _, _, _ = t.tiles_create_zvals(mine, 0, stats=False, normals=False)
        z = (z * np.float32(4.0)).astype(np.float32)
        keep["z"] = t.alloc(z.nbytes).upload(z)
        return keep["z"].ptr
    def alloc_smask(n):
        keep["sm"] = t.alloc(max(n, 1) * 130 * 130)
        return keep["sm"].ptr
    mine, sm_ptr = dmod.sharded_tile_mesh_shadows(t, dist, tiles, light, make_zvals, alloc_smask)
    np.save(os.path.join(out_dir, f"sm_{rank}.npy"), keep["sm"].download(np.uint8, (len(mine), 130, 130)))
    np.save(os.path.join(out_dir, f"tiles_{rank}.npy"), np.array(mine, np.int32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("light", [(0.7, 0.4, 0.3), (-0.8, 0.3, 0.25)])
def test_two_ranks_mesh_shadows_with_edge_exchange(emul_lib, orc, tmp_path, light):
    """the one real exchange step of the tile path: strips of tile columns on two ranks, the border sh_out_y arrays travel by send/recv (gloo here, RCCL on
    the MI355X node); the union equals the single-process result for lights from both x directions (pipeline runs either way)."""
    import torch.multiprocessing as mp
    import orclib
    port = _free_port()
    mp.spawn(_shadow_worker, args=(2, port, emul_lib, str(tmp_path), light), nprocs=2, join=True)
    tiles = [(tx, ty) for ty in range(-1, 2) for tx in range(-3, 3)]
    orc.init(orclib.make_config(mesh_gen_mode=0))
    z = np.stack([orc.tile_create_zvals(tx, ty, 0)[0] for tx, ty in tiles]) * np.float32(4.0)
    want = orc.tiles_mesh_shadows(tiles, z, light)
    seen = 0
    for r in range(2):
        sm = np.load(tmp_path / f"sm_{r}.npy"); mine = [tuple(v) for v in np.load(tmp_path / f"tiles_{r}.npy")]
        assert len(mine) == 9
        for k, tl in enumerate(mine):
            assert (sm[k] == want[tiles.index(tl)]).all(), f"rank {r} tile {tl}"
            seen += 1
    assert seen == len(tiles) and want.any()


# ---- bench.py's launch contract: `python bench.py --gpus N` starts N ranks itself and the line it prints says n_gpus = N
def _bench_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("terra_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_launch_command_and_world_check():
    import subprocess
    b = _bench_mod()
    cmd = b.launch_cmd(4, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    # a launcher that started a different number of ranks than --gpus asks for is refused (before any GPU work): the printed n_gpus is always the N asked for
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def _last_json_line(text):
    """the contract: stdout is exactly ONE line, a JSON object (nothing from RCCL or anyone else around it)"""
    import json
    lines = text.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must be one JSON line, got: " + text[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks come up (both on GPU 0 of the 1-GPU test box, so gloo carries the barrier / max-over-ranks
    instead of RCCL, which wants one device per rank).  At N > 1 the headline is the STRONG one -- ONE heightmap per step on both ranks together, erosion included, its strips
    hipMemCreate allocations exchanged as file descriptors and mapped on both ranks -- with the independent-regions aggregate beside it as value_weak"""
    import subprocess
    env = dict(os.environ, TERRA_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--size", "2048", "--no-extras", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _last_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "strong" and line["value"] > 0
    assert line["value"] == line["value_strong"] and line["value_weak"] > 0 and "ONE 2048x2048 heightmap per step on 2 GPU" in line["config"]["workload"]
    assert abs(line["value"] - 2048 * 2048 / (line["ms_per_step"] * 1e-3) / 1e9) < 1e-2 * line["value"]                      # one grid per step, not one per rank
    reg = line["detail"]["regions"]
    assert reg["scaling"] == "weak" and abs(reg["value_weak"] - 2 * 2048 * 2048 / (reg["ms_per_step"] * 1e-3) / 1e9) < 1e-2 * reg["value_weak"]
    # --workload regions keeps the weak line as the headline (a scaling sweep of that mode alone)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--size", "2048", "--no-extras", "--no-cpu-baseline", "--workload", "regions"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _last_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] == line["value_weak"] and line["value_strong"] is None


@pytest.mark.gpu
def test_bench_preflight_two_ranks_names_every_stage():
    """`bench.py --gpus 2 --preflight`: process group, a collective, the one-grid VMM mapping, a row read through the peer mapping, an erosion across both strips -- each a
    named stage in one JSON line (both ranks on GPU 0 here; on an 8-GPU node the same command is the first thing to run)"""
    import subprocess
    env = dict(os.environ, TERRA_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _last_json_line(r.stdout)
    assert line["preflight"] == "ok" and line["n_gpus"] == 2 and line["failed_stage"] is None
    assert [s["stage"] for s in line["stages"]] == ["context", "collective", "onegrid_vmm_mapping", "strip_fill_and_peer_read", "erode_across_strips", "sharded_traces"] and all(s["ok"] for s in line["stages"])


@pytest.mark.gpu
def test_bench_world1_runs_its_collectives_through_rccl():
    """N = 1 on the GPU box: bench.py creates a one-rank RCCL group, so the device-tensor all_reduce / barrier branches of the sharded paths execute through RCCL"""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TERRA_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--size", "2048", "--no-cpu-baseline", "--workload", "strips"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _last_json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["detail"]["rccl"]["world1_group"] == "ok", line["detail"].get("rccl")
    assert "RCCL" in line["config"]["parallelism"]


def _nccl_world1_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = importlib.import_module("3dworld_amd")
    dmod = importlib.import_module("3dworld_amd.dist")
    t = pkg.Terra(0)
    st = t.init_scene(pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
    nx, ny = 1030, 517
    buf = t.alloc(nx * ny * 4)
    r0, r1, mn, mx = dmod.sharded_heightmap_strips(t, dist, buf.ptr, -nx / 2, -ny / 2, st.DX_VAL, st.DY_VAL, nx, ny, pkg.GEN_GLACIATE)  # all_reduce(min / max) on device tensors over RCCL
    np.save(os.path.join(out_dir, "z.npy"), buf.download(np.float32, (ny, nx)))
    np.save(os.path.join(out_dir, "mm.npy"), np.array([mn, mx, r0, r1], np.float64))
    # the shadow pass of a one-strip terrain through the same helper (no neighbour strip: no send / recv, but the device staging branch is the RCCL one)
    tiles = [(tx, ty) for ty in range(0, 2) for tx in range(0, 3)]
    keep = {}
    def make_zvals(mine):
        z, _, _, _ = t.tiles_create_zvals(mine, 0, stats=False, normals=False)
        keep["z"] = t.alloc(z.nbytes).upload(z)
        return keep["z"].ptr
    def alloc_smask(n):
        keep["sm"] = t.alloc(n * 130 * 130)
        return keep["sm"].ptr
    mine, _ = dmod.sharded_tile_mesh_shadows(t, dist, tiles, (0.7, 0.4, 0.3), make_zvals, alloc_smask)
    np.save(os.path.join(out_dir, "sm.npy"), keep["sm"].download(np.uint8, (len(mine), 130, 130)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_one_rank_group_runs_the_device_collectives(orc, tmp_path):
    """the `nccl` (= RCCL) branches of 3dworld_amd/dist.py on the 1-GPU box: a one-rank group, device tensors through all_reduce, results equal to the oracle's"""
    import torch.multiprocessing as mp
    import orclib
    port = _free_port()
    mp.spawn(_nccl_world1_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    nx, ny = 1030, 517
    s = orc.init(orclib.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
    ref = orc.gen_grid(-nx / 2, -ny / 2, s.DX_VAL, s.DY_VAL, nx, ny, 1)
    orclib.assert_bit_equal(ref, np.load(tmp_path / "z.npy"), "strip through the RCCL group")
    mm = np.load(tmp_path / "mm.npy")
    assert np.float32(mm[0]) == ref.min() and np.float32(mm[1]) == ref.max() and (mm[2], mm[3]) == (0, ny)
    tiles = [(tx, ty) for ty in range(0, 2) for tx in range(0, 3)]  # (the worker's scene: mesh_freq_filter = 1, as the oracle still has it)
    z = np.stack([orc.tile_create_zvals(tx, ty, 0)[0] for tx, ty in tiles])
    assert (orc.tiles_mesh_shadows(tiles, z, (0.7, 0.4, 0.3)) == np.load(tmp_path / "sm.npy")).all()


# ---------------------------------------------------------------- ONE heightmap on several ranks, erosion included (terra_dgrid + OneHeightmapPipeline)

def _dgrid_download(t, ptr, shape):
    out = np.empty(shape, np.float32)
    t._ck(t.lib.terra_memcpy_d2h(t.ctx, out.ctypes.data, ptr, out.nbytes))
    return out


def _one_grid_worker(rank, world, port, lib, out_dir, nx, ny, droplets, steps, grids, shard=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if shard:  # the sparse scheduler forced on (these maps are small) with an allowance that lets it resolve every conflict itself, droplets of both strips among them
        os.environ["TERRA_ONEGRID_SHARD_TRACES"] = "1"; os.environ["TERRA_ERO_SPARSE"] = "1"; os.environ["TERRA_ERO_SPARSE_RETRACES"] = "100000"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("3dworld_amd")
    dmod = importlib.import_module("3dworld_amd.dist")
    pipe = dmod.OneHeightmapPipeline(pkg, lambda: pkg.Terra(0, lib), pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1), dist, nx, ny, droplets, tag=f"t{port}", grids=grids, eroders=2)
    assert pipe.rows[0][0] == 0 and pipe.rows[-1][1] == ny and all(a[1] == b[0] for a, b in zip(pipe.rows, pipe.rows[1:]))

    # collect() is called from the pipeline's eroder threads (two here), each with its own context: the copies go through a context of their own behind a lock -- a context
    # is not for two threads at once (through ectx[0], one thread's copy could land in the stream the other was capturing its first erosion graph on: a rare
    # `capturing stream has unjoined work`)
    import threading
    dl, dl_lock = pkg.Terra(0, lib), threading.Lock()

    def collect(s, ptr):  # on the rank that eroded step s: the WHOLE grid through the mapped pointer (the other rank's rows included)
        with dl_lock:
            a = _dgrid_download(dl, ptr, (ny, nx))
        np.save(os.path.join(out_dir, f"grid_{s}.npy"), a)

    pipe.run(steps, origin=lambda s: (-nx / 2 + 40.0 * s, -ny / 2 - 25.0 * s), collect=collect)
    dist.barrier()
    # every rank sees the same final contents of the last grids through its own mapping (rank 1 reads rows rank 0 owns and the other way round)
    for g in range(min(grids, steps)):
        s_last = max(s for s in range(steps) if s % grids == g)
        np.save(os.path.join(out_dir, f"view_{rank}_{s_last}.npy"), _dgrid_download(pipe.nctx, pipe.grids[g].ptr, (ny, nx)))
    pipe.close()
    dist.destroy_process_group()


def _check_one_grid(orc, out_dir, nx, ny, droplets, steps, grids, world):
    import orclib
    s_ = orc.init(orclib.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
    for s in range(steps):
        ref = orc.gen_grid(-nx / 2 + 40.0 * s, -ny / 2 - 25.0 * s, s_.DX_VAL, s_.DY_VAL, nx, ny, 1)
        orc.apply_erosion(ref, float(ref.min()), droplets)
        orclib.assert_bit_equal(ref, np.load(os.path.join(out_dir, f"grid_{s}.npy")), f"step {s}: the eroded grid as its eroder sees it")
        if s >= steps - grids:
            for r in range(world):
                f = os.path.join(out_dir, f"view_{r}_{s}.npy")
                if os.path.exists(f):
                    orclib.assert_bit_equal(ref, np.load(f), f"step {s}: rank {r}'s view of the grid")


def test_two_ranks_erode_one_heightmap_whose_strips_live_on_both(emul_lib, orc, tmp_path):
    """SURVEY 8e row 3, the exact form: ONE heightmap per step on two ranks -- each evaluates its row strip into its own memory, min(vals) by all_reduce, the step's
    eroder runs apply_erosion over the mapped grid (the other rank's rows included), eroders alternate, three grids in flight.  Here the strips are memfds mapped into
    both emulator processes; on the GPU box they are hipMemCreate allocations (tests/test_distributed.py::test_two_ranks_erode_one_heightmap_on_the_hip_library)."""
    import torch.multiprocessing as mp
    nx, ny, droplets, steps, grids = 256, 100, 300, 7, 3
    port = _free_port()
    mp.spawn(_one_grid_worker, args=(2, port, emul_lib, str(tmp_path), nx, ny, droplets, steps, grids), nprocs=2, join=True)
    _check_one_grid(orc, str(tmp_path), nx, ny, droplets, steps, grids, 2)


def test_two_ranks_erode_one_heightmap_with_the_traces_made_by_the_strip_owners(emul_lib, orc, tmp_path):
    """the same pipeline with terra_erosion_shard_*: after a step's all_reduce each rank traces the droplets that start in ITS rows into its own arena (a second
    terra_dgrid per grid in flight), a second collective, and the step's eroder gathers the traces through the mapping and checks / commits -- the oracle's grids"""
    import torch.multiprocessing as mp
    nx, ny, droplets, steps, grids = 256, 160, 150, 5, 3
    port = _free_port()
    mp.spawn(_one_grid_worker, args=(2, port, emul_lib, str(tmp_path), nx, ny, droplets, steps, grids, True), nprocs=2, join=True)
    _check_one_grid(orc, str(tmp_path), nx, ny, droplets, steps, grids, 2)


@pytest.mark.gpu
def test_two_ranks_shard_the_traces_on_the_hip_library(orc, tmp_path):
    """... through libterra_hip.so: two processes on GPU 0, the arenas hipMemCreate allocations mapped by the peer; the eroding process reads the other one's traces"""
    import torch.multiprocessing as mp
    nx, ny, droplets, steps, grids = 2048, 1024, 400, 6, 3
    port = _free_port()
    mp.spawn(_one_grid_worker, args=(2, port, None, str(tmp_path), nx, ny, droplets, steps, grids, True), nprocs=2, join=True)
    _check_one_grid(orc, str(tmp_path), nx, ny, droplets, steps, grids, 2)


@pytest.mark.gpu
def test_two_ranks_erode_one_heightmap_on_the_hip_library(orc, tmp_path):
    """the same through libterra_hip.so: two processes on GPU 0, every strip a hipMemCreate allocation exported as a file descriptor, imported and mapped by the peer
    (hipMemImportFromShareableHandle / hipMemMap / hipMemSetAccess) -- the erosion kernels run over memory that belongs to another process"""
    import torch.multiprocessing as mp
    nx, ny, droplets, steps, grids = 2048, 1024, 1000, 6, 3
    port = _free_port()
    mp.spawn(_one_grid_worker, args=(2, port, None, str(tmp_path), nx, ny, droplets, steps, grids), nprocs=2, join=True)
    _check_one_grid(orc, str(tmp_path), nx, ny, droplets, steps, grids, 2)


def _one_grid_device_paced_worker(rank, world, out_dir, nx, ny, droplets, steps, grids, shard=False):
    import torch
    import torch.distributed as dist
    if shard:
        os.environ["TERRA_ONEGRID_SHARD_TRACES"] = "1"
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", store=dist.HashStore(), rank=rank, world_size=world)
    pkg = importlib.import_module("3dworld_amd")
    dmod = importlib.import_module("3dworld_amd.dist")
    pipe = dmod.OneHeightmapPipeline(pkg, lambda: pkg.Terra(0), pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1), dist, nx, ny, droplets, tag=f"d{os.getpid()}", grids=grids, eroders=2,
                                     coll_device=torch.device("cuda:0"))
    assert pipe._dev_paced

    import threading
    dl, dl_lock = pkg.Terra(0), threading.Lock()  # (see _one_grid_worker: the eroder threads' copies through a context of their own)

    def collect(s, ptr):
        with dl_lock:
            a = _dgrid_download(dl, ptr, (ny, nx))
        np.save(os.path.join(out_dir, f"grid_{s}.npy"), a)

    origin = lambda s: (-nx / 2 + 40.0 * s, -ny / 2 - 25.0 * s)  # noqa: E731
    pipe.run(steps, origin=origin, collect=collect)
    pipe.run(2, origin=origin)  # (run() may be called again at once)
    pipe.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_one_rank_device_paced_pipeline_over_rccl_equals_oracle(orc, tmp_path):
    """OneHeightmapPipeline with a collective that runs on the device (RCCL, a one-rank group on the one-GPU box): the steps are only enqueued -- strip min in HBM,
    all_reduce(min) on the noise stream, the erosion's clamp reads it from HBM -- and every step's eroded grid equals the oracle's"""
    import torch.multiprocessing as mp
    nx, ny, droplets, steps, grids = 2048, 1024, 1000, 7, 3
    mp.spawn(_one_grid_device_paced_worker, args=(1, str(tmp_path), nx, ny, droplets, steps, grids), nprocs=1, join=True)
    _check_one_grid(orc, str(tmp_path), nx, ny, droplets, steps, grids, 0)


@pytest.mark.gpu
def test_one_rank_device_paced_pipeline_with_sharded_traces_equals_oracle(orc, tmp_path):
    """the enqueue-only pipeline with the trace / finish split (one rank: its strip is the whole grid, the tracer context's stream and events are the real ones)"""
    import torch.multiprocessing as mp
    nx, ny, droplets, steps, grids = 2048, 1024, 300, 7, 3
    mp.spawn(_one_grid_device_paced_worker, args=(1, str(tmp_path), nx, ny, droplets, steps, grids, True), nprocs=1, join=True)
    _check_one_grid(orc, str(tmp_path), nx, ny, droplets, steps, grids, 0)


@pytest.mark.gpu
def test_in_process_distributed_grid_two_contexts(pkg, orc):
    """terra_multi_dgrid_create: two contexts of one process, strip i allocated by context i, one pointer for both -- context 0 fills its rows, context 1 its rows,
    context 1 erodes the whole grid"""
    import orclib
    nx, ny, droplets = 2048, 1024, 2000
    m = pkg.TerraMulti([0, 0])
    try:
        st = m.init_scene(pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
        h, ptr = m.dgrid_create([nx * (ny // 2) * 4] * 2)
        mins = []
        for i in range(2):
            r0 = i * (ny // 2)
            mn, _ = m.ctxs[i].gen_grid_rows_minmax_dev(ptr + r0 * nx * 4, -nx / 2, -ny / 2, st.DX_VAL, st.DY_VAL, nx, ny, r0, ny // 2, pkg.GEN_GLACIATE)
            mins.append(mn)
        m.synchronize()
        m.ctxs[1].apply_erosion_dev(ptr, nx, ny, min(mins), droplets, pkg.ERODE_MINZ_IS_MIN)
        z = _dgrid_download(m.ctxs[1], ptr, (ny, nx))
        m.dgrid_destroy(h)
        s_ = orc.init(orclib.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
        ref = orc.gen_grid(-nx / 2, -ny / 2, s_.DX_VAL, s_.DY_VAL, nx, ny, 1)
        assert np.float32(min(mins)) == ref.min()
        orc.apply_erosion(ref, float(ref.min()), droplets)
        orclib.assert_bit_equal(ref, z, "one grid over two contexts")
    finally:
        m.close()
