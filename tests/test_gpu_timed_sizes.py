"""GPU parity at the remaining sizes bench.py / tools/bench_extra.py TIME but no test compared with the oracle yet:

  dense erosion   16384^2 with 10^6 droplets (bench.py `detail.dense_erosion`), every cell
  fBm modes       simplex / Perlin / domain warp at 16384^2 (bench.py `detail.modes`): 512 full rows and 512 full columns, in four bands across the grid
  voxel fBm       512 x 512 x 64, gen_mode 1 (simplex) and 2 (Perlin), whole field (bench_extra `C5_voxels_512x512x64_simplex`; src/voxels.cpp:327-339)
  tiles, simplex  the 64 x 64 tile batch in simplex mode, un-eroded (bench_extra `C4_tiles64x64_simplex_0iters`) and with 1000 droplets per tile, all 4096 tiles
  mesh_seed 0     the function-static generator of gen_rand_sine_table_entries that continues from (1, 1) (src/mesh_gen.cpp:213-216,238-239): a fresh context
                  against a fresh oracle PROCESS, two consecutive scene starts

The oracle side runs on the box's host cores (OpenMP inside orc.gen_grid* / orc.voxel_fill, one worker per tile for the tile batch)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import orclib
from orclib import assert_bit_equal
from test_gpu_at_size import host_threads, oracle_pool_map

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_dense_erosion_16384_million_droplets_equals_oracle(pkg, gpu, orc):
    """bench.py's `dense_erosion["16384x16384_1000000_droplets"]`: the headline grid (8 octaves) eroded by 10^6 droplets on the 32768-slot ring -- every cell and the step
    count against the oracle's serial droplet loop"""
    N, droplets = 16384, 1000000
    st = gpu.init_scene(pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
    orc.init(orclib.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
    a = gpu.alloc(N * N * 4)
    mn, _ = gpu.gen_grid_minmax_dev(a.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    gpu.apply_erosion_dev(a.ptr, N, N, mn, droplets, pkg.ERODE_MINZ_IS_MIN)
    rep = gpu.erosion_report().as_dict()
    z = a.download(np.float32, (N, N)); a.free()
    ref = orc.gen_grid(-N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 1)
    assert np.float32(mn) == ref.min()
    st_o = orc.apply_erosion_stats(ref, float(ref.min()), droplets)
    diff = z.view(np.uint32) != ref.view(np.uint32)
    assert not diff.any(), f"{int(diff.sum())} cells differ, first at {np.argwhere(diff)[:4].tolist()}"
    assert rep["steps"] == st_o[0].steps and rep["droplets"] == droplets and rep["rounds"] > 20, rep
    print("erosion report", rep)


@pytest.mark.parametrize("mode", [1, 2, 4])
def test_fbm_modes_16384_rows_and_columns_equal_oracle(pkg, gpu, orc, mode):
    """bench.py's `detail.modes` (16384^2, 8 octaves, fused min / max): four bands of 128 FULL rows and four bands of 128 FULL columns spread over the grid
    (first, one and two thirds in, last), every cell of them against the oracle's eval_index on the full grid's generator (orc.gen_grid_rect)"""
    N, B = 16384, 128
    st = gpu.init_scene(pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
    orc.init(orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
    a = gpu.alloc(N * N * 4)
    mn, mx = gpu.gen_grid_minmax_dev(a.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    z = a.download(np.float32, (N, N)); a.free()
    assert np.float32(mn) == z.min() and np.float32(mx) == z.max()
    for o in (0, N // 3 + 5, 2 * N // 3 - 77, N - B):
        rows = orc.gen_grid_rect(-N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 0, o, N, B)
        assert_bit_equal(rows, z[o:o + B], f"mode {mode} rows {o}..{o + B}")
        cols = orc.gen_grid_rect(-N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, o, 0, B, N)
        assert_bit_equal(cols, z[:, o:o + B], f"mode {mode} columns {o}..{o + B}")
    assert np.isfinite(z).all() and len(np.unique(z[::97, ::89])) > 10000


@pytest.mark.parametrize("gen_mode", [1, 2])
def test_voxel_fbm_field_512x512x64_equals_oracle(pkg, gpu, orc, gen_mode):
    """voxel_manager::create_procedural in its 3-D fBm modes (src/voxels.cpp:327-339: glm::simplex / glm::perlin of vec3, 5 octaves, + z*zscale, CLIP_TO_pm1) at
    the reference's own landscape size 512 x 512 x 64 with tools/bench_extra.py's parameters: all 16.8 M voxels"""
    dims = (512, 512, 64)
    lo, vsz, off = (-15.9, -15.9, -1.0), (0.0622, 0.0622, 0.0625), (0.0, 0.0, 0.0)
    gpu.init_scene(pkg.make_config(mesh_gen_mode=0)); orc.init(orclib.make_config(mesh_gen_mode=0))
    ref = orc.voxel_fill(dims[0], dims[1], dims[2], lo, vsz, off, 1.0, 1.0, 123, 456, gen_mode, 0.01, 1)
    a = gpu.alloc(dims[0] * dims[1] * dims[2] * 4)
    gpu.voxel_fill_dev(a.ptr, dims[0], dims[1], dims[2], lo, vsz, off, 1.0, 1.0, 123, 456, gen_mode, 0.01, 1)
    v = a.download(np.float32, ref.shape); a.free()
    diff = v.view(np.uint32) != ref.view(np.uint32)
    assert not diff.any(), f"gen_mode {gen_mode}: {int(diff.sum())} voxels differ, first at {np.argwhere(diff)[:3].tolist()}"
    assert np.abs(ref).max() <= 1.0 and len(np.unique(ref[::7, ::5, ::3])) > 1000


@pytest.mark.parametrize("iters", [0, 1000])
def test_tile_batch_64x64_simplex_every_tile_equals_oracle(pkg, gpu, orc, iters):
    """BASELINE config 4 in simplex mode (k_noise_tiles, bench_extra `C4_tiles64x64_simplex_0iters`) and the same batch eroded by 1000 droplets per tile: zvals, stats
    bytes, normal texels and min_normal_z of ALL 4096 tiles against tile_t::create_zvals of the oracle"""
    tiles = [(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)]
    n = len(tiles)
    gpu.init_scene(pkg.make_config(mesh_gen_mode=1)); orc.init(orclib.make_config(mesh_gen_mode=1))
    try:
        z, st, nm, mnz = gpu.tiles_create_zvals(tiles, iters)

        def check(i):
            tx, ty = tiles[i]
            zo, so = orc.tile_create_zvals(tx, ty, iters)
            bad = []
            if not (zo.view(np.uint32) == z[i].view(np.uint32)).all():
                bad.append("zvals")
            if bytes(so) != bytes(st[i]):
                bad.append("stats")
            no, mo = orc.tile_normals(zo)
            if not ((no == nm[i]).all() and np.float32(mo) == mnz[i]):
                bad.append("normals")
            return bad

        res = oracle_pool_map(orc, check, range(n))
        failures = [(tiles[i], r) for i, r in enumerate(res) if r]
        assert not failures, f"{len(failures)} of {n} tiles differ: {failures[:5]}"
        if iters:
            z0, _, _, _ = gpu.tiles_create_zvals(tiles, 0, stats=False, normals=False)
            changed = (z0.view(np.uint32) != z.view(np.uint32)).reshape(n, -1).any(1)
            assert 100 < changed.sum() <= n, int(changed.sum())
    finally:
        orc.set_num_threads(host_threads())


_SEED0_ORACLE = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import orclib
orc = orclib.Checker(sys.argv[2])
out = []
for mode in (0, 0, 1, 0):   # the static generator advances with every sine-mode scene start and is re-seeded (1, 12345) by a non-sine one
    s = orc.init(orclib.make_config(mesh_gen_mode=mode, mesh_seed=0))
    g = orc.gen_grid(-40, 13, s.DX_VAL, s.DY_VAL, 96, 50, 1)
    out.append({"sinTable": np.ctypeslib.as_array(s.sinTable).reshape(-1).view(np.uint32).tolist(), "rx": float(s.rx), "ry": float(s.ry), "zmax_est": float(s.zmax_est),
                "grid": g.reshape(-1).view(np.uint32).tolist()})
print(json.dumps(out))
"""


def seed0_oracle_sequence(kind="orc"):
    """the oracle's answers from a FRESH process (its function-static generator starts at (1, 1) exactly once per process, like the reference's);
    kind = "ref": the reference's own mesh_gen.cpp (oracle/_ref), used by tests/test_oracle.py to pin the restatement"""
    r = subprocess.run([sys.executable, "-c", _SEED0_ORACLE, HERE, kind], check=True, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    return json.loads(r.stdout.strip().splitlines()[-1])


def check_seed0_sequence(pkg, t):
    """mesh_seed = 0 (src/mesh_gen.cpp:213-216): sine mode keeps drawing from the function-static generator (first scene start of a process: from (1, 1); the next one
    continues where the first stopped), any other mode re-seeds it with (mesh_rgen_index + 1, 12345).  `t` must be a fresh context."""
    want = seed0_oracle_sequence()
    tables = []
    for k, mode in enumerate((0, 0, 1, 0)):
        st = t.init_scene(pkg.make_config(mesh_gen_mode=mode, mesh_seed=0))
        tab = np.ctypeslib.as_array(st.sinTable).reshape(-1).view(np.uint32)
        tables.append(tab.copy())
        assert (tab == np.array(want[k]["sinTable"], np.uint32)).all(), f"scene start {k} (mode {mode}): sinTable differs"
        assert (np.float32(st.rx), np.float32(st.ry), np.float32(st.zmax_est)) == (np.float32(want[k]["rx"]), np.float32(want[k]["ry"]), np.float32(want[k]["zmax_est"])), k
        g = t.gen_grid(-40, 13, st.DX_VAL, st.DY_VAL, 96, 50, pkg.GEN_GLACIATE)
        assert (g.reshape(-1).view(np.uint32) == np.array(want[k]["grid"], np.uint32)).all(), f"scene start {k} (mode {mode}): grid differs"
    assert not (tables[0] == tables[1]).all(), "the second sine-mode start must continue the generator, not restart it"
    assert not (tables[3] == tables[0]).all() and not (tables[3] == tables[1]).all()


def test_mesh_seed_zero_static_generator_continues(pkg):
    t = pkg.Terra(0)
    try:
        check_seed0_sequence(pkg, t)
    finally:
        t.close()
