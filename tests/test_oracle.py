import os
"""CPU tests of the oracle (oracle/terra_oracle.c): against the reference's own translation units when they can be built
here (oracle/_ref), and always against the golden vectors those produced (tests/golden/reference_vectors.npz)."""
import numpy as np
import pytest

import orclib
from orclib import assert_bit_equal
import parity_cases as pc
from parity_cases import golden, VOX


@pytest.mark.parametrize("mode", [0, 1, 2, 4])
def test_oracle_matches_golden_grids(orc, mode):
    G = golden()
    s = orc.init(orclib.make_config(mesh_gen_mode=mode))
    assert (s.sin_table_np().view(np.uint32) == G[f"m{mode}_state_sinTable"].view(np.uint32)).all()
    for n in orclib._STATE_FLOATS:
        assert np.float32(getattr(s, n)).view(np.uint32) == G[f"m{mode}_state_{n}"].view(np.uint32), n
    assert_bit_equal(orc.gen_grid(-64, -64, s.DX_VAL, s.DY_VAL, 130, 130, 0), G[f"m{mode}_tile00_raw"])
    assert_bit_equal(orc.gen_grid(-64, -64, s.DX_VAL, s.DY_VAL, 130, 130, 1), G[f"m{mode}_tile00_glac"])
    assert_bit_equal(orc.gen_grid(1000.0, -777.0, s.DX_VAL, s.DY_VAL, 67, 45, 1), G[f"m{mode}_odd_glac"])
    assert_bit_equal(orc.ground_mesh(), G[f"m{mode}_ground"])


def test_survey_anchor_values(orc):
    """SURVEY.md section 8c anchors (produced by the survey's own build of the reference)."""
    s = orc.init(orclib.make_config(mesh_gen_mode=0, hmap=orclib.HMAP_DEFAULT))
    assert np.float32(s.MESH_HEIGHT) == np.float32(0.400000006) and s.DX_VAL == 0.0625 and s.start_eval_sin == 0
    st = s.sin_table_np()
    np.testing.assert_array_equal(st[0], np.array([0.00455211475, 2.10988116, 5.98097086, 0.921557486, 0.217441186], np.float32))
    np.testing.assert_array_equal(st[89], np.array([1.25131238, 4.9162159, 6.09049273, 0.00157837011, 0.00172762678], np.float32))
    z = orc.gen_grid(-64, -64, s.DX_VAL, s.DY_VAL, 130, 130, 0)
    assert z[0, 0] == np.float32(0.514751375) and z[64, 64] == np.float32(0.316928059) and z[129, 129] == np.float32(-0.470116168)
    assert abs(float(z.astype(np.float64).sum()) - 7214.74102) < 1e-4
    orc.set_zmax_est(2.0)
    zg = orc.gen_grid(-64, -64, s.DX_VAL, s.DY_VAL, 130, 130, 1)
    assert zg[0, 0] == np.float32(-1.00604844) and zg[64, 64] == np.float32(-1.22264802)
    assert list(orc.rand_ints(11, 121, 3)) == [2142999984, 1939390777, 815763296]
    assert orc.simplex2(.3, .7) == np.float32(-0.442619652) and orc.perlin2(.3, .7) == np.float32(-0.427569121)
    assert orc.simplex3(.3, .7, 1.1) == np.float32(-0.0851529166) and orc.perlin3(.3, .7, 1.1) == np.float32(-0.0404006653)
    for mode, v1, v2 in ((1, -2.5190413, -2.875772), (2, 1.49259567, 0.469020426), (4, -1.32977569, -2.59287405)):
        assert orc.noise_zval(10, 20, mode) == np.float32(v1) and orc.noise_zval(-37, 5, mode) == np.float32(v2)
    tab = orc.sin_table()
    assert tab[1] == np.float32(0.000191747604) and tab[12345] == np.float32(0.69933629) and tab[32768 + 777] == np.float32(0.988921821)


def test_oracle_matches_golden_misc(orc):
    G = golden()
    hm = [0.2, 0.5, 2.0, 0.2, 0.5, 2.0, 0.0, 0.05, 4.0, 5.0, 0.001, -4.0, 1200.0, 4.0]
    s = orc.init(orclib.make_config(mesh_gen_mode=0, mesh_gen_shape=1, mesh_freq_filter=1, hmap=hm))
    assert_bit_equal(orc.gen_grid(-50, -50, s.DX_VAL, s.DY_VAL, 100, 100, 1), G["shape1_sine"])
    s = orc.init(orclib.make_config(mesh_gen_mode=1, mesh_gen_shape=2, mesh_freq_filter=1))
    assert_bit_equal(orc.gen_grid(-50, -50, s.DX_VAL, s.DY_VAL, 64, 64, 1), G["shape2_simplex"])
    s = orc.init(orclib.make_config(mesh_gen_mode=0))
    assert_bit_equal(orc.apply_erosion(G["ero_in"].copy(), float(G["ero_min"]), 400), G["ero_out_400"])
    z, st = orc.tile_create_zvals(-3, 7, 150)
    assert_bit_equal(z, G["tile_m3_7_z"]); assert bytes(st) == G["tile_m3_7_stats"].tobytes()
    nm, mnz = orc.tile_normals(z)
    assert (nm == G["tile_m3_7_normals"]).all() and np.float32(mnz) == G["tile_m3_7_min_normal_z"]
    q, mn, dz = orc.quantize16(G["ero_out_400"])
    assert (q == G["quant_bytes"]).all() and (mn, dz) == tuple(float(v) for v in G["quant_range"])
    for mode in (0, 1, 2):
        nx, ny, nz = (40, 24, 32) if mode == 0 else (12, 10, 16)
        assert_bit_equal(orc.voxel_fill(nx, ny, nz, VOX["lo"], VOX["vsz"], VOX["off"], 1.0, 1.0, 123, 456, mode, 0.01, 1), G[f"vox{mode}"])
    assert_bit_equal(orc.voxel_rdata(123, 456, 1.0, 1.0), G["vox_rdata"])
    pts = G["pts"]
    for name in ("simplex2", "perlin2"):
        assert_bit_equal(np.array([getattr(orc, name)(x, y) for x, y, _ in pts], np.float32), G[name], name)
    for name in ("simplex3", "perlin3"):
        assert_bit_equal(np.array([getattr(orc, name)(x, y, z) for x, y, z in pts], np.float32), G[name], name)
    assert_bit_equal(np.array([orc.eval_mesh_sin_terms(x, y) for x, y, _ in pts], np.float32), G["sin_terms"])
    for mode in (1, 2, 4):
        assert_bit_equal(np.array([orc.noise_zval(x, y, mode) for x, y, _ in pts[:128]], np.float32), G[f"noise_zval_{mode}"])
    assert (orc.rand_ints(11, 121, 256) == G["rand_ints_11_121"]).all()
    assert_bit_equal(orc.rand_floats(1, 12345, 256), G["rand_floats_1_12345"])
    assert_bit_equal(orc.rand_uniforms(1, 12345, 0.2, 1.0, 256), G["rand_uniforms_1_12345"])
    assert_bit_equal(orc.sin_table(), G["sin_table"])


def test_oracle_matches_golden_tile_ao(orc):
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    orc.init(orclib.make_config(mesh_gen_mode=0))
    z, _ = orc.tile_create_zvals(-3, 7, 150)
    assert (orc.tile_ao_lighting(-3, 7, z) == G["tile_m3_7_ao"]).all()
    z0, _ = orc.tile_create_zvals(0, 0, 0)
    assert (orc.tile_ao_lighting(0, 0, z0) == G["tile_0_0_ao"]).all()
    orc.init(orclib.make_config(mesh_gen_mode=4))
    orc.set_tiled_mesh_ao(1)
    try:
        z4, st4 = orc.tile_create_zvals(2, -1, 40)
        assert_bit_equal(z4, G["tile_m4ao_2_m1_z"], "AO-context zvals, mode 4")
        assert bytes(st4) == G["tile_m4ao_2_m1_stats"].tobytes()
        assert (orc.tile_ao_lighting(2, -1, z4) == G["tile_m4ao_2_m1_ao"]).all()
    finally:
        orc.set_tiled_mesh_ao(0)


def test_oracle_vs_reference_tus(orc, ref):
    """Only where /root/reference exists: the restatement against the reference's own TUs on fresh (non-golden) inputs."""
    ref.set_num_threads(1)
    for mode, n in ((0, 300), (1, 150), (2, 150), (4, 96)):
        cfg = orclib.make_config(mesh_gen_mode=mode, mesh_seed=5, mesh_freq_filter=1)
        sr, so = ref.init(cfg), orc.init(cfg)
        assert bytes(sr.sinTable) == bytes(so.sinTable) and sr.zmax_est == so.zmax_est and sr.water_plane_z == so.water_plane_z and sr.clip_hd1 == so.clip_hd1
        for gl in (0, 1):
            assert_bit_equal(ref.gen_grid(-n / 2, 11, sr.DX_VAL, sr.DY_VAL, n, n - 9, gl), orc.gen_grid(-n / 2, 11, sr.DX_VAL, sr.DY_VAL, n, n - 9, gl), f"mode {mode}")
        # mesh_xy_grid_cache_t with every argument of build_arrays / eval_index in play (force_sine_mode, cache_values, min_start_sin, use_cache)
        for fs, cv, mss, uc in ((True, 0, 50, True), (True, 1, 50, True), (True, 1, 50, False), (False, 1, 30, False), (True, 0, 70, False)):
            a = ref.gen_grid(-64, 64, 80 * sr.DX_VAL, 80 * sr.DY_VAL, 129, 129, 0, cv, mss, force_sine=fs, use_cache=uc)
            b = orc.gen_grid(-64, 64, 80 * sr.DX_VAL, 80 * sr.DY_VAL, 129, 129, 0, cv, mss, force_sine=fs, use_cache=uc)
            assert_bit_equal(a, b, f"gen_grid_ex mode {mode} {fs} {cv} {mss} {uc}")
    cfg = orclib.make_config(mesh_gen_mode=0)
    sr, so = ref.init(cfg), orc.init(cfg)
    g = ref.gen_grid(-128, -128, sr.DX_VAL, sr.DY_VAL, 256, 256, 1)
    for iters in (700, 5000):
        a, b = g.copy(), g.copy()
        ref.apply_erosion(a, float(g.min()), iters); orc.apply_erosion(b, float(g.min()), iters)
        assert_bit_equal(a, b, f"erosion {iters}")
    for tx, ty in ((1, 2), (-9, 4)):
        za, sa = ref.tile_create_zvals(tx, ty, 100); zb, sb = orc.tile_create_zvals(tx, ty, 100)
        assert_bit_equal(za, zb); assert bytes(sa) == bytes(sb)
    rng = np.random.default_rng(99)
    pts = rng.uniform(-2000, 2000, (3000, 3)).astype(np.float32)
    for name in ("simplex2", "perlin2"):
        assert_bit_equal(np.array([getattr(ref, name)(x, y) for x, y, _ in pts], np.float32), np.array([getattr(orc, name)(x, y) for x, y, _ in pts], np.float32), name)
    for name in ("simplex3", "perlin3"):
        assert_bit_equal(np.array([getattr(ref, name)(x, y, z) for x, y, z in pts], np.float32), np.array([getattr(orc, name)(x, y, z) for x, y, z in pts], np.float32), name)


def test_oracle_vs_reference_epilogue_configs(orc, ref):
    """plateau / crater / crack / volcano / custom glaciate exponent in sine mode: the configurations of case_sine_epilogue_variants."""
    ref.set_num_threads(1)
    base = list(orclib.HMAP_ISLANDS)
    variants = [{}, {0: 0.1, 1: 0.5, 2: 2.0, 3: 0.2}, {4: 0.3, 5: 2.0}, {0: 1.2}, {0: 3.0}, {6: 0.0, 7: 0.05, 8: 4.0}, {12: 1200.0, 13: 4.0}, {9: 0.0}]
    for v in variants:
        hm = list(base)
        for i, x in v.items():
            hm[i] = x
        for extra in ({}, {"custom_glaciate_exp": 2.5}, {"glaciate": 0}):
            cfg = orclib.make_config(mesh_gen_mode=0, hmap=hm, **extra)
            sr, so = ref.init(cfg), orc.init(cfg)
            assert_bit_equal(ref.gen_grid(-131, 40, sr.DX_VAL, sr.DY_VAL, 260, 150, 1), orc.gen_grid(-131, 40, sr.DX_VAL, sr.DY_VAL, 260, 150, 1), f"{v} {extra}")


def test_oracle_vs_reference_hmap_edits_files_export(orc, ref, tmp_path):
    """rest of row f4 against the reference's own heightmap.cpp (compiled in place): brushes, mod map, the .mod file both ways, read_and_apply_mod,
    interpolate / nearest sampling, heightmap_t::proc_gen, and the exporter driver (map_view.cpp is GL-bound; its pieces are the reference's)."""
    ref.set_num_threads(1)
    both = lambda f: (f(orc), f(ref))
    rng = np.random.default_rng(5)
    try:
        for ms in (1.0, 0.5, 2.0):
            for nc in (2, 1):
                cfg = orclib.make_config(mesh_gen_mode=0, mesh_scale=ms)
                pix = rng.integers(0, 256, (96, 80, 2) if nc == 2 else (96, 80), dtype=np.uint8)
                for c in (orc, ref):
                    c.init(cfg); c.hmap_set(pix.copy(), -1.5, 0.01)
                for i, row in enumerate(pc.HMAP_BRUSHES):
                    br = orclib.make_brushes([row])[0]
                    step, ns = ((1, 1), (2, 1), (1, 2), (3, 2))[i % 4]
                    for c in (orc, ref):
                        c.hmap_apply_brush(br, step, ns)
                    a, b = both(lambda c: c.hmap_pixels())
                    assert (a == b).all(), (ms, nc, row, np.argwhere(a != b)[:4])
                mods = orclib.make_mods(pc.HMAP_MODS)
                for c in (orc, ref):
                    c.hmap_apply_mods(mods)
                a, b = both(lambda c: c.hmap_pixels())
                assert (a == b).all() and (a != pix).any()
                brs = orclib.make_brushes(pc.HMAP_BRUSHES[:3])
                fo, fr = str(tmp_path / "o.mod"), str(tmp_path / "r.mod")
                assert orc.hmap_write_mod(fo, mods, brs) and ref.hmap_write_mod(fr, mods, brs)
                assert os.path.getsize(fo) == os.path.getsize(fr) == 4 + 4 + 6 * 8 + 4 + 3 * 20 + 4
                mo, bo = orc.hmap_read_mod(fr); mr, br_ = ref.hmap_read_mod(fo)  # (the reference leaves the brush padding bytes uninitialised: compare fields)
                assert mo.tobytes() == mr.tobytes() and all((bo[k] == br_[k]).all() for k in bo.dtype.names)
                assert orc.hmap_read_and_apply_mod(fr) and ref.hmap_read_and_apply_mod(fo)
                a, b = both(lambda c: c.hmap_pixels())
                assert (a == b).all()
                for x, y in ((0.3, 7.9), (-100.5, 33.25), (1e3, -2e3), (39.5, 47.5)):
                    a, b = both(lambda c: c.hmap_interpolate_height(x, y)); assert a == b
                    a, b = both(lambda c: c.hmap_get_nearest_height(x, y)); assert a == b
                a, b = both(lambda c: c.export_heightmap(-1.3, 0.7, 70, 50))
                assert (a[0] == b[0]).all() and a[1] == b[1] and a[2] == b[2]
                for c in (orc, ref):
                    c.hmap_set(None)
        for mode in (0, 1, 4):
            cfg = orclib.make_config(mesh_gen_mode=mode)
            for c in (orc, ref):
                c.init(cfg)
            for iters in (0, 300):
                a, b = both(lambda c: c.heightmap_proc_gen(96, 64, iters))
                assert (a[0] == b[0]).all() and a[1] == b[1] and a[2] == b[2], (mode, iters)
            a, b = both(lambda c: c.export_heightmap(-2.0, 1.1, 90, 33))
            assert (a[0] == b[0]).all() and a[1] == b[1] and a[2] == b[2]
    finally:
        for c in (orc, ref):
            c.hmap_set(None)


def test_oracle_matches_golden_hmap_edits(orc):
    G = pc.golden()
    orc.init(orclib.make_config(mesh_gen_mode=0))
    try:
        orc.hmap_set(G["hmap_edit_in"].copy(), -1.5, 0.01)
        for i, row in enumerate(pc.HMAP_BRUSHES):
            orc.hmap_apply_brush(orclib.make_brushes([row])[0], 1 + (i % 2), 1 + (i % 3) // 2)
        orc.hmap_apply_mods(orclib.make_mods(pc.HMAP_MODS))
        assert (orc.hmap_pixels() == G["hmap_edit_out"]).all()
        p, mn, dz = orc.export_heightmap(-1.3, 0.7, 40, 30)
        assert (p == G["hmap_export_pix"]).all() and (np.array([mn, dz], np.float32).view(np.uint32) == G["hmap_export_range"].view(np.uint32)).all()
    finally:
        orc.hmap_set(None)
    p, sc, tz = orc.heightmap_proc_gen(64, 48, 200)
    assert (p == G["proc_gen_pix"]).all() and (np.array([sc, tz], np.float32).view(np.uint32) == G["proc_gen_scale_tz"].view(np.uint32)).all()


def test_oracle_matches_golden_heightmap_postprocess(orc):
    """rest of row a12 (heightmap_t::to_floats / from_floats / postprocess_height): the restatement against the reference's own members (golden)"""
    G = golden()
    for key, scale_tz in (("island128", pc.ISLAND_SCALE_TZ), ("rand16", (170.0, -17.0))):
        pix = G[f"pp_{key}_in"]
        pc.island_setup(orc, pc.island_cfg(orclib.make_config), scale_tz=scale_tz)
        v = orc.heightmap_to_floats(pix)
        assert_bit_equal(v, G[f"pp_{key}_vals"], key)
        pc.island_setup(orc, pc.island_cfg(orclib.make_config), (v.min(), v.max()), scale_tz)
        o, bad = orc.heightmap_postprocess(pix, int(G[f"pp_{key}_iters"]))
        assert bad == 0 and (o == G[f"pp_{key}_out"]).all()
        f, badf = orc.heightmap_from_floats(v, 2 if pix.ndim == 3 else 1)
        assert badf == 0 and (f == G[f"pp_{key}_from"]).all()
    orc.set_mesh_file_scale(1.0, 0.0)


def test_oracle_vs_reference_heightmap_postprocess(orc, ref, pkg, emul_lib):
    """the same against the reference's heightmap.cpp compiled in place, on the island image (8 bit) and random 8- / 16-bit images, incl. the out-of-range count"""
    import os
    island = pkg.terra.read_png(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "heightmap_island_128.png"), lib=pkg.terra.load_library(emul_lib))
    assert (island == golden()["pp_island128_in"]).all()
    rng = np.random.default_rng(21)
    imgs = [(island, pc.ISLAND_SCALE_TZ, 2500)]
    for k in range(3):
        h, w = int(rng.integers(30, 90)), int(rng.integers(30, 90))
        yy, xx = np.mgrid[0:h, 0:w]
        hgt = 128 + rng.uniform(20, 70) * np.sin(xx * rng.uniform(0.05, 0.2)) * np.cos(yy * rng.uniform(0.05, 0.2)) + rng.uniform(-5, 5, xx.shape)
        pix = hgt.astype(np.uint8) if k == 0 else np.stack([((hgt % 1.0) * 256).astype(np.uint8), hgt.astype(np.uint8)], axis=-1)
        imgs.append((np.ascontiguousarray(pix), (float(rng.uniform(100, 250)), float(rng.uniform(-25, -10))), 900))
    try:
        for pix, scale_tz, iters in imgs:
            for ck in (orc, ref):
                pc.island_setup(ck, pc.island_cfg(orclib.make_config), scale_tz=scale_tz)
            vo, vr = orc.heightmap_to_floats(pix), ref.heightmap_to_floats(pix)
            assert_bit_equal(vo, vr, "to_floats")
            for ck in (orc, ref):
                pc.island_setup(ck, pc.island_cfg(orclib.make_config), (vo.min(), vo.max()), scale_tz)
            oo, ob = orc.heightmap_postprocess(pix, iters); ro, rb = ref.heightmap_postprocess(pix, iters)
            assert ob == rb == 0 and (oo == ro).all() and (oo != pix).any()
            nc = 2 if pix.ndim == 3 else 1
            fo, fb = orc.heightmap_from_floats(vo, nc); fr, fbr = ref.heightmap_from_floats(vo, nc)
            assert fb == fbr == 0 and (fo == fr).all()
            shifted = vo + np.float32(200.0 * 0.0008 * 0.7 * scale_tz[0])  # + 200 pixel units: many values above 256: both count them (the member itself would assert)
            assert orc.heightmap_from_floats(shifted, nc)[1] == ref.heightmap_from_floats(shifted, nc)[1] > 0
    finally:
        orc.set_mesh_file_scale(1.0, 0.0); ref.set_mesh_file_scale(1.0, 0.0)


def test_oracle_matches_golden_tile_weights(orc):
    G = pc.golden()
    orc.init(orclib.make_config(mesh_gen_mode=0))
    orc.set_landscape(orclib.make_landscape(grass_density=100))
    try:
        z, _ = orc.tile_create_zvals(-3, 2, 0)
        w, gb, hg = orc.tile_create_weights(-3, 2, z)
        assert (w == G["tile_m3_2_weights"]).all() and gb.tobytes() == G["tile_m3_2_grass_blocks"].tobytes() and hg == bool(G["tile_m3_2_has_grass"])
        prm = np.stack([orc.tile_terrain_params(-3, 2), orc.tile_terrain_params(40, 41)])
        assert (prm.view(np.uint32) == G["tile_terrain_params"].view(np.uint32)).all()
    finally:
        orc.set_landscape(orclib.make_landscape())


def test_oracle_vs_reference_tile_weights(orc, ref):
    """row f3 against oracle/_ref: the noise field, the biome parameters, the texture height tables and the water level come from the reference's own
    functions there; the blend itself is the shim's statement-by-statement driver of tile_t::create_texture (tiled_mesh.cpp cannot be built here)."""
    ref.set_num_threads(1)  # the reference's erosion loop races under OpenMP
    try:
        for mode, shape, tweaks, lkw, iters, tiles in pc.LANDSCAPE_CASES:
            for c in (orc, ref):
                cfg = orclib.make_config(mesh_gen_mode=mode, mesh_gen_shape=shape, mesh_scale=tweaks.get("mesh_scale", 1.0))
                cfg.water_h_off_rel = tweaks.get("water_h_off_rel", 0.0); cfg.relh_adj_tex = tweaks.get("relh_adj_tex", 0.0)
                c.init(cfg); c.set_landscape(orclib.make_landscape(**lkw))
            for tx, ty in tiles[:6]:
                za, _ = ref.tile_create_zvals(tx, ty, iters)
                zb, _ = orc.tile_create_zvals(tx, ty, iters)
                assert_bit_equal(za, zb, "zvals")
                a, b = ref.tile_create_weights(tx, ty, za), orc.tile_create_weights(tx, ty, zb)
                assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes() and a[2] == b[2], (mode, shape, lkw, tx, ty)
                assert_bit_equal(ref.tile_terrain_params(tx, ty), orc.tile_terrain_params(tx, ty), "terrain params")
    finally:
        for c in (orc, ref):
            c.set_landscape(orclib.make_landscape())


def test_oracle_vs_reference_tile_ao(orc, ref):
    """row f1 (tile_t::calc_mesh_ao_lighting) and the AO-context variant of create_zvals (enable_tiled_mesh_ao with the GL modes)."""
    ref.set_num_threads(1)
    try:
        for mode, ao_flag, tiles in ((0, 0, [(0, 0), (-3, 2), (5, -7)]), (1, 1, [(1, 1)]), (4, 1, [(0, 0), (2, -1)]), (3, 1, [(-1, 0)]), (4, 0, [(0, 0)])):
            cfg = orclib.make_config(mesh_gen_mode=mode)
            ref.init(cfg); orc.init(cfg)
            ref.set_tiled_mesh_ao(ao_flag); orc.set_tiled_mesh_ao(ao_flag)
            for tx, ty in tiles:
                for iters in (0, 60):
                    za, sa = ref.tile_create_zvals(tx, ty, iters); zb, sb = orc.tile_create_zvals(tx, ty, iters)
                    assert_bit_equal(za, zb, f"zvals mode {mode} ao {ao_flag} tile {tx},{ty} iters {iters}")
                    a, b = ref.tile_ao_lighting(tx, ty, za), orc.tile_ao_lighting(tx, ty, zb)
                    assert (a == b).all(), f"ao mode {mode} tile {tx},{ty}: {(a != b).sum()} texels differ"
                    assert a.min() < 255 or za.max() - za.min() < 1e-3  # something is occluded on any non-flat tile
    finally:
        ref.set_tiled_mesh_ao(0); orc.set_tiled_mesh_ao(0)


def _hmap_pixels(ck, n=192, mode=0):
    """a small eroded heightmap quantised like heightmap_t::from_floats: (n, n, 2) bytes + (min_z, dz/255)"""
    s = ck.init(orclib.make_config(mesh_gen_mode=mode))
    g = ck.gen_grid(-n / 2, -n / 2, s.DX_VAL, s.DY_VAL, n, n, 1)
    q, mn, dz = ck.quantize16(g)
    return np.ascontiguousarray(q.reshape(n, n, 2)), float(mn), float(np.float32(np.float64(dz) / 255.0))


def test_oracle_vs_reference_hmap_tiles(orc, ref):
    """tiles sampled from a heightmap texture (terrain_hmap_manager_t): nearest / bilinear / mirror-wrapped reads, procedural detail, AO context"""
    ref.set_num_threads(1)
    pix, mn, dzs = _hmap_pixels(ref)
    pix8 = np.ascontiguousarray(pix[:, :, 1])
    try:
        for mesh_scale, img in ((1.0, pix), (2.0, pix), (0.8, pix), (0.5, pix), (1.0, pix8), (0.6, pix8)):
            cfg = orclib.make_config(mesh_gen_mode=0, mesh_scale=mesh_scale)
            ref.init(cfg); orc.init(cfg)
            ref.hmap_set(img, mn, dzs); orc.hmap_set(img, mn, dzs)
            for x, y in ((0, 0), (-96, 95), (500, -777), (-100000, 123456), (97, 96)):
                assert np.float32(ref.get_clamped_height(x, y)).view(np.uint32) == np.float32(orc.get_clamped_height(x, y)).view(np.uint32), (mesh_scale, x, y)
            for tx, ty in ((0, 0), (-1, 0), (1, -2), (7, 5)):
                za, sa = ref.tile_create_zvals(tx, ty, 50); zb, sb = orc.tile_create_zvals(tx, ty, 50)
                assert_bit_equal(za, zb, f"hmap zvals scale {mesh_scale} tile {tx},{ty}")
                assert bytes(sa) == bytes(sb)
                a, b = ref.tile_ao_lighting(tx, ty, za), orc.tile_ao_lighting(tx, ty, zb)
                assert (a == b).all(), f"hmap ao scale {mesh_scale} tile {tx},{ty}"
    finally:
        ref.hmap_set(None); orc.hmap_set(None)


def test_oracle_vs_reference_point_queries(orc, ref):
    """a8: eval_mesh_sin_terms_scaled / get_exact_zval of the restatement against the reference's own functions (src/mesh_gen.cpp:807-847, compiled in place), all noise
    modes, scroll offsets, the heightmap-texture branch with and without detail noise"""
    import parity_cases as pc
    ref.set_num_threads(1)
    vol = [1000.0, 0, 0, 0, 1000.0, 0, 0, 0, 0, 5.0, 0.001, -4.0, 1200.0, 4.0]
    for kw in (dict(mesh_gen_mode=0), dict(mesh_gen_mode=0, mesh_gen_shape=1, mesh_freq_filter=1), dict(mesh_gen_mode=1), dict(mesh_gen_mode=2, mesh_gen_shape=2), dict(mesh_gen_mode=3),
               dict(mesh_gen_mode=4), dict(mesh_gen_mode=0, hmap=vol), dict(mesh_gen_mode=1, custom_glaciate_exp=2.5), dict(mesh_gen_mode=0, glaciate=0), dict(mesh_gen_mode=2, mesh_scale=0.5)):
        cfg = orclib.make_config(**kw)
        ref.init(cfg); orc.init(cfg)
        xy = pc.points_sample(400, 3, 40.0)
        for (nox, xo, yo) in ((False, 0, 0), (False, 137, -4021), (True, 55, 66)):
            assert_bit_equal(ref.eval_points(xy, True, no_xyoff=nox, xoff2=xo, yoff2=yo), orc.eval_points(xy, True, no_xyoff=nox, xoff2=xo, yoff2=yo), f"get_exact_zval {kw} {nox} {xo} {yo}")
        ixy = pc.points_sample(400, 4, 300.0)
        for sc in (1.0, 16.0, 0.37):
            assert_bit_equal(ref.eval_points(ixy, False, xy_scale=sc), orc.eval_points(ixy, False, xy_scale=sc), f"eval_mesh_sin_terms_scaled {kw} {sc}")
    pix, mn, dzs = _hmap_pixels(ref)
    try:
        for mesh_scale, mode in ((1.0, 0), (0.5, 0), (0.6, 1), (2.0, 2)):
            cfg = orclib.make_config(mesh_gen_mode=mode, mesh_scale=mesh_scale)
            ref.init(cfg); orc.init(cfg)
            ref.hmap_set(pix, mn, dzs); orc.hmap_set(pix, mn, dzs)
            xy = pc.points_sample(400, 5, 12.0)
            for (nox, xo, yo) in ((False, 0, 0), (False, -31, 77), (True, 0, 0)):
                assert_bit_equal(ref.eval_points(xy, True, no_xyoff=nox, xoff2=xo, yoff2=yo), orc.eval_points(xy, True, no_xyoff=nox, xoff2=xo, yoff2=yo), f"hmap texture points scale {mesh_scale} mode {mode}")
    finally:
        ref.hmap_set(None); orc.hmap_set(None)


LIGHTS = [(0.6, 0.5, 0.4), (-0.8, 0.3, 0.25), (0.2, -0.9, 0.15), (-0.5, -0.5, 0.8), (1.0, 0.0, 0.3), (0.0, -1.0, 0.2), (0.3, 0.4, -5.0), (0.0, 0.0, 1.0), (0.05, 0.9, 0.02)]


def test_oracle_vs_reference_mesh_shadows(orc, ref):
    """row f2: the reference's own calc_mesh_shadows / mesh_shadow_gen / do_line_clip (visibility.cpp, Math3d.cpp compiled in place) against the
    restatement: single meshes of several shapes with and without incoming edge shadows, then chained tile batches."""
    ref.set_num_threads(1)  # the two OpenMP sections race on smask / sh_out: one thread is the defined order
    cfg = orclib.make_config(mesh_gen_mode=0)
    sr = ref.init(cfg); orc.init(cfg)
    rng = np.random.default_rng(5)
    for shape in ((130, 130), (64, 96), (33, 17)):
        g = ref.gen_grid(-40, 25, sr.DX_VAL, sr.DY_VAL, shape[1], shape[0], 1)
        g = (g * np.float32(3.0)).astype(np.float32)  # steeper: more shadow
        for lp in LIGHTS:
            for with_in in (False, True):
                six = (g.max() + rng.uniform(-0.3, 0.3, shape[1])).astype(np.float32) if with_in else None
                siy = np.where(rng.uniform(size=shape[0]) < 0.5, np.float32(-1.0e6), g.mean() + rng.uniform(-0.2, 0.4, shape[0])).astype(np.float32) if with_in else None
                a = ref.calc_mesh_shadows(lp, g, six, siy); b = orc.calc_mesh_shadows(lp, g, six, siy)
                assert (a[0] == b[0]).all(), f"smask {shape} light {lp} in {with_in}: {(a[0] != b[0]).sum()} cells"
                assert_bit_equal(a[1], b[1], "sh_out_x"); assert_bit_equal(a[2], b[2], "sh_out_y")
    tiles = [(tx, ty) for ty in range(-1, 2) for tx in range(0, 3)] + [(7, 7)]
    z = np.stack([ref.tile_create_zvals(tx, ty, 0)[0] for tx, ty in tiles]) * np.float32(4.0)
    for lp in LIGHTS:
        a = ref.tiles_mesh_shadows(tiles, z, lp); b = orc.tiles_mesh_shadows(tiles, z, lp)
        assert (a == b).all(), f"tile batch light {lp}: {(a != b).sum()} cells"
    assert any(ref.tiles_mesh_shadows(tiles, z, lp).any() for lp in LIGHTS)


def test_libm_sincosf_is_not_correctly_rounded_but_reproducible():
    """The droplet's random-direction branch calls libm cosf/sinf on a = rand_float()*TWO_PI (src/erosion.cpp:80-83).
    glibc's sinf/cosf are not correctly rounded, which is why 3dworld_amd/csrc/terra_sincosf.hpp restates their algorithm
    instead of using (float)cos((double)a); this documents the fact on a sample (the exhaustive comparison of the restatement
    is the flat-terrain erosion case, CPU emulator and GPU)."""
    import ctypes
    m = ctypes.CDLL("libm.so.6")
    m.cosf.restype = m.sinf.restype = ctypes.c_float
    m.cosf.argtypes = m.sinf.argtypes = [ctypes.c_float]
    two_pi = np.float32(2.0 * np.float64(np.float32(3.141592654)))
    a = (0.000001 * np.arange(0, 1000000, 50)).astype(np.float32) * two_pi
    diff = sum(np.float32(m.cosf(float(v))) != np.float32(np.cos(np.float64(v))) for v in a)
    assert diff > 0  # ~2.6% on glibc 2.35


@pytest.mark.parametrize("block", [0, 1])
def test_oracle_vs_reference_random_configs(orc, ref, block):
    """the fixed-seed random sweep of parity_cases.case_random_configs, oracle against the reference build: grids (arbitrary origin / spacing / size,
    min_start_sin), tile zvals + stats + normals + AO + weights + shadows, whole-map erosion, over random modes / shapes / post-processing / island /
    volcano parameters, scales, water levels and landscape globals"""
    ref.set_num_threads(1)
    try:
        for seed in range(block * 10, block * 10 + 10):
            rng = np.random.default_rng(7000 + seed)
            mode = int(rng.choice([0, 0, 1, 2, 3, 4])); shape = int(rng.choice([0, 0, 1, 2]))
            hm = list(orclib.HMAP_DEFAULT)
            if rng.random() < 0.5: hm[0:4] = [float(rng.uniform(-1, 1.5)), float(rng.uniform(0, 1)), float(rng.uniform(0, 2)), float(rng.uniform(0, 1))]
            if rng.random() < 0.4: hm[4:6] = [float(rng.uniform(0, 2)), float(rng.uniform(0, 3))]
            if rng.random() < 0.4: lo = float(rng.uniform(-1, 1)); hm[6:9] = [lo, lo + float(rng.uniform(0.05, 1)), float(rng.uniform(0, 2))]
            if rng.random() < 0.6: hm[9:12] = [float(rng.uniform(0.5, 6)), float(rng.uniform(0.0005, 0.01)), float(rng.uniform(-5, 1))]
            if rng.random() < 0.3 and hm[9] > 0: hm[12:14] = [float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.5, 3))]
            cfg = orclib.make_config(mesh_gen_mode=mode, mesh_gen_shape=shape, mesh_seed=int(rng.integers(1, 50)), mesh_freq_filter=int(rng.integers(0, 4)), hmap=hm,
                                     glaciate=int(rng.random() < 0.85), mesh_scale=float(rng.choice([1.0, 1.0, 0.5, 2.0, 1.37])), mesh_height=float(rng.uniform(0.3, 1.5)),
                                     erode_amount=float(rng.choice([1.0, 1.0, 0.4, 2.5])))
            cfg.water_h_off_rel = float(rng.choice([0.0, 0.0, 0.1, -0.15])); cfg.relh_adj_tex = float(rng.choice([0.0, 0.0, 0.05, -0.04])); cfg.water_h_off = float(rng.choice([0.0, 0.0, 0.2]))
            lkw = dict(vegetation=float(rng.choice([1.0, 1.0, 0.0, 0.5])), temperature=float(rng.choice([20.0, 20.0, 48.0])), biome_x_offset=float(rng.uniform(-5, 5)),
                       water_is_lava=int(rng.random() < 0.2), disable_water=int(rng.choice([0, 0, 2])), enable_terrain_env=int(rng.random() < 0.8),
                       grass_density=int(rng.choice([0, 50])), num_rnd_grass_blocks=int(rng.integers(1, 33)))
            sa, sb = ref.init(cfg), orc.init(cfg)
            for n_ in orclib._STATE_FLOATS:
                assert np.float32(getattr(sa, n_)).view(np.uint32) == np.float32(getattr(sb, n_)).view(np.uint32), (seed, n_)
            for c in (ref, orc):
                c.set_landscape(orclib.make_landscape(**lkw))
            nx, ny = int(rng.integers(1, 160)), int(rng.integers(1, 120))
            x0, y0 = float(rng.uniform(-5000, 5000)), float(rng.uniform(-5000, 5000))
            dx, dy = sa.DX_VAL * float(rng.choice([1.0, 1.0, 16.0, 0.37])), sa.DY_VAL * float(rng.choice([1.0, 1.0, 80.0, 2.5]))
            gl, mss = int(rng.random() < 0.7), int(rng.choice([0, 0, 50, 23]))
            assert_bit_equal(ref.gen_grid(x0, y0, dx, dy, nx, ny, gl, 0, mss), orc.gen_grid(x0, y0, dx, dy, nx, ny, gl, 0, mss), f"grid seed {seed}")
            tiles = [(int(rng.integers(-60, 60)), int(rng.integers(-60, 60)))]
            tiles += [(tiles[0][0] + 1, tiles[0][1]), (tiles[0][0], tiles[0][1] + 1)]
            iters = int(rng.choice([0, 0, 40, 150])); ao_flag = int(rng.random() < 0.5)
            for c in (ref, orc):
                c.set_tiled_mesh_ao(ao_flag)
            za = []
            for tx, ty in tiles:
                a, sta = ref.tile_create_zvals(tx, ty, iters); b, stb = orc.tile_create_zvals(tx, ty, iters)
                assert_bit_equal(a, b, f"tile zvals seed {seed} {tx},{ty}"); assert bytes(sta) == bytes(stb)
                na, ma = ref.tile_normals(a); nb, mb = orc.tile_normals(b)
                assert (na == nb).all() and np.float32(ma).view(np.uint32) == np.float32(mb).view(np.uint32)
                assert (ref.tile_ao_lighting(tx, ty, a) == orc.tile_ao_lighting(tx, ty, b)).all(), (seed, "ao")
                wa, wb = ref.tile_create_weights(tx, ty, a), orc.tile_create_weights(tx, ty, b)
                assert (wa[0] == wb[0]).all() and wa[1].tobytes() == wb[1].tobytes() and wa[2] == wb[2], (seed, "weights")
                za.append(a)
            light = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(0.05, 1)))
            assert (ref.tiles_mesh_shadows(tiles, np.stack(za), light) == orc.tiles_mesh_shadows(tiles, np.stack(za), light)).all(), (seed, "shadows")
            n = int(rng.integers(40, 120)); d = int(rng.integers(1, 300))
            g = orc.gen_grid(-n / 2, -n / 2, sa.DX_VAL, sa.DY_VAL, n, n, 1)
            mz = float(g.min()) if rng.random() < 0.7 else float(g.min()) + 0.1
            ga = ref.apply_erosion(g.copy(), mz, d); gb = orc.apply_erosion(g.copy(), mz, d)
            assert_bit_equal(ga, gb, f"erosion seed {seed}")
    finally:
        for c in (ref, orc):
            c.set_tiled_mesh_ao(0); c.set_landscape(orclib.make_landscape())


def test_oracle_vs_reference_random_voxels(orc, ref):
    """noise_gen_3d (the reference's own upsurface.cpp) and the 3-D fBm fill loop over random dimensions, origins, steps, seeds, frequency filters"""
    rng = np.random.default_rng(2024)
    for k in range(12):
        ff = int(rng.integers(0, 4))
        cfg = orclib.make_config(mesh_gen_mode=0, mesh_freq_filter=ff, mesh_seed=int(rng.integers(1, 30)))
        ref.init(cfg); orc.init(cfg)
        mode = int(rng.choice([0, 0, 1, 2]))
        nx, ny, nz = (int(rng.integers(1, 40)) for _ in range(3))
        if mode:
            nx, ny, nz = min(nx, 14), min(ny, 12), min(nz, 16)
        lo = tuple(float(v) for v in rng.uniform(-5, 5, 3)); vsz = tuple(float(v) for v in rng.uniform(0.005, 0.3, 3)); off = tuple(float(v) for v in rng.uniform(-1, 1, 3))
        mag, freq = float(rng.uniform(0.2, 2.0)), float(rng.uniform(0.3, 3.0))
        rs1, rs2 = int(rng.integers(1, 10000)), int(rng.integers(1, 10000))
        zscale, norm = float(rng.choice([0.0, 0.01, -0.05])), int(rng.random() < 0.7)
        a = ref.voxel_fill(nx, ny, nz, lo, vsz, off, mag, freq, rs1, rs2, mode, zscale, norm)
        b = orc.voxel_fill(nx, ny, nz, lo, vsz, off, mag, freq, rs1, rs2, mode, zscale, norm)
        assert_bit_equal(a, b, f"voxels case {k} mode {mode} {nx}x{ny}x{nz}")
        assert_bit_equal(ref.voxel_rdata(rs1, rs2, mag, freq), orc.voxel_rdata(rs1, rs2, mag, freq), "rdata")


def test_oracle_vs_reference_random_heightmap_textures(orc, ref):
    """terrain_hmap_manager_t (the reference's heightmap.cpp) over random image sizes (odd ones too), 8- and 16-bit pixels, mesh scales on both sides of 1
    and of the 0.75 detail threshold: point samples far outside the image (mirror wrap), tiles with stats and AO, the exporter"""
    rng = np.random.default_rng(77)
    ref.set_num_threads(1)
    try:
        for k in range(8):
            ms = float(rng.choice([0.3, 0.6, 0.74, 0.75, 0.9, 1.0, 1.5, 3.0]))
            cfg = orclib.make_config(mesh_gen_mode=int(rng.choice([0, 0, 1])), mesh_scale=ms)
            w, h = int(rng.integers(3, 150)), int(rng.integers(3, 150))
            nc = int(rng.choice([1, 2]))
            pix = rng.integers(0, 256, (h, w, 2) if nc == 2 else (h, w), dtype=np.uint8)
            mn, dzs = float(rng.uniform(-3, 0)), float(rng.uniform(0.001, 0.03))
            for c in (ref, orc):
                c.init(cfg); c.hmap_set(pix.copy(), mn, dzs)
            for _ in range(300):
                x, y = int(rng.integers(-100000, 100000)), int(rng.integers(-100000, 100000))
                assert ref.get_clamped_height(x, y) == orc.get_clamped_height(x, y), (k, x, y)
                fx, fy = float(rng.uniform(-1e4, 1e4)), float(rng.uniform(-1e4, 1e4))
                assert ref.hmap_interpolate_height(fx, fy) == orc.hmap_interpolate_height(fx, fy)
                assert ref.hmap_get_nearest_height(fx, fy) == orc.hmap_get_nearest_height(fx, fy)
            for tx, ty in ((0, 0), (int(rng.integers(-50, 50)), int(rng.integers(-50, 50)))):
                a, sa = ref.tile_create_zvals(tx, ty, 25); b, sb = orc.tile_create_zvals(tx, ty, 25)
                assert_bit_equal(a, b, f"hmap tile case {k} {tx},{ty} scale {ms}"); assert bytes(sa) == bytes(sb)
                assert (ref.tile_ao_lighting(tx, ty, a) == orc.tile_ao_lighting(tx, ty, b)).all()
            xs, ys = float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))
            ea, eb = ref.export_heightmap(xs, ys, 45, 31), orc.export_heightmap(xs, ys, 45, 31)
            assert (ea[0] == eb[0]).all() and ea[1] == eb[1] and ea[2] == eb[2], (k, "export")
    finally:
        for c in (ref, orc):
            c.hmap_set(None)


def test_oracle_vs_reference_grid_rectangles(orc, ref):
    """orc.gen_grid_rect (eval_index inside a rectangle of a large grid's generator) against the reference's own mesh_xy_grid_cache_t, and against the full double loop"""
    for mode, n in ((0, 700), (1, 300), (4, 130)):
        cfg = orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=1)
        sr, so = ref.init(cfg), orc.init(cfg)
        full = orc.gen_grid(-n / 2, -n / 2, sr.DX_VAL, sr.DY_VAL, n, n, 1)
        for rx0, ry0, rw, rh in ((0, 0, n, 3), (0, n - 5, n, 5), (n // 3, 0, 4, n), (17, 29, 100, 64)):
            a = ref.gen_grid_rect(-n / 2, -n / 2, sr.DX_VAL, sr.DY_VAL, n, n, rx0, ry0, rw, rh)
            b = orc.gen_grid_rect(-n / 2, -n / 2, sr.DX_VAL, sr.DY_VAL, n, n, rx0, ry0, rw, rh)
            assert_bit_equal(a, b, f"rect mode {mode}")
            assert_bit_equal(b, full[ry0:ry0 + rh, rx0:rx0 + rw], f"rect vs full mode {mode}")


def test_oracle_vs_reference_mesh_seed_zero_fresh_process(ref):
    """mesh_seed 0: the reference's function-static generator (src/mesh_gen.cpp:238) and the oracle's, each in a fresh process, over the same four scene starts"""
    from test_gpu_timed_sizes import seed0_oracle_sequence
    a, b = seed0_oracle_sequence("ref"), seed0_oracle_sequence("orc")
    assert len(a) == len(b) == 4
    for k in range(4):
        for key in ("sinTable", "rx", "ry", "zmax_est", "grid"):
            assert a[k][key] == b[k][key], (k, key)
    assert a[0]["sinTable"] != a[1]["sinTable"]


def test_oracle_read_write_mesh_vs_golden_and_reference(orc, ref, tmp_path):
    """read_mesh / write_mesh (src/mesh_gen.cpp:895-965): the restatement against the reference's own reader on its own mapx/mesh128.txt (golden; the real file too when
    the reference tree is here), the writer's text byte for byte, and a mesh with extreme values through both"""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    src = tmp_path / "mesh128.txt"
    src.write_bytes(G["rm_mesh128_txt"].tobytes())
    real = "/root/reference/mapx/mesh128.txt"
    for key, (scale, tz, zmm) in (("plain", (1.0, 0.0, 0.0)), ("scaled", (2.5, -0.75, 3.0))):
        for path in [str(src)] + ([real] if os.path.exists(real) else []):
            for ck in (orc, ref):
                ck.init(orclib.make_config(mesh_gen_mode=0))
                ck.set_mesh_file_scale(scale, tz)
                ok, zz = ck.read_mesh(path, zmm)
                assert ok
                assert ck.ground_mesh().tobytes() == G[f"rm_{key}_mesh"].tobytes(), (key, ck.kind)
                assert np.array(zz, np.float32).tobytes() == G[f"rm_{key}_zbottom_ztop"].tobytes()
                st = ck.state()
                for k in ("zmin", "zmax", "zmax_est", "water_plane_z"):
                    assert np.float32(getattr(st, k)).tobytes() == G[f"rm_{key}_state_{k}"].tobytes(), (key, ck.kind, k)
    rng = np.random.default_rng(5)
    m = (rng.standard_normal((128, 128)) * 10.0 ** rng.integers(-8, 9, (128, 128))).astype(np.float32)
    texts = []
    for ck in (orc, ref):
        ck.init(orclib.make_config(mesh_gen_mode=0))
        ck.set_mesh_file_scale(1.0, 0.0)
        p = tmp_path / f"w_{ck.kind}.txt"
        assert ck.write_mesh(str(p), m)
        texts.append(p.read_bytes())
    assert texts[0] == texts[1]
    backs = []
    for ck in (orc, ref):
        ok, zz = ck.read_mesh(str(tmp_path / "w_orc.txt"))
        assert ok
        backs.append((ck.ground_mesh().tobytes(), tuple(zz), ck.state().zmax_est))
    assert backs[0] == backs[1]
    assert not orc.read_mesh(str(tmp_path / "nope.txt"))[0] and not ref.read_mesh(str(tmp_path / "nope.txt"))[0]
