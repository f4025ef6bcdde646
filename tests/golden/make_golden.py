"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref = 3DWorld's own mesh_gen.cpp / erosion.cpp /
upsurface.cpp compiled in place, see oracle/Makefile).  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The fixtures pin the C restatement (oracle/terra_oracle.c) and, transitively, the HIP path on machines where the reference
tree does not exist (the GPU box).  All inputs are seeded; erosion runs with OMP_NUM_THREADS=1,
the only deterministic order of the reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orclib  # noqa: E402

MODES = (0, 1, 2, 4)
VOX = dict(lo=(-3.9, -3.9, -1.0), vsz=(0.0152, 0.0152, 0.0625), off=(0.1, 0.2, 0.3), mag=1.0, freq=1.0, rs1=123, rs2=456, zscale=0.01)


def state_dict(s):
    d = {"sinTable": s.sin_table_np(), "start_eval_sin": np.int32(s.start_eval_sin)}
    for n in orclib._STATE_FLOATS:
        d[n] = np.float32(getattr(s, n))
    return d


def main():
    orclib.build_oracle()
    assert orclib.ref_available(), "needs /root/reference (oracle/_ref)"
    R = orclib.Checker("ref")
    R.set_num_threads(1)
    out = {}
    for mode in MODES:
        s = R.init(orclib.make_config(mesh_gen_mode=mode))
        for k, v in state_dict(s).items():
            out[f"m{mode}_state_{k}"] = v
        out[f"m{mode}_tile00_raw"] = R.gen_grid(-64, -64, s.DX_VAL, s.DY_VAL, 130, 130, 0)
        out[f"m{mode}_tile00_glac"] = R.gen_grid(-64, -64, s.DX_VAL, s.DY_VAL, 130, 130, 1)
        out[f"m{mode}_odd_glac"] = R.gen_grid(1000.0, -777.0, s.DX_VAL, s.DY_VAL, 67, 45, 1)
        out[f"m{mode}_ground"] = R.ground_mesh()
    # shapes / post-process / 8 octaves in sine mode
    s = R.init(orclib.make_config(mesh_gen_mode=0, mesh_gen_shape=1, mesh_freq_filter=1, hmap=[0.2, 0.5, 2.0, 0.2, 0.5, 2.0, 0.0, 0.05, 4.0, 5.0, 0.001, -4.0, 1200.0, 4.0]))
    out["shape1_sine"] = R.gen_grid(-50, -50, s.DX_VAL, s.DY_VAL, 100, 100, 1)
    s = R.init(orclib.make_config(mesh_gen_mode=1, mesh_gen_shape=2, mesh_freq_filter=1))
    out["shape2_simplex"] = R.gen_grid(-50, -50, s.DX_VAL, s.DY_VAL, 64, 64, 1)
    # erosion + tiles + quantise (sine mode)
    s = R.init(orclib.make_config(mesh_gen_mode=0))
    g = R.gen_grid(-80, -80, s.DX_VAL, s.DY_VAL, 160, 160, 1)
    out["ero_in"] = g.copy()
    out["ero_min"] = np.float32(g.min())
    out["ero_out_400"] = R.apply_erosion(g.copy(), float(g.min()), 400)
    z, st = R.tile_create_zvals(-3, 7, 150)
    out["tile_m3_7_z"] = z
    out["tile_m3_7_stats"] = np.frombuffer(bytes(st), np.uint8).copy()
    nm, mnz = R.tile_normals(z)
    out["tile_m3_7_normals"] = nm
    out["tile_m3_7_min_normal_z"] = np.float32(mnz)
    q, mn, dz = R.quantize16(out["ero_out_400"])
    out["quant_bytes"] = q
    out["quant_range"] = np.array([mn, dz], np.float32)
    out["max_sea_level"] = np.float32(R.get_max_sea_level())
    # row f1: tile AO lighting (tile_t::calc_mesh_ao_lighting) and the AO-context create_zvals of the GL modes (enable_tiled_mesh_ao)
    out["tile_m3_7_ao"] = R.tile_ao_lighting(-3, 7, z)
    z0, _ = R.tile_create_zvals(0, 0, 0)
    out["tile_0_0_ao"] = R.tile_ao_lighting(0, 0, z0)
    R.init(orclib.make_config(mesh_gen_mode=4))
    R.set_tiled_mesh_ao(1)
    z4, st4 = R.tile_create_zvals(2, -1, 40)
    out["tile_m4ao_2_m1_z"] = z4
    out["tile_m4ao_2_m1_stats"] = np.frombuffer(bytes(st4), np.uint8).copy()
    out["tile_m4ao_2_m1_ao"] = R.tile_ao_lighting(2, -1, z4)
    R.set_tiled_mesh_ao(0)
    s = R.init(orclib.make_config(mesh_gen_mode=0))
    # rest of row f4, produced by the reference's own heightmap.cpp: brushes + mods on a random 16-bit image, the exporter over it, heightmap_t::proc_gen
    import parity_cases as pc_
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (96, 80, 2), dtype=np.uint8)
    out["hmap_edit_in"] = img
    R.set_num_threads(1)
    R.hmap_set(img.copy(), -1.5, 0.01)
    for i, row in enumerate(pc_.HMAP_BRUSHES):
        R.hmap_apply_brush(orclib.make_brushes([row])[0], 1 + (i % 2), 1 + (i % 3) // 2)
    R.hmap_apply_mods(orclib.make_mods(pc_.HMAP_MODS))
    out["hmap_edit_out"] = R.hmap_pixels()
    p, mn, dz = R.export_heightmap(-1.3, 0.7, 40, 30)
    out["hmap_export_pix"] = p
    out["hmap_export_range"] = np.array([mn, dz], np.float32)
    R.hmap_set(None)
    p, sc, tz = R.heightmap_proc_gen(64, 48, 200)
    out["proc_gen_pix"] = p
    out["proc_gen_scale_tz"] = np.array([sc, tz], np.float32)
    # rest of row a12: the reference's own heightmap_t::to_floats / from_floats / postprocess_height on the 8-bit island image of
    # scene_config/config_heightmap.txt:84 (decoded by the library's PNG reader, itself checked against libpng in tests/test_png_io.py) and a random 16-bit image
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    pkg_ = importlib.import_module("3dworld_amd")
    emul_so = os.path.join(os.path.dirname(HERE), "emul", "libterra_emul.so")
    island = pkg_.terra.read_png(os.path.join(HERE, "heightmap_island_128.png"), lib=pkg_.terra.load_library(emul_so))
    assert island.shape == (128, 128)
    rng = np.random.default_rng(12)
    # smooth random 16-bit terrain (a sum of a few sines, kept inside (8, 248) pixel units so that erosion cannot leave [0, 256))
    yy, xx = np.mgrid[0:80, 0:96]
    hgt = 128 + 50 * np.sin(xx * 0.11 + 1.0) * np.cos(yy * 0.09) + 40 * np.sin((xx + yy) * 0.05 + 2.0) + rng.uniform(-4, 4, xx.shape)
    rand16 = np.stack([((hgt % 1.0) * 256).astype(np.uint8), hgt.astype(np.uint8)], axis=-1)
    for key, pix, scale_tz, iters in (("island128", island, pc_.ISLAND_SCALE_TZ, 6000), ("rand16", np.ascontiguousarray(rand16), (170.0, -17.0), 3000)):
        pc_.island_setup(R, pc_.island_cfg(orclib.make_config), scale_tz=scale_tz)
        v = R.heightmap_to_floats(pix)
        pc_.island_setup(R, pc_.island_cfg(orclib.make_config), (v.min(), v.max()), scale_tz)
        o, bad = R.heightmap_postprocess(pix, iters)
        assert bad == 0
        f, badf = R.heightmap_from_floats(v, 2 if pix.ndim == 3 else 1)
        assert badf == 0
        out[f"pp_{key}_in"] = pix; out[f"pp_{key}_vals"] = v; out[f"pp_{key}_out"] = o; out[f"pp_{key}_from"] = f; out[f"pp_{key}_iters"] = np.int32(iters)
        print(key, "pixels changed by postprocess_height:", int((o != pix).sum()), "of", pix.size)
    s = R.init(orclib.make_config(mesh_gen_mode=0))
    R.set_mesh_file_scale(float(sc), float(tz))  # as heightmap_proc_gen above left them
    # row f3: landscape weights texture (create_texture driver over the reference's build_arrays / eval_index / eval_mesh_sin_terms / lttex tables)
    R.set_landscape(orclib.make_landscape(grass_density=100))
    zt, _ = R.tile_create_zvals(-3, 2, 0)
    w, gb, hg = R.tile_create_weights(-3, 2, zt)
    out["tile_m3_2_weights"] = w
    out["tile_m3_2_grass_blocks"] = np.frombuffer(gb.tobytes(), np.uint8).copy()
    out["tile_m3_2_has_grass"] = np.uint8(hg)
    out["tile_terrain_params"] = np.stack([R.tile_terrain_params(-3, 2), R.tile_terrain_params(40, 41)])
    R.set_landscape(orclib.make_landscape())
    # voxels
    for mode in (0, 1, 2):
        nx, ny, nz = (40, 24, 32) if mode == 0 else (12, 10, 16)
        out[f"vox{mode}"] = R.voxel_fill(nx, ny, nz, VOX["lo"], VOX["vsz"], VOX["off"], VOX["mag"], VOX["freq"], VOX["rs1"], VOX["rs2"], mode, VOX["zscale"], 1)
    out["vox_rdata"] = R.voxel_rdata(123, 456, 1.0, 1.0)
    # glm noise, point queries, RNG, sin table
    rng = np.random.default_rng(20260923)
    pts = rng.uniform(-300, 300, (512, 3)).astype(np.float32)
    out["pts"] = pts
    out["simplex2"] = np.array([R.simplex2(x, y) for x, y, _ in pts], np.float32)
    out["perlin2"] = np.array([R.perlin2(x, y) for x, y, _ in pts], np.float32)
    out["simplex3"] = np.array([R.simplex3(x, y, z) for x, y, z in pts], np.float32)
    out["perlin3"] = np.array([R.perlin3(x, y, z) for x, y, z in pts], np.float32)
    out["sin_terms"] = np.array([R.eval_mesh_sin_terms(x, y) for x, y, _ in pts], np.float32)
    for mode in (1, 2, 4):
        out[f"noise_zval_{mode}"] = np.array([R.noise_zval(x, y, mode, 0) for x, y, _ in pts[:128]], np.float32)
    out["rand_ints_11_121"] = R.rand_ints(11, 121, 256)
    out["rand_floats_1_12345"] = R.rand_floats(1, 12345, 256)
    out["rand_uniforms_1_12345"] = R.rand_uniforms(1, 12345, 0.2, 1.0, 256)
    out["sin_table"] = R.sin_table()
    # read_mesh / write_mesh (src/mesh_gen.cpp:895-965; BASELINE config 1's `mesh_file mapx/mesh128.txt`): the reference's own file through its own reader, with the default and
    # with a non-trivial mesh_file_scale / mesh_file_tz / read_mesh_zmm; and the text its writer produces for a generated ground mesh
    import tempfile
    mesh_txt = "/root/reference/mapx/mesh128.txt"
    out["rm_mesh128_txt"] = np.frombuffer(open(mesh_txt, "rb").read(), np.uint8).copy()
    for key, (scale, tz, zmm) in (("plain", (1.0, 0.0, 0.0)), ("scaled", (2.5, -0.75, 3.0))):
        R.init(orclib.make_config(mesh_gen_mode=0))
        R.set_mesh_file_scale(scale, tz)
        ok, zz = R.read_mesh(mesh_txt, zmm)
        assert ok
        out[f"rm_{key}_mesh"] = R.ground_mesh()
        out[f"rm_{key}_zbottom_ztop"] = np.array(zz, np.float32)
        for k, v in state_dict(R.state()).items():
            if k in ("zmin", "zmax", "zmax_est", "water_plane_z"):
                out[f"rm_{key}_state_{k}"] = v
    R.set_mesh_file_scale(1.0, 0.0)
    R.init(orclib.make_config(mesh_gen_mode=0))
    with tempfile.TemporaryDirectory() as td:
        assert R.write_mesh(os.path.join(td, "m.txt"), out["m0_ground"])
        out["wm_m0_ground_txt"] = np.frombuffer(open(os.path.join(td, "m.txt"), "rb").read(), np.uint8).copy()
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), os.path.getsize(os.path.join(HERE, "reference_vectors.npz")), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
