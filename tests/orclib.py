"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

    oracle/liboracle.so            plain-C restatement (prefix "orc_")
    oracle/_ref/liboracle_ref.so   the reference's own TUs compiled in place (prefix "ref_"), only where it was built

Both export the same harness, so `Checker("orc")` and `Checker("ref")` are interchangeable.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class Config(C.Structure):
    _fields_ = [("mesh_x", C.c_int), ("mesh_y", C.c_int), ("scene_x", C.c_float), ("scene_y", C.c_float), ("scene_z", C.c_float),
                ("mesh_height", C.c_float), ("mesh_scale", C.c_float), ("mesh_seed", C.c_int), ("mesh_freq_filter", C.c_int),
                ("mesh_gen_mode", C.c_int), ("mesh_gen_shape", C.c_int), ("glaciate", C.c_int), ("custom_glaciate_exp", C.c_float),
                ("hmap", C.c_float * 14), ("erode_amount", C.c_float), ("water_h_off", C.c_float), ("water_h_off_rel", C.c_float),
                ("relh_adj_tex", C.c_float), ("ocean_wave_height", C.c_float),
                ("start_mag", C.c_float), ("start_freq", C.c_float), ("mag_mult", C.c_float), ("freq_mult", C.c_float)]


_STATE_FLOATS = ("MESH_HEIGHT DX_VAL DY_VAL DX_VAL_INV DY_VAL_INV HALF_DXY dxdy XY_SCENE_SIZE mesh_scale mesh_scale_z_inv "
                 "mesh_height_scale zmax_est zmin zmax water_plane_z glaciate_exp clip_hd1 relh_adj_tex rx ry").split()


class State(C.Structure):
    _fields_ = [("sinTable", (C.c_float * 5) * 90), ("start_eval_sin", C.c_int)] + [(n, C.c_float) for n in _STATE_FLOATS]

    def sin_table_np(self):
        return np.ctypeslib.as_array(self.sinTable).reshape(90, 5).copy()


class TileStats(C.Structure):
    _fields_ = [("sub_zmin", C.c_float * 16), ("sub_zmax", C.c_float * 16), ("mzmin", C.c_float), ("mzmax", C.c_float), ("radius", C.c_float),
                ("wx1", C.c_int), ("wy1", C.c_int), ("wx2", C.c_int), ("wy2", C.c_int)]


class ErosionStats(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("erode_steps", C.c_uint64), ("deposit_steps", C.c_uint64), ("ocean_stops", C.c_uint64),
                ("pit_stops", C.c_uint64), ("nan_droplets", C.c_uint64), ("max_steps", C.c_uint32)]


HMAP_DEFAULT = [1000.0, 0, 0, 0, 1000.0, 0, 0, 0, 0, 0, 0, 0, 0, 0]          # hmap_params_t defaults, src/mesh.h:84-88
HMAP_ISLANDS = [1000.0, 0, 0, 0, 1000.0, 0, 0, 0, 0, 5.0, 0.001, -4.0, 0, 0]  # scene_config/config.txt:76


class Landscape(C.Structure):
    """orc_landscape_t / ref_landscape_t / terra_landscape: the globals create_texture and update_terrain_params read beyond Config."""
    _fields_ = [("vegetation", C.c_float), ("temperature", C.c_float), ("biome_x_offset", C.c_float), ("mesh_scale_z", C.c_float),
                ("water_is_lava", C.c_int32), ("disable_water", C.c_int32), ("enable_terrain_env", C.c_int32),
                ("grass_density", C.c_uint32), ("num_rnd_grass_blocks", C.c_uint32)]


def make_landscape(vegetation=1.0, temperature=20.0, biome_x_offset=0.0, mesh_scale_z=1.0, water_is_lava=0, disable_water=0,
                   enable_terrain_env=1, grass_density=0, num_rnd_grass_blocks=16):
    return Landscape(vegetation, temperature, biome_x_offset, mesh_scale_z, water_is_lava, disable_water, enable_terrain_env, grass_density, num_rnd_grass_blocks)


BRUSH_DTYPE = np.dtype({"names": ["x", "y", "radius", "delta", "shape"], "formats": [np.int32, np.int32, np.uint32, np.int32, np.int16], "itemsize": 20})  # hmap_brush_t
MOD_DTYPE = np.dtype([("x", np.uint16), ("y", np.uint16), ("delta", np.int32)])  # mod_elem_t
BSHAPE_CONST_SQ, BSHAPE_CNST_CIR, BSHAPE_LINEAR, BSHAPE_QUADRATIC, BSHAPE_COSINE, BSHAPE_SINE, BSHAPE_FLAT_SQ, BSHAPE_FLAT_CIR = range(8)


def make_brushes(rows):
    """[(x, y, radius, delta, shape), ...] -> BRUSH_DTYPE array (padding bytes zero)"""
    b = np.zeros(len(rows), BRUSH_DTYPE)
    for i, r in enumerate(rows):
        b[i] = tuple(r)
    return b


def make_mods(rows):
    m = np.zeros(len(rows), MOD_DTYPE)
    for i, r in enumerate(rows):
        m[i] = tuple(r)
    return m


GRASS_BLOCK_DTYPE = np.dtype([("ix", np.uint32), ("zmin", np.float32), ("zmax", np.float32)])


def make_config(mesh_gen_mode=0, mesh_gen_shape=0, mesh_seed=1, mesh_freq_filter=0, hmap=None, glaciate=1, mesh_scale=1.0,
                mesh_height=0.7, custom_glaciate_exp=0.0, erode_amount=1.0, mesh_xy=128, scene=(4.0, 4.0, 4.0)):
    """BASELINE.md section 3 synthetic inputs (scene_config/config.txt:56-97)."""
    c = Config()
    c.mesh_x = c.mesh_y = mesh_xy
    c.scene_x, c.scene_y, c.scene_z = scene
    c.mesh_height = mesh_height
    c.mesh_scale = mesh_scale
    c.mesh_seed = mesh_seed
    c.mesh_freq_filter = mesh_freq_filter
    c.mesh_gen_mode = mesh_gen_mode
    c.mesh_gen_shape = mesh_gen_shape
    c.glaciate = glaciate
    c.custom_glaciate_exp = custom_glaciate_exp
    for i, v in enumerate(HMAP_ISLANDS if hmap is None else hmap):
        c.hmap[i] = v
    c.erode_amount = erode_amount
    c.water_h_off = c.water_h_off_rel = c.relh_adj_tex = c.ocean_wave_height = 0.0
    c.start_mag, c.start_freq, c.mag_mult, c.freq_mult = 0.02, 240.0, 2.0, 0.5
    return c


def build_oracle():
    """make oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    import fcntl
    with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lk:  # one make at a time: several xdist workers may find the targets stale at once
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)


def engine_lib(which):
    """oracle/_ref/libengine_<emul|hip>.so: the reference's own mesh_gen / heightmap / tiled_mesh ... with the INTEGRATION.md patch applied, linked against the host
    emulation library or against libterra_hip.so (oracle/Makefile: engine); None when it was not built (no /root/reference here and nothing prebuilt)"""
    p = os.path.join(ORACLE_DIR, "_ref", f"libengine_{which}.so")
    return p if os.path.exists(p) else None


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "liboracle_ref.so"))


_fp = C.POINTER(C.c_float)


class Checker:
    """Thin wrapper; kind = 'orc' (C restatement) or 'ref' (reference TUs)."""

    def __init__(self, kind="orc", path=None):
        """path: another build of the "ref" harness, e.g. oracle/_ref/libengine_emul.so (the reference's callers compiled against include/terra.h, see engine_available())"""
        self.kind = kind
        if path is None:
            path = os.path.join(ORACLE_DIR, "liboracle.so" if kind == "orc" else os.path.join("_ref", "liboracle_ref.so"))
        if not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path)
        self.p = kind + "_"
        f = self._f
        f("init", None, [C.POINTER(Config)])
        f("get_state", None, [C.POINTER(State)])
        f("set_zmax_est", None, [C.c_float])
        f("set_water_plane_z", None, [C.c_float])
        f("set_mode", None, [C.c_int, C.c_int])
        if kind == "orc":
            f("set_fused", None, [C.c_int])  # the product's TOLERANCE mode restated (oracle/terra_oracle.c: g_fused); the reference has no such mode
        f("set_start_eval_sin", None, [C.c_int])
        f("set_erode_amount", None, [C.c_float])
        f("get_ground_mesh", None, [C.c_void_p])
        f("set_ground_mesh", None, [C.c_void_p])
        f("read_mesh", C.c_int, [C.c_char_p, C.c_float, C.c_void_p])
        f("write_mesh", C.c_int, [C.c_char_p])
        f("sin_table", C.c_float, [C.c_int])
        f("num_threads", C.c_int, [])
        f("set_num_threads", None, [C.c_int])
        f("gen_grid", None, [C.c_float] * 4 + [C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p])
        f("gen_grid_ex", None, [C.c_float] * 4 + [C.c_uint, C.c_uint] + [C.c_int] * 5 + [C.c_void_p])
        f("gen_grid_rect", None, [C.c_float] * 4 + [C.c_uint, C.c_uint, C.c_int, C.c_int] + [C.c_uint] * 4 + [C.c_void_p])
        f("apply_erosion", None, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint])
        f("get_noise_zval", C.c_float, [C.c_float, C.c_float, C.c_int, C.c_int])
        f("gen_noise", C.c_float, [C.c_float, C.c_float, C.c_int, C.c_int])
        f("eval_mesh_sin_terms", C.c_float, [C.c_float, C.c_float])
        f("eval_points", None, [C.c_void_p, C.c_uint, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p])
        f("glm_simplex2", C.c_float, [C.c_float] * 2)
        f("glm_perlin2", C.c_float, [C.c_float] * 2)
        f("glm_simplex3", C.c_float, [C.c_float] * 3)
        f("glm_perlin3", C.c_float, [C.c_float] * 3)
        f("get_bare_ls_tid_is_rock", C.c_int, [C.c_float])
        f("get_max_sea_level", C.c_float, [])
        f("rand_ints", None, [C.c_long, C.c_long, C.c_int, C.c_void_p])
        f("rand_floats", None, [C.c_long, C.c_long, C.c_int, C.c_void_p])
        f("rand_uniforms", None, [C.c_long, C.c_long, C.c_float, C.c_float, C.c_int, C.c_void_p])
        f("tile_create_zvals", None, [C.c_int, C.c_int, C.c_uint, C.c_void_p, C.POINTER(TileStats)])
        f("tile_normals", C.c_float, [C.c_void_p, C.c_void_p])
        f("set_tiled_mesh_ao", None, [C.c_int])
        f("calc_mesh_shadows", None, [C.c_float] * 3 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4)
        f("tiles_mesh_shadows", None, [C.c_void_p, C.c_uint, C.c_void_p] + [C.c_float] * 3 + [C.c_void_p])
        f("hmap_set", None, [C.c_void_p, C.c_int, C.c_int, C.c_int])
        f("set_mesh_height_scales_for_zval_range", None, [C.c_float, C.c_float])
        f("get_clamped_height", C.c_float, [C.c_int, C.c_int])
        f("tile_ao_lighting", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        f("hmap_get", None, [C.c_void_p])
        f("hmap_interpolate_height", C.c_float, [C.c_float, C.c_float])
        f("hmap_get_nearest_height", C.c_float, [C.c_float, C.c_float])
        f("hmap_apply_brush", None, [C.c_void_p, C.c_int, C.c_uint])
        f("hmap_apply_mods", None, [C.c_void_p, C.c_uint])
        f("hmap_write_mod", C.c_int, [C.c_char_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint])
        f("hmap_read_mod", C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(C.c_uint), C.c_void_p, C.POINTER(C.c_uint)])
        f("hmap_read_and_apply_mod", C.c_int, [C.c_char_p])
        f("heightmap_proc_gen", None, [C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p])
        f("set_mesh_file_scale", None, [C.c_float, C.c_float])
        f("heightmap_to_floats", None, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p])
        f("heightmap_from_floats", C.c_uint, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p])
        f("heightmap_postprocess", C.c_uint, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint])
        f("export_heightmap", None, [C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        f("set_landscape", None, [C.POINTER(Landscape)])
        f("tile_terrain_params", None, [C.c_int, C.c_int, C.c_void_p])
        f("tile_create_weights", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)])
        f("quantize16", None, [C.c_void_p, C.c_size_t, C.c_void_p, _fp, _fp])
        f("voxel_fill", None, [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, _fp, _fp, _fp, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int])
        f("voxel_rdata", None, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p])
        if hasattr(self.lib, "ref_set_use_hip_terrain"):  # the engine-in-the-loop build (oracle/Makefile: engine)
            f("set_use_hip_terrain", None, [C.c_int])
            f("get_use_hip_terrain", C.c_int, [])
            f("hip_terrain_calls", C.c_uint, [])
            f("set_use_hip_proc_gen", None, [C.c_int])
        if kind == "orc":
            f("apply_erosion_stats", None, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint, C.POINTER(ErosionStats), C.c_void_p])

    def _f(self, name, restype, argtypes):
        fn = getattr(self.lib, self.p + name)
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, "_" + name, fn)

    # ---- harness
    def init(self, cfg):
        self._init(C.byref(cfg))
        return self.state()

    def state(self):
        s = State()
        self._get_state(C.byref(s))
        return s

    def set_zmax_est(self, v): self._set_zmax_est(v)
    def set_water_plane_z(self, v): self._set_water_plane_z(v)
    def set_mode(self, mode, shape=0): self._set_mode(mode, shape)
    def set_fused(self, on): self._set_fused(int(bool(on)))
    def set_start_eval_sin(self, v): self._set_start_eval_sin(v)
    def set_erode_amount(self, v): self._set_erode_amount(v)
    def set_use_hip_terrain(self, v): self._set_use_hip_terrain(int(v))
    def set_use_hip_proc_gen(self, v): self._set_use_hip_proc_gen(int(v))
    def set_num_threads(self, n): self._set_num_threads(n)
    def num_threads(self): return self._num_threads()
    def get_max_sea_level(self): return self._get_max_sea_level()

    def ground_mesh(self, n=128):
        out = np.zeros((n, n), np.float32)
        self._get_ground_mesh(out.ctypes.data)
        return out

    def read_mesh(self, path, zmm=0.0):
        """read_mesh (src/mesh_gen.cpp:895-933) -> (ok, mesh [MESH_Y, MESH_X] or None, (zbottom, ztop))"""
        zz = np.zeros(2, np.float32)
        ok = bool(self._read_mesh(os.fsencode(path), zmm, zz.ctypes.data))
        return ok, (tuple(zz) if ok else None)

    def write_mesh(self, path, mesh):
        mesh = np.ascontiguousarray(mesh, np.float32)
        self._set_ground_mesh(mesh.ctypes.data)
        return bool(self._write_mesh(os.fsencode(path)))

    def sin_table(self):
        return np.array([self._sin_table(i) for i in range(65536)], np.float32)

    def gen_grid(self, x0, y0, dx, dy, nx, ny, glaciate=1, cache_values=0, min_start_sin=0, force_sine=False, use_cache=True):
        out = np.zeros((ny, nx), np.float32)
        if force_sine or not use_cache:
            self._gen_grid_ex(x0, y0, dx, dy, nx, ny, glaciate, cache_values, int(force_sine), min_start_sin, int(use_cache), out.ctypes.data)
        else:
            self._gen_grid(x0, y0, dx, dy, nx, ny, glaciate, cache_values, min_start_sin, out.ctypes.data)
        return out

    def gen_grid_rect(self, x0, y0, dx, dy, nx, ny, rx0, ry0, rw, rh, glaciate=1, min_start_sin=0):
        """cells [ry0, ry0 + rh) x [rx0, rx0 + rw) of the nx x ny grid (tables of the whole grid, eval_index inside the rectangle only)"""
        assert rx0 + rw <= nx and ry0 + rh <= ny
        out = np.zeros((rh, rw), np.float32)
        self._gen_grid_rect(x0, y0, dx, dy, nx, ny, glaciate, min_start_sin, rx0, ry0, rw, rh, out.ctypes.data)
        return out

    def apply_erosion(self, hmap, min_zval, iters):
        assert hmap.dtype == np.float32 and hmap.flags.c_contiguous
        ys, xs = hmap.shape
        self._apply_erosion(hmap.ctypes.data, xs, ys, min_zval, iters)
        return hmap

    def apply_erosion_stats(self, hmap, min_zval, iters):
        ys, xs = hmap.shape
        st = ErosionStats()
        steps = np.zeros(iters, np.uint32)
        self._apply_erosion_stats(hmap.ctypes.data, xs, ys, min_zval, iters, C.byref(st), steps.ctypes.data)
        return st, steps

    def noise_zval(self, x, y, mode, shape=0): return self._get_noise_zval(x, y, mode, shape)
    def gen_noise(self, x, y, mode, shape=0): return self._gen_noise(x, y, mode, shape)
    def eval_mesh_sin_terms(self, x, y): return self._eval_mesh_sin_terms(x, y)

    def eval_points(self, xy, exact, xy_scale=1.0, no_xyoff=False, xoff2=0, yoff2=0):
        """eval_mesh_sin_terms_scaled (exact=False) / get_exact_zval (exact=True) for the points xy[n][2] (src/mesh_gen.cpp:807-847)"""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.empty(len(xy), np.float32)
        self._eval_points(xy.ctypes.data, len(xy), int(bool(exact)), xy_scale, int(bool(no_xyoff)), xoff2, yoff2, out.ctypes.data)
        return out
    def simplex2(self, x, y): return self._glm_simplex2(x, y)
    def perlin2(self, x, y): return self._glm_perlin2(x, y)
    def simplex3(self, x, y, z): return self._glm_simplex3(x, y, z)
    def perlin3(self, x, y, z): return self._glm_perlin3(x, y, z)
    def is_rock(self, z): return self._get_bare_ls_tid_is_rock(z)

    def rand_ints(self, s1, s2, n):
        out = np.zeros(n, np.int32); self._rand_ints(s1, s2, n, out.ctypes.data); return out

    def rand_floats(self, s1, s2, n):
        out = np.zeros(n, np.float32); self._rand_floats(s1, s2, n, out.ctypes.data); return out

    def rand_uniforms(self, s1, s2, a, b, n):
        out = np.zeros(n, np.float32); self._rand_uniforms(s1, s2, a, b, n, out.ctypes.data); return out

    def tile_create_zvals(self, tx, ty, iters_tt=0):
        z = np.zeros((130, 130), np.float32)
        st = TileStats()
        self._tile_create_zvals(tx, ty, iters_tt, z.ctypes.data, C.byref(st))
        return z, st

    def set_tiled_mesh_ao(self, v): self._set_tiled_mesh_ao(int(v))

    def calc_mesh_shadows(self, lpos, mh, sh_in_x=None, sh_in_y=None):
        """calc_mesh_shadows(LIGHT_SUN, lpos, mh, ...) -> (smask, sh_out_x, sh_out_y); sh_out arrays start at MESH_MIN_Z like tile_t does"""
        mh = np.ascontiguousarray(mh, np.float32)
        ys, xs = mh.shape
        smask = np.zeros((ys, xs), np.uint8)
        so_x = np.full(xs, -1.0e6, np.float32); so_y = np.full(ys, -1.0e6, np.float32)
        p = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).ctypes.data
        keep = [np.ascontiguousarray(a, np.float32) for a in (sh_in_x, sh_in_y) if a is not None]
        self._calc_mesh_shadows(lpos[0], lpos[1], lpos[2], mh.ctypes.data, smask.ctypes.data, xs, ys,
                                None if sh_in_x is None else keep[0].ctypes.data, None if sh_in_y is None else keep[-1].ctypes.data, so_x.ctypes.data, so_y.ctypes.data)
        return smask, so_x, so_y

    def tiles_mesh_shadows(self, tile_xy, zvals, lpos):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        z = np.ascontiguousarray(zvals, np.float32).reshape(len(txy), 130, 130)
        smask = np.zeros((len(txy), 130, 130), np.uint8)
        self._tiles_mesh_shadows(txy.ctypes.data, len(txy), z.ctypes.data, lpos[0], lpos[1], lpos[2], smask.ctypes.data)
        return smask

    def hmap_set(self, pixels, min_z=None, dz=None):
        """pixels: (h, w, 2) or (h, w) uint8 array kept alive by the caller (None: back to procedural tiles); min_z/dz: set_mesh_height_scales_for_zval_range"""
        if pixels is None:
            self._hmap_set(None, 0, 0, 0); self._hm_keep = None
            return
        assert pixels.dtype == np.uint8 and pixels.flags.c_contiguous
        self._hm_keep = pixels
        self._hmap_set(pixels.ctypes.data, pixels.shape[1], pixels.shape[0], 2 if pixels.ndim == 3 else 1)
        if min_z is not None:
            self._set_mesh_height_scales_for_zval_range(min_z, dz)

    def get_clamped_height(self, x, y): return self._get_clamped_height(x, y)

    def hmap_pixels(self):
        """the checker's current image (it keeps its own copy, edited by brushes and mods)"""
        out = np.empty_like(self._hm_keep)
        self._hmap_get(out.ctypes.data)
        return out

    def hmap_interpolate_height(self, x, y): return self._hmap_interpolate_height(x, y)
    def hmap_get_nearest_height(self, x, y): return self._hmap_get_nearest_height(x, y)

    def hmap_apply_brush(self, brush, step_sz=1, num_steps=1):
        b = np.ascontiguousarray(brush, BRUSH_DTYPE).reshape(1)
        self._hmap_apply_brush(b.ctypes.data, step_sz, num_steps)

    def hmap_apply_mods(self, mods):
        m = np.ascontiguousarray(mods, MOD_DTYPE)
        self._hmap_apply_mods(m.ctypes.data, len(m))

    def hmap_write_mod(self, fn, mods, brushes):
        m = np.ascontiguousarray(mods, MOD_DTYPE); b = np.ascontiguousarray(brushes, BRUSH_DTYPE)
        return bool(self._hmap_write_mod(str(fn).encode(), m.ctypes.data, len(m), b.ctypes.data, len(b)))

    def hmap_read_mod(self, fn):
        n, nb = C.c_uint(0), C.c_uint(0)
        if not self._hmap_read_mod(str(fn).encode(), None, C.byref(n), None, C.byref(nb)):
            return None
        m = np.zeros(n.value, MOD_DTYPE); b = np.zeros(nb.value, BRUSH_DTYPE)
        assert self._hmap_read_mod(str(fn).encode(), m.ctypes.data, C.byref(n), b.ctypes.data, C.byref(nb))
        return m, b

    def hmap_read_and_apply_mod(self, fn): return bool(self._hmap_read_and_apply_mod(str(fn).encode()))

    def heightmap_proc_gen(self, width, height, iters):
        """-> (pixels u8 [h,w,2], mesh_file_scale, mesh_file_tz)"""
        pix = np.zeros((height, width, 2), np.uint8); st = np.zeros(2, np.float32)
        self._heightmap_proc_gen(width, height, iters, pix.ctypes.data, st.ctypes.data)
        return pix, st[0], st[1]

    def set_mesh_file_scale(self, scale, tz): self._set_mesh_file_scale(scale, tz)

    def heightmap_to_floats(self, pixels):
        """heightmap_t::to_floats: (h, w) or (h, w, 2) uint8 -> (h, w) float32"""
        assert pixels.dtype == np.uint8 and pixels.flags.c_contiguous
        out = np.zeros(pixels.shape[:2], np.float32)
        self._heightmap_to_floats(pixels.ctypes.data, pixels.shape[1], pixels.shape[0], 2 if pixels.ndim == 3 else 1, out.ctypes.data)
        return out

    def heightmap_from_floats(self, vals, ncolors):
        """heightmap_t::from_floats with the scale in force -> (pixels, number of values outside [0, 256))"""
        vals = np.ascontiguousarray(vals, np.float32)
        h, w = vals.shape
        pix = np.zeros((h, w, 2) if ncolors == 2 else (h, w), np.uint8)
        bad = self._heightmap_from_floats(vals.ctypes.data, w, h, ncolors, pix.ctypes.data)
        return pix, bad

    def heightmap_postprocess(self, pixels, iters_tt):
        """heightmap_t::postprocess_height on a copy -> (pixels, number of values outside [0, 256))"""
        pix = np.ascontiguousarray(pixels, np.uint8).copy()
        bad = self._heightmap_postprocess(pix.ctypes.data, pix.shape[1], pix.shape[0], 2 if pix.ndim == 3 else 1, iters_tt)
        return pix, bad

    def export_heightmap(self, xstart, ystart, width, height):
        """-> (pixels u8 [h,w,2], min_z, dz)"""
        pix = np.zeros((height, width, 2), np.uint8); r = np.zeros(2, np.float32)
        self._export_heightmap(xstart, ystart, width, height, pix.ctypes.data, r.ctypes.data)
        return pix, r[0], r[1]

    def tile_ao_lighting(self, tx, ty, zvals):
        assert zvals.dtype == np.float32 and zvals.shape == (130, 130) and zvals.flags.c_contiguous
        ao = np.zeros((129, 129), np.uint8)
        self._tile_ao_lighting(tx, ty, zvals.ctypes.data, ao.ctypes.data)
        return ao

    def set_landscape(self, ls): self._set_landscape(C.byref(ls))

    def tile_terrain_params(self, tx, ty):
        out = np.zeros((2, 2, 3), np.float32)
        self._tile_terrain_params(tx, ty, out.ctypes.data)
        return out

    def tile_create_weights(self, tx, ty, zvals):
        """-> (weights u8[129,129,4], grass blocks [32,32] of GRASS_BLOCK_DTYPE, has_any_grass)"""
        zvals = np.ascontiguousarray(zvals, np.float32)
        w = np.zeros((129, 129, 4), np.uint8); gb = np.zeros((32, 32), GRASS_BLOCK_DTYPE); hg = C.c_int(0)
        self._tile_create_weights(tx, ty, zvals.ctypes.data, w.ctypes.data, gb.ctypes.data, C.byref(hg))
        return w, gb, bool(hg.value)

    def tile_normals(self, zvals):
        rgba = np.zeros((129, 129, 4), np.uint8)
        mnz = self._tile_normals(np.ascontiguousarray(zvals, np.float32).ctypes.data, rgba.ctypes.data)
        return rgba, mnz

    def quantize16(self, vals):
        vals = np.ascontiguousarray(vals, np.float32)
        out = np.zeros(vals.size * 2, np.uint8)
        mn, dz = C.c_float(), C.c_float()
        self._quantize16(vals.ctypes.data, vals.size, out.ctypes.data, C.byref(mn), C.byref(dz))
        return out, mn.value, dz.value

    def voxel_fill(self, nx, ny, nz, lo_pos, vsz, offset, mag, freq, rseed1, rseed2, gen_mode, zscale, normalize):
        out = np.zeros((ny, nx, nz), np.float32)
        a = lambda v: (C.c_float * 3)(*v)
        self._voxel_fill(out.ctypes.data, nx, ny, nz, a(lo_pos), a(vsz), a(offset), mag, freq, rseed1, rseed2, gen_mode, zscale, normalize)
        return out

    def voxel_rdata(self, rseed1, rseed2, mag, freq):
        out = np.zeros(420, np.float32)
        self._voxel_rdata(rseed1, rseed2, mag, freq, out.ctypes.data)
        return out


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)
        i = tuple(idx[0])
        raise AssertionError(f"{what}: {len(idx)} of {a.size} values differ; first at {i}: {a[i]!r} vs {b[i]!r}; max abs diff {np.nanmax(np.abs(a - b))}")
