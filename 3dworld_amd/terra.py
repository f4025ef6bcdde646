"""ctypes bindings for include/terra.h (libterra_hip.so).  Host glue only -- no arithmetic happens in Python."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

GEN_GLACIATE, GEN_FORCE_SINE, GEN_NO_WAIT, GEN_CACHE_VALUES, GEN_FUSED, GEN_FAST = 1, 2, 4, 8, 16, 32
ERODE_SERIAL, ERODE_MINZ_IS_MIN, ERODE_SERIAL_WAVE = 1, 2, 4
MGEN_SINE, MGEN_SIMPLEX, MGEN_PERLIN, MGEN_SIMPLEX_GPU, MGEN_DWARP_GPU = range(5)


class TerraError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"terra error {code}: {msg}")
        self.code = code


class Config(C.Structure):  # terra_config
    _fields_ = [("mesh_x", C.c_int32), ("mesh_y", C.c_int32), ("scene_x", C.c_float), ("scene_y", C.c_float), ("scene_z", C.c_float),
                ("mesh_height", C.c_float), ("mesh_scale", C.c_float), ("mesh_seed", C.c_int32), ("mesh_freq_filter", C.c_int32),
                ("mesh_gen_mode", C.c_int32), ("mesh_gen_shape", C.c_int32), ("glaciate", C.c_int32), ("custom_glaciate_exp", C.c_float),
                ("hmap", C.c_float * 14), ("erode_amount", C.c_float), ("water_h_off", C.c_float), ("water_h_off_rel", C.c_float),
                ("relh_adj_tex", C.c_float), ("ocean_wave_height", C.c_float),
                ("start_mag", C.c_float), ("start_freq", C.c_float), ("mag_mult", C.c_float), ("freq_mult", C.c_float)]


_STATE_FLOATS = ("MESH_HEIGHT DX_VAL DY_VAL DX_VAL_INV DY_VAL_INV HALF_DXY dxdy XY_SCENE_SIZE mesh_scale mesh_scale_z_inv "
                 "mesh_height_scale zmax_est zmin zmax water_plane_z glaciate_exp clip_hd1 relh_adj_tex rx ry").split()


class State(C.Structure):  # terra_state
    _fields_ = [("sinTable", (C.c_float * 5) * 90), ("start_eval_sin", C.c_int32)] + [(n, C.c_float) for n in _STATE_FLOATS]


class TileStats(C.Structure):  # terra_tile_stats
    _fields_ = [("sub_zmin", C.c_float * 16), ("sub_zmax", C.c_float * 16), ("mzmin", C.c_float), ("mzmax", C.c_float), ("radius", C.c_float),
                ("wx1", C.c_int32), ("wy1", C.c_int32), ("wx2", C.c_int32), ("wy2", C.c_int32)]


class Landscape(C.Structure):  # terra_landscape
    _fields_ = [("vegetation", C.c_float), ("temperature", C.c_float), ("biome_x_offset", C.c_float), ("mesh_scale_z", C.c_float),
                ("water_is_lava", C.c_int32), ("disable_water", C.c_int32), ("enable_terrain_env", C.c_int32),
                ("grass_density", C.c_uint32), ("num_rnd_grass_blocks", C.c_uint32)]


def make_landscape(vegetation=1.0, temperature=20.0, biome_x_offset=0.0, mesh_scale_z=1.0, water_is_lava=0, disable_water=0,
                   enable_terrain_env=1, grass_density=0, num_rnd_grass_blocks=16):
    """terra_landscape with the reference's defaults."""
    return Landscape(vegetation, temperature, biome_x_offset, mesh_scale_z, water_is_lava, disable_water, enable_terrain_env, grass_density, num_rnd_grass_blocks)


BRUSH_DTYPE = np.dtype({"names": ["x", "y", "radius", "delta", "shape"], "formats": [np.int32, np.int32, np.uint32, np.int32, np.int16], "itemsize": 20})  # terra_hmap_brush
MOD_DTYPE = np.dtype([("x", np.uint16), ("y", np.uint16), ("delta", np.int32)])  # terra_hmap_mod
GRASS_BLOCK_DTYPE = np.dtype([("ix", np.uint32), ("zmin", np.float32), ("zmax", np.float32)])  # terra_grass_block


class ErosionReport(C.Structure):  # terra_erosion_report
    _fields_ = [("droplets", C.c_uint32), ("windows", C.c_uint32), ("rounds", C.c_uint32), ("traces", C.c_uint32),
                ("serial_fallbacks", C.c_uint32), ("nan_droplets", C.c_uint32), ("steps", C.c_uint64), ("traced_steps", C.c_uint64),
                ("retraces_same", C.c_uint64), ("checkpoint_resumes", C.c_uint64), ("checkpoint_steps_saved", C.c_uint64),
                ("window_shifts", C.c_uint64), ("critical_steps", C.c_uint64), ("critical_shifts", C.c_uint64),
                ("clk_wave", C.c_uint64), ("clk_init", C.c_uint64), ("clk_shift", C.c_uint64), ("clk_tail", C.c_uint64), ("clk_critical", C.c_uint64),
                ("clk_shift_flush", C.c_uint64), ("clk_shift_prep", C.c_uint64), ("clk_shift_load", C.c_uint64),
                ("crit_clk_flush", C.c_uint64), ("crit_clk_load", C.c_uint64), ("crit_clk_prep", C.c_uint64),
                ("crit_clk_shift", C.c_uint64), ("crit_clk_edge", C.c_uint64), ("crit_steps_own", C.c_uint64),
                ("sparse_droplets", C.c_uint64), ("sparse_retraces", C.c_uint64), ("sparse_probe_only", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


HMAP_ISLANDS = [1000.0, 0, 0, 0, 1000.0, 0, 0, 0, 0, 5.0, 0.001, -4.0, 0, 0]  # hmap_params_t defaults + scene_config/config.txt:76


def make_config(mesh_gen_mode=0, mesh_gen_shape=0, mesh_seed=1, mesh_freq_filter=0, hmap=None, glaciate=1, mesh_scale=1.0,
                mesh_height=0.7, custom_glaciate_exp=0.0, erode_amount=1.0, mesh_xy=128, scene=(4.0, 4.0, 4.0)):
    """The synthetic scene of BASELINE.md section 3 (scene_config/config.txt:56-97)."""
    c = Config()
    c.mesh_x = c.mesh_y = mesh_xy
    c.scene_x, c.scene_y, c.scene_z = scene
    c.mesh_height, c.mesh_scale = mesh_height, mesh_scale
    c.mesh_seed, c.mesh_freq_filter, c.mesh_gen_mode, c.mesh_gen_shape, c.glaciate = mesh_seed, mesh_freq_filter, mesh_gen_mode, mesh_gen_shape, glaciate
    c.custom_glaciate_exp = custom_glaciate_exp
    for i, v in enumerate(HMAP_ISLANDS if hmap is None else hmap):
        c.hmap[i] = v
    c.erode_amount = erode_amount
    c.water_h_off = c.water_h_off_rel = c.relh_adj_tex = c.ocean_wave_height = 0.0
    c.start_mag, c.start_freq, c.mag_mult, c.freq_mult = 0.02, 240.0, 2.0, 0.5
    return c


def default_lib_path():
    """3dworld_amd/libterra_hip.so; TERRA_LIB names another build of the same library (A/B experiments with tools/ab_build.sh -- never set in tests or bench runs)"""
    return os.environ.get("TERRA_LIB") or os.path.join(HERE, "libterra_hip.so")


_vp, _f, _u32, _i32, _sz = C.c_void_p, C.c_float, C.c_uint32, C.c_int, C.c_size_t
_f3 = C.POINTER(C.c_float)

_PROTOS = {
    "terra_last_error": (C.c_char_p, []),
    "terra_device_count": (_i32, []),
    "terra_create": (_i32, [C.POINTER(_vp), _i32]),
    "terra_destroy": (None, [_vp]),
    "terra_set_stream": (_i32, [_vp, _vp]),
    "terra_synchronize": (_i32, [_vp]),
    "terra_set_option": (_i32, [_vp, C.c_char_p, C.c_char_p]),
    "terra_eval_points": (_i32, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_float, _i32, _i32, _i32, _vp]),
    "terra_eval_points_dev": (_i32, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_float, _i32, _i32, _i32, _vp]),
    "terra_host_alloc": (_vp, [_sz]),
    "terra_host_free": (None, [_vp]),
    "terra_download_async": (_i32, [_vp, _vp, _vp, _sz]),
    "terra_download_wait": (_i32, [_vp]),
    "terra_init_scene": (_i32, [_vp, C.POINTER(Config)]),
    "terra_set_config": (_i32, [_vp, C.POINTER(Config)]),
    "terra_get_state": (_i32, [_vp, C.POINTER(State)]),
    "terra_set_state": (_i32, [_vp, C.POINTER(State)]),
    "terra_set_mode": (_i32, [_vp, _i32, _i32]),
    "terra_set_zmax_est": (_i32, [_vp, _f]),
    "terra_set_water_plane_z": (_i32, [_vp, _f]),
    "terra_set_start_eval_sin": (_i32, [_vp, _i32]),
    "terra_set_erode_amount": (_i32, [_vp, _f]),
    "terra_get_max_sea_level": (_f, [_vp]),
    "terra_gen_create": (_i32, [_vp, C.POINTER(_vp)]),
    "terra_gen_destroy": (None, [_vp]),
    "terra_gen_build_arrays": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32]),
    "terra_gen_enable_glaciate": (_i32, [_vp]),
    "terra_gen_is_running": (_i32, [_vp]),
    "terra_gen_collect": (_i32, [_vp, _vp]),
    "terra_gen_eval_index": (_f, [_vp, _u32, _u32, _i32, _i32]),
    "terra_gen_grid_rows_minmax_dev": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32, _u32, _u32, _vp, _vp, _vp]),
    "terra_gen_device_values": (_vp, [_vp]),
    "terra_gen_grid_dev": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32, _vp]),
    "terra_gen_grid": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32, _vp]),
    "terra_gen_grid_minmax_dev": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32, _vp, _f3, _f3]),
    "terra_eval_mesh_sin_terms": (_i32, [_vp, _f, _f, _f3]),
    "terra_glaciate_mesh_dev": (_i32, [_vp, _vp, _u32, _u32, _i32, _i32, _vp]),
    "terra_apply_erosion_dev": (_i32, [_vp, _vp, _i32, _i32, _f, _u32, _u32]),
    "terra_apply_erosion": (_i32, [_vp, _vp, _i32, _i32, _f, _u32]),
    "terra_release_scratch": (_i32, [_vp]),
    "terra_event_create": (_i32, [_vp, C.POINTER(_vp)]),
    "terra_event_record": (_i32, [_vp, _vp]),
    "terra_event_wait": (_i32, [_vp, _vp]),
    "terra_event_synchronize": (_i32, [_vp]),
    "terra_event_destroy": (None, [_vp]),
    "terra_apply_erosion_devmin_dev": (_i32, [_vp, _vp, _i32, _i32, _vp, _u32, _u32]),
    "terra_erosion_shard_arena_bytes": (C.c_size_t, [_vp, _u32]),
    "terra_erosion_shard_trace_dev": (_i32, [_vp, _vp, _i32, _i32, _u32, _u32, _u32, _vp]),
    "terra_erosion_shard_finish_dev": (_i32, [_vp, _vp, _i32, _i32, _vp, _u32, _u32, _u32, _u32, C.POINTER(_u32), _vp, C.c_size_t]),
    "terra_gen_grid_minmax_async_dev": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32, _vp, _vp]),
    "terra_gen_grid_rows_minmax_async_dev": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32, _u32, _u32, _vp, _vp]),
    "terra_get_erosion_report": (_i32, [_vp, C.POINTER(ErosionReport)]),
    "terra_set_erosion_tuning": (_i32, [_vp, _u32, _u32, _u32]),
    "terra_set_erosion_slice_steps": (_i32, [_vp, _u32]),
    "terra_set_tiled_mesh_ao": (_i32, [_vp, _i32]),
    "terra_heightmap_write_png": (_i32, [C.c_char_p, _vp, _u32, _u32, _i32]),
    "terra_heightmap_read_png": (_i32, [C.c_char_p, _i32, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_i32), _vp, C.c_size_t]),
    "terra_read_mesh": (_i32, [_vp, C.c_char_p, _f, _vp, _u32, _u32, _vp]),
    "terra_write_mesh": (_i32, [C.c_char_p, _vp, _u32, _u32]),
    "terra_tiles_mesh_shadows_dev": (_i32, [_vp, _vp, _u32, _vp, _f3, _vp]),
    "terra_tiles_mesh_shadows": (_i32, [_vp, _vp, _u32, _vp, _f3, _vp]),
    "terra_tiles_mesh_shadows_halo_dev": (_i32, [_vp, _vp, _u32, _vp, _f3, _vp, _vp, _vp, _vp]),
    "terra_hmap_set_dev": (_i32, [_vp, _vp, _i32, _i32, _i32]),
    "terra_set_mesh_height_scales_for_zval_range": (_i32, [_vp, _f, _f]),
    "terra_set_mesh_file_scale": (_i32, [_vp, _f, _f]),
    "terra_get_mesh_file_scale": (_i32, [_vp, _f3, _f3]),
    "terra_heightmap_to_floats_dev": (_i32, [_vp, _vp, _u32, _u32, _i32, _vp]),
    "terra_heightmap_from_floats_dev": (_i32, [_vp, _vp, _u32, _u32, _i32, _vp, C.POINTER(_u32)]),
    "terra_heightmap_postprocess_dev": (_i32, [_vp, _vp, _u32, _u32, _i32, _u32, _vp, C.POINTER(_u32)]),
    "terra_hmap_apply_brushes_dev": (_i32, [_vp, _vp, _u32, _i32, _u32]),
    "terra_hmap_apply_mods_dev": (_i32, [_vp, _vp, _u32]),
    "terra_hmap_read_and_apply_mod_dev": (_i32, [_vp, C.c_char_p]),
    "terra_hmap_write_mod": (_i32, [C.c_char_p, _vp, _u32, _vp, _u32]),
    "terra_hmap_read_mod": (_i32, [C.c_char_p, _vp, _u32, _vp, _vp, _u32, _vp]),
    "terra_export_heightmap_dev": (_i32, [_vp, _f, _f, _u32, _u32, _vp, _vp, _vp]),
    "terra_write_map_mode_heightmap_image": (_i32, [_vp, C.c_char_p, _f, _f, _u32, _u32]),
    "terra_set_landscape": (_i32, [_vp, _vp]),
    "terra_get_landscape": (_i32, [_vp, _vp]),
    "terra_tiles_terrain_params": (_i32, [_vp, _vp, _u32, _vp]),
    "terra_tiles_create_weights_dev": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "terra_tiles_create_weights": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "terra_tiles_ao_lighting_dev": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "terra_tiles_ao_lighting": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "terra_heightmap_proc_gen": (_i32, [_vp, _u32, _u32, _u32, _vp, _f3]),
    "terra_heightmap_proc_gen_dev": (_i32, [_vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "terra_minmax_dev": (_i32, [_vp, _vp, _sz, _f3, _f3]),
    "terra_quantize16_dev": (_i32, [_vp, _vp, _sz, _f, _f, _vp]),
    "terra_tiles_create_zvals_dev": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "terra_tiles_post_dev": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "terra_tiles_post": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "terra_tiles_create_zvals": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "terra_selftest_hot_sqrt": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_uint64)]),
    "terra_voxel_fill_dev": (_i32, [_vp, _vp, _u32, _u32, _u32, _f3, _f3, _f3, _f, _f, _i32, _i32, _i32, _f, _i32]),
    "terra_voxel_fill_slab_dev": (_i32, [_vp, _vp, _u32, _u32, _u32, _f3, _f3, _f3, _f, _f, _i32, _i32, _i32, _f, _i32, _u32, _u32]),
    "terra_voxel_fill": (_i32, [_vp, _vp, _u32, _u32, _u32, _f3, _f3, _f3, _f, _f, _i32, _i32, _i32, _f, _i32]),
    "terra_multi_create": (_i32, [C.POINTER(_vp), C.POINTER(_i32), _u32]),
    "terra_multi_destroy": (None, [_vp]),
    "terra_multi_size": (_u32, [_vp]),
    "terra_multi_ctx": (_vp, [_vp, _u32]),
    "terra_multi_partition": (None, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "terra_multi_foreach": (_i32, [_vp, _vp, _vp]),
    "terra_multi_synchronize": (_i32, [_vp]),
    "terra_multi_init_scene": (_i32, [_vp, C.POINTER(Config)]),
    "terra_multi_tiles_create_zvals_dev": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "terra_multi_tiles_create_zvals": (_i32, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "terra_multi_gen_grid_rows_dev": (_i32, [_vp, _f, _f, _f, _f, _u32, _u32, _u32, _i32, _vp, _f3, _f3]),
    "terra_multi_voxel_fill_dev": (_i32, [_vp, _vp, _u32, _u32, _u32, _f3, _f3, _f3, _f, _f, _i32, _i32, _i32, _f, _i32]),
    "terra_multi_tiles_mesh_shadows": (_i32, [_vp, _vp, _u32, _vp, _f3, _vp]),
    "terra_multi_shadow_layout": (_i32, [_vp, _vp, _u32, _f3, _vp, _vp, _vp]),
    "terra_multi_tiles_mesh_shadows_dev": (_i32, [_vp, _vp, _u32, _vp, _f3, _vp]),
    "terra_dgrid_granularity": (_sz, [_vp]),
    "terra_dgrid_create": (_i32, [_vp, _u32, C.POINTER(_sz), _u32, C.POINTER(_vp)]),
    "terra_dgrid_export_fd": (_i32, [_vp, C.POINTER(_i32)]),
    "terra_dgrid_import_fd": (_i32, [_vp, _u32, _i32]),
    "terra_dgrid_map": (_i32, [_vp, C.POINTER(_vp)]),
    "terra_dgrid_destroy": (None, [_vp]),
    "terra_multi_dgrid_create": (_i32, [_vp, C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_vp)]),
    "terra_tiles_mesh_shadows_edges_dev": (_i32, [_vp, _vp, _u32, _vp, _f3, _vp, _vp, _vp, _vp]),
    "terra_malloc": (_i32, [_vp, C.POINTER(_vp), _sz]),
    "terra_free": (_i32, [_vp, _vp]),
    "terra_memcpy_h2d": (_i32, [_vp, _vp, _vp, _sz]),
    "terra_memcpy_d2h": (_i32, [_vp, _vp, _vp, _sz]),
    "terra_timer_start": (_i32, [_vp]),
    "terra_timer_stop": (_i32, [_vp, _f3]),
}
EXPORTED_SYMBOLS = sorted(_PROTOS)


def load_library(path=None):
    path = path or default_lib_path()
    if not os.path.exists(path):
        raise TerraError(-4, f"{path} not found: build it with __graft_entry__.build() (hipcc, gfx950). There is no CPU fall-back.")
    if os.path.abspath(path) == os.path.join(HERE, "libterra_hip.so"):  # the shipped library must be the tree's sources (a .so left by an experiment computes something else)
        from . import build as _build
        if _build.needs_build():
            raise TerraError(-4, f"{path} was not built from the sources in 3dworld_amd/csrc: run __graft_entry__.build()")
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # raises AttributeError if the ABI is incomplete
        fn.restype, fn.argtypes = res, args
    return lib


def read_png(path, allow_two_byte_grayscale=True, lib=None):
    """terra_heightmap_read_png without a context (host-only entry point): (h, w) or (h, w, 2) uint8, rows flipped like texture_t::load_png"""
    lib = lib or load_library()
    w, h, nc = _u32(), _u32(), _i32()
    args = (str(path).encode(), int(allow_two_byte_grayscale), C.byref(w), C.byref(h), C.byref(nc))
    if lib.terra_heightmap_read_png(*args, None, 0) < 0:
        raise TerraError(-1, lib.terra_last_error().decode())
    out = np.empty((h.value, w.value, 2) if nc.value == 2 else (h.value, w.value), np.uint8)
    if lib.terra_heightmap_read_png(*args, out.ctypes.data, out.nbytes) < 0:
        raise TerraError(-1, lib.terra_last_error().decode())
    return out


class DeviceBuffer:
    """A device allocation owned through the C ABI (for callers that have no torch tensor to hand over)."""

    def __init__(self, terra, nbytes):
        self.t, self.nbytes = terra, nbytes
        p = _vp()
        terra._ck(terra.lib.terra_malloc(terra.ctx, C.byref(p), nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.t._ck(self.t.lib.terra_memcpy_h2d(self.t.ctx, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.t._ck(self.t.lib.terra_memcpy_d2h(self.t.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.t.lib.terra_free(self.t.ctx, self.ptr)
            self.ptr = None


class PinnedArray:
    """A numpy array over pinned host memory from terra_host_alloc: the DMA target of downloads (no staging copy).  free() it explicitly."""

    def __init__(self, lib, shape, dtype=np.float32):
        self.lib = lib
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        lib.terra_host_alloc.restype = _vp
        self.ptr = lib.terra_host_alloc(max(n, 1))
        if not self.ptr:
            raise TerraError(-1, lib.terra_last_error().decode())
        self.array = np.frombuffer((C.c_uint8 * n).from_address(self.ptr), dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.lib.terra_host_free(self.ptr)
            self.ptr = None


# The library reads no environment variable.  Tests and tools keep their TERRA_* experiment variables: this table turns them into terra_set_option calls
# (variable -> (option key, value when the variable is unset)); Terra.apply_env_options() is called at construction and by whoever changes a variable afterwards.
ENV_OPTIONS = {
    "TERRA_GEN_FUSED": ("gen.fused", "0"), "TERRA_SIMPLE_KERNELS": ("kernels.simple", "0"), "TERRA_GRAPHS": ("graphs", "1"),
    "TERRA_SG_KC": ("sg.kc", "27"), "TERRA_SG_KC_TILES": ("sg.kc_tiles", "27"), "TERRA_SG_ROWGROUP": ("sg.rowgroup", "4"), "TERRA_TILE_EROSION": ("tile_erosion", "lds"),
    "TERRA_WEIGHTS_SIMPLE": ("weights.simple", "0"), "TERRA_VOXELS_COLS": ("voxels.cols", "1"), "TERRA_AO_BANDS": ("ao.bands", "1"), "TERRA_AO_WHOLE": ("ao.whole", "1"), "TERRA_SHADOWS_LEVELS": ("shadows.levels", "0"),
    "TERRA_ERO_SPARSE": ("ero.sparse", "auto"), "TERRA_ERO_SPARSE_RETRACES": ("ero.sparse_retraces", "-1"), "TERRA_ERO_LEAD": ("ero.lead", "2"), "TERRA_ERO_BATCH": ("ero.batch", "0"), "TERRA_ERO_FUSE": ("ero.fuse", "3"),
    "TERRA_ERO_LIVE": ("ero.live", "1"), "TERRA_ERO_DIAG": ("ero.diag", "0"), "TERRA_ERO_CK": ("ero.ck", "default"), "TERRA_ERO_NEAR": ("ero.near", "default"),
    "TERRA_ERO_MEM_BUDGET": ("ero.mem_budget", "-1"),
}


class Terra:
    """One terra_ctx (one GPU, one stream)."""

    def __init__(self, device=0, lib_path=None, env_options=True):
        self.lib = load_library(lib_path)
        ctx = _vp()
        rc = self.lib.terra_create(C.byref(ctx), device)
        if rc != 0:
            raise TerraError(rc, self.lib.terra_last_error().decode())
        self.ctx = ctx
        self._env_applied = {}
        if env_options:
            self.apply_env_options()

    def set_option(self, key, value):
        """terra_set_option: every behaviour switch of the library (include/terra.h lists the keys)"""
        self._ck(self.lib.terra_set_option(self.ctx, str(key).encode(), str(value).encode()))

    def apply_env_options(self):
        """experiment knobs of tests / tools: TERRA_* variables -> options (a variable that is unset puts its option back to the default)"""
        for var, (key, default) in ENV_OPTIONS.items():
            val = os.environ.get(var, default)
            if self._env_applied.get(key, default) != val:
                self.set_option(key, val)
                self._env_applied[key] = val

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.terra_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise TerraError(rc, self.lib.terra_last_error().decode())
        return rc

    # ---- scene
    def init_scene(self, cfg):
        self._ck(self.lib.terra_init_scene(self.ctx, C.byref(cfg)))
        return self.state()

    def state(self):
        s = State()
        self._ck(self.lib.terra_get_state(self.ctx, C.byref(s)))
        return s

    def set_state(self, s): self._ck(self.lib.terra_set_state(self.ctx, C.byref(s)))
    def set_config(self, cfg): self._ck(self.lib.terra_set_config(self.ctx, C.byref(cfg)))
    def set_mode(self, mode, shape=0): self._ck(self.lib.terra_set_mode(self.ctx, mode, shape))
    def set_zmax_est(self, v): self._ck(self.lib.terra_set_zmax_est(self.ctx, v))
    def set_water_plane_z(self, v): self._ck(self.lib.terra_set_water_plane_z(self.ctx, v))
    def set_start_eval_sin(self, v): self._ck(self.lib.terra_set_start_eval_sin(self.ctx, v))
    def set_erode_amount(self, v): self._ck(self.lib.terra_set_erode_amount(self.ctx, v))
    def set_stream(self, stream_ptr): self._ck(self.lib.terra_set_stream(self.ctx, stream_ptr))
    def release_scratch(self): self._ck(self.lib.terra_release_scratch(self.ctx))
    def synchronize(self): self._ck(self.lib.terra_synchronize(self.ctx))
    def pinned(self, shape, dtype=np.float32): return PinnedArray(self.lib, shape, dtype)

    def download_async(self, dev_ptr, host_arr):
        """device -> host array (numpy, C-contiguous; pageable or PinnedArray.array) behind everything enqueued so far; returns at once -- complete after download_wait()"""
        assert host_arr.flags["C_CONTIGUOUS"]
        self._ck(self.lib.terra_download_async(self.ctx, dev_ptr, host_arr.ctypes.data, host_arr.nbytes))

    def download_wait(self): self._ck(self.lib.terra_download_wait(self.ctx))
    def max_sea_level(self): return self.lib.terra_get_max_sea_level(self.ctx)
    def alloc(self, nbytes): return DeviceBuffer(self, nbytes)
    def timer_start(self): self._ck(self.lib.terra_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.lib.terra_timer_stop(self.ctx, C.byref(ms)))
        return ms.value

    # ---- host-buffer drop-ins
    def gen_grid(self, x0, y0, dx, dy, nx, ny, flags=GEN_GLACIATE, min_start_sin=0):
        out = np.empty((ny, nx), np.float32)
        self._ck(self.lib.terra_gen_grid(self.ctx, x0, y0, dx, dy, nx, ny, flags, min_start_sin, out.ctypes.data))
        return out

    def apply_erosion(self, hmap, min_zval, iters):
        assert hmap.dtype == np.float32 and hmap.flags.c_contiguous
        ys, xs = hmap.shape
        self._ck(self.lib.terra_apply_erosion(self.ctx, hmap.ctypes.data, xs, ys, min_zval, iters))
        return hmap

    def set_erosion_tuning(self, window=0, log_capacity_log2=0, block_list_capacity=0):
        self._ck(self.lib.terra_set_erosion_tuning(self.ctx, window, log_capacity_log2, block_list_capacity))

    def set_erosion_slice_steps(self, steps):
        self._ck(self.lib.terra_set_erosion_slice_steps(self.ctx, steps))

    def erosion_report(self):
        r = ErosionReport()
        self._ck(self.lib.terra_get_erosion_report(self.ctx, C.byref(r)))
        return r

    def tiles_create_zvals(self, tile_xy, iters_tt=0, stats=True, normals=True):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        z = np.empty((n, 130, 130), np.float32)
        st = (TileStats * n)() if stats else None
        nm = np.empty((n, 129, 129, 4), np.uint8) if normals else None
        mnz = np.empty(n, np.float32) if normals else None
        self._ck(self.lib.terra_tiles_create_zvals(self.ctx, txy.ctypes.data, n, iters_tt, z.ctypes.data,
                                                    C.addressof(st) if stats else None, nm.ctypes.data if normals else None,
                                                    mnz.ctypes.data if normals else None))
        return z, st, nm, mnz

    def selftest_hot_sqrt(self, stride=1):
        """disagreements of the droplet step's square roots with sqrtf / the correctly rounded root over every stride-th fp32 bit pattern (must be 0)"""
        n = C.c_uint64(0)
        self._ck(self.lib.terra_selftest_hot_sqrt(self.ctx, stride, C.byref(n)))
        return int(n.value)

    def hmap_set_dev(self, ptr, width=0, height=0, ncolors=2, min_z=None, dz=None):
        """heightmap texture for the tile path (device pointer kept, not copied; None / 0 switches it off)"""
        self._ck(self.lib.terra_hmap_set_dev(self.ctx, ptr, width, height, ncolors))
        if ptr and min_z is not None:
            self._ck(self.lib.terra_set_mesh_height_scales_for_zval_range(self.ctx, min_z, dz))

    def set_mesh_file_scale(self, scale, tz):
        """config `mh_filename <png> <scale> <tz>`: pixel value -> height"""
        self._ck(self.lib.terra_set_mesh_file_scale(self.ctx, scale, tz))

    def get_mesh_file_scale(self):
        a, b = C.c_float(), C.c_float()
        self._ck(self.lib.terra_get_mesh_file_scale(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def heightmap_to_floats_dev(self, pix_ptr, width, height, ncolors, vals_ptr):
        self._ck(self.lib.terra_heightmap_to_floats_dev(self.ctx, pix_ptr, width, height, ncolors, vals_ptr))

    def heightmap_from_floats_dev(self, vals_ptr, width, height, ncolors, pix_ptr):
        """-> number of values outside [0, 256) pixel units (the reference asserts on those)"""
        bad = _u32()
        self._ck(self.lib.terra_heightmap_from_floats_dev(self.ctx, vals_ptr, width, height, ncolors, pix_ptr, C.byref(bad)))
        return bad.value

    def heightmap_postprocess_dev(self, pix_ptr, width, height, ncolors, erosion_iters_tt, vals_ptr=None):
        """heightmap_t::postprocess_height in place on the device image -> number of out-of-range values"""
        bad = _u32()
        self._ck(self.lib.terra_heightmap_postprocess_dev(self.ctx, pix_ptr, width, height, ncolors, erosion_iters_tt, vals_ptr, C.byref(bad)))
        return bad.value

    def tiles_mesh_shadows(self, tile_xy, zvals, light_pos):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        z = np.ascontiguousarray(zvals, np.float32).reshape(n, 130, 130)
        sm = np.empty((n, 130, 130), np.uint8)
        lp = (C.c_float * 3)(*light_pos)
        self._ck(self.lib.terra_tiles_mesh_shadows(self.ctx, txy.ctypes.data, n, z.ctypes.data, lp, sm.ctypes.data))
        return sm

    def tiles_mesh_shadows_dev(self, tile_xy, z_ptr, light_pos, smask_ptr):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        lp = (C.c_float * 3)(*light_pos)
        self._ck(self.lib.terra_tiles_mesh_shadows_dev(self.ctx, txy.ctypes.data, len(txy), z_ptr, lp, smask_ptr))

    def tiles_mesh_shadows_halo_dev(self, tile_xy, z_ptr, light_pos, smask_ptr, edge_in=None, edge_in_present=None, want_edge_out=True):
        """part of a terrain: edge_in (n,2,130) float32 + edge_in_present (n,2) uint8 on the host, returns edge_out (n,2,130) or None"""
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        lp = (C.c_float * 3)(*light_pos)
        ei = np.ascontiguousarray(edge_in, np.float32).reshape(n, 2, 130) if edge_in is not None else None
        ep = np.ascontiguousarray(edge_in_present, np.uint8).reshape(n, 2) if edge_in is not None else None
        eo = np.empty((n, 2, 130), np.float32) if want_edge_out else None
        self._ck(self.lib.terra_tiles_mesh_shadows_halo_dev(self.ctx, txy.ctypes.data, n, z_ptr, lp, smask_ptr,
                                                             ei.ctypes.data if ei is not None else None, ep.ctypes.data if ep is not None else None,
                                                             eo.ctypes.data if eo is not None else None))
        return eo

    def heightmap_write_png(self, path, pixels):
        """pixels: (h, w) uint8 or (h, w, 2) uint8 {lo, hi}"""
        px = np.ascontiguousarray(pixels, np.uint8)
        self._ck(self.lib.terra_heightmap_write_png(str(path).encode(), px.ctypes.data, px.shape[1], px.shape[0], 2 if px.ndim == 3 else 1))

    def heightmap_read_png(self, path, allow_two_byte_grayscale=True):
        w, h, nc = _u32(), _u32(), _i32()
        self._ck(self.lib.terra_heightmap_read_png(str(path).encode(), int(allow_two_byte_grayscale), C.byref(w), C.byref(h), C.byref(nc), None, 0))
        out = np.empty((h.value, w.value, 2) if nc.value == 2 else (h.value, w.value), np.uint8)
        self._ck(self.lib.terra_heightmap_read_png(str(path).encode(), int(allow_two_byte_grayscale), C.byref(w), C.byref(h), C.byref(nc), out.ctypes.data, out.nbytes))
        return out

    def read_mesh(self, path, nx, ny, zmm=0.0):
        """read_mesh (src/mesh_gen.cpp:895-933): -> (mesh [ny, nx] float32 = mesh_file_scale*height + mesh_file_tz, (zbottom, ztop)); the context's zmax_est / zmin / zmax /
        water_plane_z are updated as the reference's globals are"""
        out = np.empty((ny, nx), np.float32)
        zz = np.zeros(2, np.float32)
        self._ck(self.lib.terra_read_mesh(self.ctx, os.fsencode(str(path)), float(zmm), out.ctypes.data, nx, ny, zz.ctypes.data))
        return out, (zz[0], zz[1])

    def write_mesh(self, path, mesh):
        m = np.ascontiguousarray(mesh, np.float32)
        self._ck(self.lib.terra_write_mesh(os.fsencode(str(path)), m.ctypes.data, m.shape[1], m.shape[0]))

    def set_tiled_mesh_ao(self, enable):
        self._ck(self.lib.terra_set_tiled_mesh_ao(self.ctx, int(bool(enable))))

    def hmap_apply_brushes_dev(self, brushes, step_sz=1, num_steps=1):
        b = np.ascontiguousarray(brushes, BRUSH_DTYPE).reshape(-1)
        self._ck(self.lib.terra_hmap_apply_brushes_dev(self.ctx, b.ctypes.data, len(b), step_sz, num_steps))

    def hmap_apply_mods_dev(self, mods):
        m = np.ascontiguousarray(mods, MOD_DTYPE).reshape(-1)
        self._ck(self.lib.terra_hmap_apply_mods_dev(self.ctx, m.ctypes.data, len(m)))

    def hmap_read_and_apply_mod_dev(self, path):
        self._ck(self.lib.terra_hmap_read_and_apply_mod_dev(self.ctx, str(path).encode()))

    def hmap_write_mod(self, path, mods, brushes):
        m = np.ascontiguousarray(mods, MOD_DTYPE).reshape(-1); b = np.ascontiguousarray(brushes, BRUSH_DTYPE).reshape(-1)
        self._ck(self.lib.terra_hmap_write_mod(str(path).encode(), m.ctypes.data, len(m), b.ctypes.data, len(b)))

    def hmap_read_mod(self, path):
        n, nb = _u32(), _u32()
        self._ck(self.lib.terra_hmap_read_mod(str(path).encode(), None, 0, C.byref(n), None, 0, C.byref(nb)))
        m = np.zeros(n.value, MOD_DTYPE); b = np.zeros(nb.value, BRUSH_DTYPE)
        self._ck(self.lib.terra_hmap_read_mod(str(path).encode(), m.ctypes.data, len(m), C.byref(n), b.ctypes.data, len(b), C.byref(nb)))
        return m, b

    def export_heightmap_dev(self, xstart, ystart, width, height, vals_ptr, pix_ptr=None):
        r = (C.c_float * 2)()
        self._ck(self.lib.terra_export_heightmap_dev(self.ctx, xstart, ystart, width, height, vals_ptr, pix_ptr, r))
        return r[0], r[1]

    def write_map_mode_heightmap_image(self, path, xstart, ystart, width, height):
        self._ck(self.lib.terra_write_map_mode_heightmap_image(self.ctx, str(path).encode(), xstart, ystart, width, height))

    def set_landscape(self, ls):
        self._ck(self.lib.terra_set_landscape(self.ctx, C.byref(ls)))

    def get_landscape(self):
        ls = Landscape()
        self._ck(self.lib.terra_get_landscape(self.ctx, C.byref(ls)))
        return ls

    def tiles_terrain_params(self, tile_xy):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        out = np.empty((len(txy), 2, 2, 3), np.float32)
        self._ck(self.lib.terra_tiles_terrain_params(self.ctx, txy.ctypes.data, len(txy), out.ctypes.data))
        return out

    def tiles_create_weights(self, tile_xy, zvals):
        """-> (weights u8 [n,129,129,4], grass blocks [n,32,32] of GRASS_BLOCK_DTYPE, has_any_grass bool [n])"""
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        z = np.ascontiguousarray(zvals, np.float32).reshape(n, 130, 130)
        w = np.empty((n, 129, 129, 4), np.uint8); gb = np.empty((n, 32, 32), GRASS_BLOCK_DTYPE); hg = np.empty(n, np.uint8)
        self._ck(self.lib.terra_tiles_create_weights(self.ctx, txy.ctypes.data, n, z.ctypes.data, w.ctypes.data, gb.ctypes.data, hg.ctypes.data))
        return w, gb, hg.astype(bool)

    def tiles_ao_lighting(self, tile_xy, zvals):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        z = np.ascontiguousarray(zvals, np.float32).reshape(n, 130, 130)
        ao = np.empty((n, 129, 129), np.uint8)
        self._ck(self.lib.terra_tiles_ao_lighting(self.ctx, txy.ctypes.data, n, z.ctypes.data, ao.ctypes.data))
        return ao

    def voxel_fill(self, nx, ny, nz, lo_pos, vsz, offset, mag, freq, rseed1, rseed2, gen_mode, zscale, normalize):
        out = np.empty((ny, nx, nz), np.float32)
        a = lambda v: (C.c_float * 3)(*v)
        self._ck(self.lib.terra_voxel_fill(self.ctx, out.ctypes.data, nx, ny, nz, a(lo_pos), a(vsz), a(offset), mag, freq, rseed1, rseed2, gen_mode, zscale, normalize))
        return out

    # ---- device-resident entry points (ptr = raw device pointer, e.g. torch_tensor.data_ptr())
    def gen_grid_dev(self, ptr, x0, y0, dx, dy, nx, ny, flags=GEN_GLACIATE, min_start_sin=0):
        self._ck(self.lib.terra_gen_grid_dev(self.ctx, x0, y0, dx, dy, nx, ny, flags, min_start_sin, ptr))

    def gen_grid_minmax_dev(self, ptr, x0, y0, dx, dy, nx, ny, flags=GEN_GLACIATE, min_start_sin=0):
        mn, mx = C.c_float(), C.c_float()
        self._ck(self.lib.terra_gen_grid_minmax_dev(self.ctx, x0, y0, dx, dy, nx, ny, flags, min_start_sin, ptr, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    def gen_grid_rows_minmax_dev(self, ptr, x0, y0, dx, dy, nx, ny, row0, nrows, flags=GEN_GLACIATE, min_start_sin=0, want_minmax=True):
        mn, mx = C.c_float(), C.c_float()
        self._ck(self.lib.terra_gen_grid_rows_minmax_dev(self.ctx, x0, y0, dx, dy, nx, ny, flags, min_start_sin, row0, nrows, ptr,
                                                         C.byref(mn) if want_minmax else None, C.byref(mx) if want_minmax else None))
        return (mn.value, mx.value) if want_minmax else None

    def eval_points(self, xy, exact, xy_scale=1.0, no_xyoff=False, xoff2=0, yoff2=0):
        """terra_eval_points: eval_mesh_sin_terms_scaled (exact=False) / get_exact_zval (exact=True) for the points xy[n][2]"""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.empty(len(xy), np.float32)
        self._ck(self.lib.terra_eval_points(self.ctx, xy.ctypes.data, len(xy), 1 if exact else 0, xy_scale, int(bool(no_xyoff)), xoff2, yoff2, out.ctypes.data))
        return out

    def eval_points_dev(self, xy_ptr, n, out_ptr, exact, xy_scale=1.0, no_xyoff=False, xoff2=0, yoff2=0):
        self._ck(self.lib.terra_eval_points_dev(self.ctx, xy_ptr, n, 1 if exact else 0, xy_scale, int(bool(no_xyoff)), xoff2, yoff2, out_ptr))

    def eval_mesh_sin_terms(self, xv, yv):
        out = C.c_float()
        self._ck(self.lib.terra_eval_mesh_sin_terms(self.ctx, xv, yv, C.byref(out)))
        return out.value

    def glaciate_mesh_dev(self, ptr, nx, ny, xoff2=0, yoff2=0):
        r = (C.c_float * 2)()
        self._ck(self.lib.terra_glaciate_mesh_dev(self.ctx, ptr, nx, ny, xoff2, yoff2, C.addressof(r)))
        return r[0], r[1]

    def gen_grid_minmax_async_dev(self, ptr, x0, y0, dx, dy, nx, ny, minmax_ptr, flags=GEN_GLACIATE, min_start_sin=0):
        """noise (+ glaciate) with {min, max} left in device memory at minmax_ptr (2 floats); nothing is read back, the call only enqueues"""
        self._ck(self.lib.terra_gen_grid_minmax_async_dev(self.ctx, x0, y0, dx, dy, nx, ny, flags, min_start_sin, ptr, minmax_ptr))

    def gen_grid_rows_minmax_async_dev(self, ptr, x0, y0, dx, dy, nx, ny, row0, nrows, minmax_ptr, flags=GEN_GLACIATE, min_start_sin=0):
        """rows [row0, row0 + nrows) with the strip's {min, max} left in device memory at minmax_ptr; the call only enqueues"""
        self._ck(self.lib.terra_gen_grid_rows_minmax_async_dev(self.ctx, x0, y0, dx, dy, nx, ny, flags, min_start_sin, row0, nrows, ptr, minmax_ptr))

    def apply_erosion_devmin_dev(self, ptr, xsize, ysize, min_ptr, iters, flags=0):
        """apply_erosion with min_zval read from device memory (one float) when the final clamp runs"""
        self._ck(self.lib.terra_apply_erosion_devmin_dev(self.ctx, ptr, xsize, ysize, min_ptr, iters, flags))

    def erosion_shard_arena_bytes(self, iters):
        """bytes of one rank's arena of a sharded erosion (terra_erosion_shard_*)"""
        return int(self.lib.terra_erosion_shard_arena_bytes(self.ctx, iters))

    def erosion_shard_trace_dev(self, ptr, xsize, ysize, iters, row0, nrows, arena_ptr):
        """probe / trace the droplets that start in rows [row0, row0 + nrows) into this rank's arena (nothing is written to the grid)"""
        self._ck(self.lib.terra_erosion_shard_trace_dev(self.ctx, ptr, xsize, ysize, iters, row0, nrows, arena_ptr))

    def erosion_shard_finish_dev(self, ptr, xsize, ysize, min_ptr, iters, flags, world, self_rank, row_end, arena_self_ptr, arena_stride):
        """the eroding rank: gather the ranks' traces, check / commit / re-trace / clamp = apply_erosion_devmin_dev on the same grid"""
        re = (C.c_uint32 * world)(*[int(v) for v in row_end])
        self._ck(self.lib.terra_erosion_shard_finish_dev(self.ctx, ptr, xsize, ysize, min_ptr, iters, flags, world, self_rank, re, arena_self_ptr, arena_stride))

    def event_create(self):
        e = C.c_void_p()
        self._ck(self.lib.terra_event_create(self.ctx, C.byref(e)))
        return e

    def event_record(self, ev): self._ck(self.lib.terra_event_record(self.ctx, ev))
    def event_wait(self, ev): self._ck(self.lib.terra_event_wait(self.ctx, ev))
    def event_synchronize(self, ev): self._ck(self.lib.terra_event_synchronize(ev))
    def event_destroy(self, ev): self.lib.terra_event_destroy(ev)

    def apply_erosion_dev(self, ptr, xsize, ysize, min_zval, iters, flags=0):
        self._ck(self.lib.terra_apply_erosion_dev(self.ctx, ptr, xsize, ysize, min_zval, iters, flags))

    def heightmap_proc_gen_dev(self, ptr, width, height, erosion_iters, pix_ptr=None):
        rng = (C.c_float * 2)()
        self._ck(self.lib.terra_heightmap_proc_gen_dev(self.ctx, width, height, erosion_iters, ptr, pix_ptr, C.addressof(rng)))
        return rng[0], rng[1]

    def heightmap_proc_gen(self, width, height, erosion_iters):
        """-> (pixels u8 [h, w, 2], min_z, dz): heightmap_t::proc_gen's texture on the host"""
        pix = np.empty((height, width, 2), np.uint8)
        rng = (C.c_float * 2)()
        self._ck(self.lib.terra_heightmap_proc_gen(self.ctx, width, height, erosion_iters, pix.ctypes.data, rng))
        return pix, rng[0], rng[1]

    def minmax_dev(self, ptr, n):
        mn, mx = C.c_float(), C.c_float()
        self._ck(self.lib.terra_minmax_dev(self.ctx, ptr, n, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    def quantize16_dev(self, ptr, n, min_z, dz, pix_ptr):
        self._ck(self.lib.terra_quantize16_dev(self.ctx, ptr, n, min_z, dz, pix_ptr))

    def tiles_create_zvals_dev(self, tile_xy, iters_tt, z_ptr, stats_ptr=None, normals_ptr=None, mnz_ptr=None):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        self._ck(self.lib.terra_tiles_create_zvals_dev(self.ctx, txy.ctypes.data, len(txy), iters_tt, z_ptr, stats_ptr, normals_ptr, mnz_ptr))

    def tiles_post_dev(self, tile_xy, z_ptr, stats_ptr=None, normals_ptr=None, mnz_ptr=None):
        """sub-block ranges / water bbox / radius and normals of zvals the caller has (tile_t::create_zvals' last loop + upload_normal_texture)"""
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        self._ck(self.lib.terra_tiles_post_dev(self.ctx, txy.ctypes.data, len(txy), z_ptr, stats_ptr, normals_ptr, mnz_ptr))

    def tiles_create_weights_dev(self, tile_xy, z_ptr, weights_ptr, blocks_ptr=None, has_grass_ptr=None):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        self._ck(self.lib.terra_tiles_create_weights_dev(self.ctx, txy.ctypes.data, len(txy), z_ptr, weights_ptr, blocks_ptr, has_grass_ptr))

    def tiles_ao_lighting_dev(self, tile_xy, z_ptr, ao_ptr):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        self._ck(self.lib.terra_tiles_ao_lighting_dev(self.ctx, txy.ctypes.data, len(txy), z_ptr, ao_ptr))

    def voxel_fill_dev(self, ptr, nx, ny, nz, lo_pos, vsz, offset, mag, freq, rseed1, rseed2, gen_mode, zscale, normalize):
        a = lambda v: (C.c_float * 3)(*v)
        self._ck(self.lib.terra_voxel_fill_dev(self.ctx, ptr, nx, ny, nz, a(lo_pos), a(vsz), a(offset), mag, freq, rseed1, rseed2, gen_mode, zscale, normalize))

    def voxel_fill_slab_dev(self, ptr, nx, ny, nz, lo_pos, vsz, offset, mag, freq, rseed1, rseed2, gen_mode, zscale, normalize, y0, nys):
        a = lambda v: (C.c_float * 3)(*v)
        self._ck(self.lib.terra_voxel_fill_slab_dev(self.ctx, ptr, nx, ny, nz, a(lo_pos), a(vsz), a(offset), mag, freq, rseed1, rseed2, gen_mode, zscale, normalize, y0, nys))

    # ---- async generator handle (mesh_xy_grid_cache_t protocol)
    def generator(self):
        return Generator(self)


class DistributedGrid:
    """terra_dgrid: ONE array whose strips live on several GPUs, mapped back to back in this rank's address space (HIP virtual memory management).
    rank r: g = DistributedGrid(terra, strip_bytes, r); fd = g.export_fd() -> peers; g.import_fd(j, fd_j) for every other strip; ptr = g.map()"""

    def __init__(self, terra, strip_bytes, local_strip):
        self.t = terra
        arr = (_sz * len(strip_bytes))(*[int(b) for b in strip_bytes])
        h = _vp()
        terra._ck(terra.lib.terra_dgrid_create(terra.ctx, len(strip_bytes), arr, local_strip, C.byref(h)))
        self.h, self.strip_bytes, self.local, self.ptr = h, [int(b) for b in strip_bytes], local_strip, None

    @staticmethod
    def granularity(terra):
        return int(terra.lib.terra_dgrid_granularity(terra.ctx))

    def export_fd(self):
        fd = _i32()
        self.t._ck(self.t.lib.terra_dgrid_export_fd(self.h, C.byref(fd)))
        return fd.value

    def import_fd(self, strip, fd):
        self.t._ck(self.t.lib.terra_dgrid_import_fd(self.h, strip, fd))

    def map(self):
        p = _vp()
        self.t._ck(self.t.lib.terra_dgrid_map(self.h, C.byref(p)))
        self.ptr = p.value
        return self.ptr

    def strip_ptr(self, i):
        return self.ptr + sum(self.strip_bytes[:i])

    def destroy(self):
        if self.h:
            self.t.lib.terra_dgrid_destroy(self.h)
            self.h = None


class TerraMulti:
    """terra_multi: several contexts (one per entry of `devices`, indices may repeat) driven from this process, each by its own host thread inside a call."""

    def __init__(self, devices, lib_path=None):
        self.lib = load_library(lib_path)
        m = _vp()
        arr = (_i32 * len(devices))(*devices)
        rc = self.lib.terra_multi_create(C.byref(m), arr, len(devices))
        if rc != 0:
            raise TerraError(rc, self.lib.terra_last_error().decode())
        self.m = m
        self.n = len(devices)
        self.ctxs = []
        for i in range(self.n):  # borrowed contexts: the plumbing (alloc / upload / download) of Terra works on them, close() is the multi handle's
            t = Terra.__new__(Terra)
            t.lib, t.ctx = self.lib, _vp(self.lib.terra_multi_ctx(self.m, i))
            t.close = lambda: None
            self.ctxs.append(t)

    def close(self):
        if getattr(self, "m", None):
            for t in self.ctxs:
                t.ctx = None
            self.lib.terra_multi_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise TerraError(rc, self.lib.terra_last_error().decode())
        return rc

    def partition(self, n_units, part):
        a, b = _u32(), _u32()
        self.lib.terra_multi_partition(n_units, self.n, part, C.byref(a), C.byref(b))
        return a.value, b.value

    def init_scene(self, cfg):
        self._ck(self.lib.terra_multi_init_scene(self.m, C.byref(cfg)))
        return self.ctxs[0].state()

    def synchronize(self): self._ck(self.lib.terra_multi_synchronize(self.m))

    def dgrid_create(self, strip_bytes):
        """strip i on context i's device, all mapped back to back: (handle, device pointer valid on every context's device); free with dgrid_destroy"""
        arr = (_sz * self.n)(*[int(b) for b in strip_bytes])
        h, p = _vp(), _vp()
        self._ck(self.lib.terra_multi_dgrid_create(self.m, arr, C.byref(h), C.byref(p)))
        return h, p.value

    def dgrid_destroy(self, h): self.lib.terra_dgrid_destroy(h)

    def tiles_create_zvals(self, tile_xy, iters_tt=0):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        z = np.empty((n, 130, 130), np.float32); st = (TileStats * n)(); nm = np.empty((n, 129, 129, 4), np.uint8); mnz = np.empty(n, np.float32)
        self._ck(self.lib.terra_multi_tiles_create_zvals(self.m, txy.ctypes.data, n, iters_tt, z.ctypes.data, C.addressof(st), nm.ctypes.data, mnz.ctypes.data))
        return z, st, nm, mnz

    def tiles_create_zvals_dev(self, tile_xy, iters_tt, z_ptrs, stats_ptrs=None, normals_ptrs=None, mnz_ptrs=None):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        arr = lambda ps: None if ps is None else (_vp * self.n)(*ps)
        self._ck(self.lib.terra_multi_tiles_create_zvals_dev(self.m, txy.ctypes.data, len(txy), iters_tt, arr(z_ptrs), arr(stats_ptrs), arr(normals_ptrs), arr(mnz_ptrs)))

    def gen_grid_rows_dev(self, ptrs, x0, y0, dx, dy, nx, ny, flags=GEN_GLACIATE, min_start_sin=0):
        mn, mx = C.c_float(), C.c_float()
        self._ck(self.lib.terra_multi_gen_grid_rows_dev(self.m, x0, y0, dx, dy, nx, ny, flags, min_start_sin, (_vp * self.n)(*ptrs), C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    def voxel_fill_dev(self, ptrs, nx, ny, nz, lo_pos, vsz, offset, mag, freq, rseed1, rseed2, gen_mode, zscale, normalize):
        a = lambda v: (C.c_float * 3)(*v)
        self._ck(self.lib.terra_multi_voxel_fill_dev(self.m, (_vp * self.n)(*ptrs), nx, ny, nz, a(lo_pos), a(vsz), a(offset), mag, freq, rseed1, rseed2, gen_mode, zscale, normalize))

    def tiles_mesh_shadows(self, tile_xy, zvals, light_pos):
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        z = np.ascontiguousarray(zvals, np.float32).reshape(n, 130, 130)
        sm = np.empty((n, 130, 130), np.uint8)
        self._ck(self.lib.terra_multi_tiles_mesh_shadows(self.m, txy.ctypes.data, n, z.ctypes.data, (C.c_float * 3)(*light_pos), sm.ctypes.data))
        return sm

    def shadow_layout(self, tile_xy, light_pos):
        """where the device-resident mesh-shadow pass wants the tiles: (ctx_of_tile, pos_in_ctx, tiles_per_ctx)"""
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        n = len(txy)
        a, b, c = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(self.n, np.uint32)
        self._ck(self.lib.terra_multi_shadow_layout(self.m, txy.ctypes.data, n, (C.c_float * 3)(*light_pos), a.ctypes.data, b.ctypes.data, c.ctypes.data))
        return a, b, c

    def tiles_mesh_shadows_dev(self, tile_xy, z_ptrs, light_pos, smask_ptrs):
        """z_ptrs[s] / smask_ptrs[s]: context s's strip in the order of shadow_layout, on its device"""
        txy = np.ascontiguousarray(tile_xy, np.int32).reshape(-1, 2)
        zp = (_vp * self.n)(*z_ptrs); sp = (_vp * self.n)(*smask_ptrs)
        self._ck(self.lib.terra_multi_tiles_mesh_shadows_dev(self.m, txy.ctypes.data, len(txy), zp, (C.c_float * 3)(*light_pos), sp))

    def foreach(self, fn):
        """fn(Terra, index) -> int on every context's host thread at once (the callback re-enters Python: the calls serialise on the GIL except inside the library)"""
        cb_t = C.CFUNCTYPE(C.c_int, _vp, _u32, _vp)
        def tramp(ctx, index, _user):
            try:
                r = fn(self.ctxs[index], index)
                return 0 if r is None else int(r)
            except Exception:  # noqa: BLE001
                return -2
        cb = cb_t(tramp)
        self._ck(self.lib.terra_multi_foreach(self.m, C.cast(cb, _vp), None))


class Generator:
    def __init__(self, terra):
        self.t = terra
        g = _vp()
        terra._ck(terra.lib.terra_gen_create(terra.ctx, C.byref(g)))
        self.g = g
        self.shape = None

    def build_arrays(self, x0, y0, dx, dy, nx, ny, flags=0, min_start_sin=0):
        self.shape = (ny, nx)
        return self.t._ck(self.t.lib.terra_gen_build_arrays(self.g, x0, y0, dx, dy, nx, ny, flags, min_start_sin))

    def enable_glaciate(self): self.t._ck(self.t.lib.terra_gen_enable_glaciate(self.g))
    def is_running(self): return bool(self.t.lib.terra_gen_is_running(self.g))
    def eval_index(self, x, y, min_start_sin=0, use_cache=True): return self.t.lib.terra_gen_eval_index(self.g, x, y, min_start_sin, 1 if use_cache else 0)

    def collect(self):
        out = np.empty(self.shape, np.float32)
        self.t._ck(self.t.lib.terra_gen_collect(self.g, out.ctypes.data))
        return out

    def close(self):
        if self.g:
            self.t.lib.terra_gen_destroy(self.g)
            self.g = None
