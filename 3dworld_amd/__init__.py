"""3dworld_amd -- MI355X-native procedural-terrain hot path of 3DWorld (heightmap noise, droplet erosion, tile normals,
voxel noise) behind a C ABI (include/terra.h, libterra_hip.so).

This package is only the Python glue used by tests/, bench.py and __graft_entry__.py: ctypes bindings to the C ABI and the
hipcc build recipe.  The product is the shared library; the engine-side (C++) mirror of the reference interface is
include/terra_cxx.hpp.  There is no CPU fall-back: loading fails loudly when libterra_hip.so or a HIP device is missing.

(The directory name starts with a digit, so import it with importlib: `terra = importlib.import_module("3dworld_amd")`.)
"""
from .terra import (Terra, TerraMulti, DistributedGrid, TerraError, Config, State, TileStats, ErosionReport, Landscape, make_landscape, GRASS_BLOCK_DTYPE, BRUSH_DTYPE, MOD_DTYPE, make_config, default_lib_path,
                    GEN_GLACIATE, GEN_FORCE_SINE, GEN_NO_WAIT, GEN_CACHE_VALUES, GEN_FUSED, GEN_FAST, ERODE_SERIAL, ERODE_MINZ_IS_MIN, ERODE_SERIAL_WAVE,
                    MGEN_SINE, MGEN_SIMPLEX, MGEN_PERLIN, MGEN_SIMPLEX_GPU, MGEN_DWARP_GPU)
from .build import build_library

__all__ = ["Terra", "TerraError", "Config", "State", "TileStats", "ErosionReport", "Landscape", "make_landscape", "GRASS_BLOCK_DTYPE", "BRUSH_DTYPE", "MOD_DTYPE", "make_config", "default_lib_path", "build_library"]
