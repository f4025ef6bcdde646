#!/usr/bin/env python3
"""oracle/engine_patch.py -- TEST INFRASTRUCTURE.  Applies the engine-side patch of INTEGRATION.md (sections 2 and 3) to a SCRATCH copy of the reference's sources,
outside the repository, so that the reference's own callers (heightmap_t::proc_gen, tile_t::create_zvals, gen_mesh ...) can be compiled against the C ABI of include/terra.h
and run: "engine in the loop".  Nothing of the reference is copied into the repository; the scratch directory (default /tmp/terra_engine_src) holds the copies, oracle/_ref/
only receives the linked harness libraries (oracle/Makefile, target `engine`).

    engine_patch.py <reference root> <scratch dir>

What is patched (every anchor must be found exactly once, otherwise the script fails -- a changed reference must not be patched blindly):
  src/mesh.h        class mesh_xy_grid_cache_t gets the opaque handle member `struct terra_gen *hip_gen`                                     
  src/mesh_gen.cpp  `bool use_hip_terrain`; build_arrays(): the HIP backend beside the GL one -- launch / no_wait / collect into cached_vals;
                    eval_index(): a first sine term above start_eval_sin goes through terra_gen_eval_index; clear_context(): destroy the handle   (section 2)
  src/erosion.cpp   apply_erosion(): one line that hands the call to terra_cxx::apply_erosion                                               (section 3)
  src/heightmap.cpp heightmap_t::proc_gen(): the whole body as one call (terra_cxx::heightmap_proc_gen) when there are no cities                    (section 4)
"""
import os
import shutil
import sys

TUS = ["mesh_gen", "erosion", "upsurface", "visibility", "Math3d", "heightmap", "tiled_mesh", "Textures"]


def sub_once(text, anchor, replacement, what):
    if text.count(anchor) != 1:
        raise SystemExit(f"engine_patch: anchor for {what} found {text.count(anchor)} times (expected 1): {anchor[:70]!r}")
    return text.replace(anchor, replacement)


def main(ref, out):
    src = os.path.join(ref, "src")
    os.makedirs(out, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith((".h", ".hpp", ".inl")) or f[:-4] in TUS and f.endswith(".cpp"):
            shutil.copyfile(os.path.join(src, f), os.path.join(out, f))
    # ---- mesh.h
    p = os.path.join(out, "mesh.h")
    t = open(p).read()
    t = sub_once(t, "\tgrid_gen_shader_t *cshader=nullptr;\n", "\tgrid_gen_shader_t *cshader=nullptr;\n\tstruct terra_gen *hip_gen=nullptr; // opaque handle from terra.h (INTEGRATION.md section 2)\n", "mesh.h member")
    open(p, "w").write(t)
    # ---- mesh_gen.cpp
    p = os.path.join(out, "mesh_gen.cpp")
    t = open(p).read()
    t = sub_once(t, "bool mesh_xy_grid_cache_t::build_arrays(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, bool cache_values, bool force_sine_mode, bool no_wait) {",
                 '#include "terra_cxx.hpp" // <repo>/include\nbool use_hip_terrain(0); // config key bound in load_config (src/3DWorld.cpp:1892)\nunsigned hip_terrain_calls(0); // (test only: calls that went through the HIP backend)\n\n'
                 "bool mesh_xy_grid_cache_t::build_arrays(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, bool cache_values, bool force_sine_mode, bool no_wait) {", "build_arrays head")
    t = sub_once(t, "\tif (gen_mode >= MGEN_SIMPLEX_GPU) { // GPU simplex noise - always cache values\n",
                 "\tif (use_hip_terrain) { // same protocol as the GL path: launch, maybe return 0, collect into cached_vals\n"
                 "\t\t++hip_terrain_calls;\n"
                 "\t\tif (!hip_gen) {terra_cxx::check(terra_gen_create(terra_cxx::default_ctx(), &hip_gen), \"terra_gen_create\");}\n"
                 "\t\tunsigned const flags((force_sine_mode ? TERRA_GEN_FORCE_SINE : 0) | (no_wait ? TERRA_GEN_NO_WAIT : 0));\n"
                 "\t\tint const ready(terra_gen_build_arrays(hip_gen, x0, y0, dx, dy, nx, ny, flags, 0));\n"
                 "\t\tterra_cxx::check(ready, \"build_arrays\");\n"
                 "\t\tif (!ready) return 0; // just launched (tile_draw_t::update polls next frame)\n"
                 "\t\tcached_vals.resize(size_t(nx)*ny);\n"
                 "\t\tterra_cxx::check(terra_gen_collect(hip_gen, cached_vals.data()), \"collect\"); // like cache_gpu_simplex_vals(); values are NOT glaciated, eval_index() does that as before\n"
                 "\t\treturn 1;\n"
                 "\t}\n"
                 "\tif (gen_mode >= MGEN_SIMPLEX_GPU) { // GPU simplex noise - always cache values\n", "build_arrays backend")
    t = sub_once(t, "\tif ((use_cache || gen_mode >= MGEN_SIMPLEX_GPU) && !cached_vals.empty()) {\n",
                 "\tif (hip_gen && use_hip_terrain && gen_mode == MGEN_SINE && max(start_eval_sin, min_start_sin) != start_eval_sin) { // a later first sine term (create_texture: 50): evaluated from that term on the device\n"
                 "\t\tzval += terra_gen_eval_index(hip_gen, x, y, min_start_sin, use_cache);\n"
                 "\t}\n"
                 "\telse if ((use_cache || gen_mode >= MGEN_SIMPLEX_GPU || (hip_gen && use_hip_terrain)) && !cached_vals.empty()) {\n", "eval_index")
    t = sub_once(t, "void mesh_xy_grid_cache_t::clear_context() { // for GPU-mode cached state\n",
                 "void mesh_xy_grid_cache_t::clear_context() { // for GPU-mode cached state\n\tif (hip_gen) {terra_gen_destroy(hip_gen); hip_gen = nullptr;}\n", "clear_context")
    open(p, "w").write(t)
    # ---- erosion.cpp
    p = os.path.join(out, "erosion.cpp")
    t = open(p).read()
    t = sub_once(t, "void apply_erosion(float *heightmap, int xsize, int ysize, float min_zval, unsigned num_iters) {\n",
                 '#include "terra_cxx.hpp"\nextern bool use_hip_terrain;\nextern unsigned hip_terrain_calls;\n\n'
                 "void apply_erosion(float *heightmap, int xsize, int ysize, float min_zval, unsigned num_iters) {\n"
                 "\tif (use_hip_terrain) {++hip_terrain_calls; terra_cxx::apply_erosion(heightmap, xsize, ysize, min_zval, num_iters); return;} // serial-order-exact\n", "apply_erosion")
    open(p, "w").write(t)
    # ---- heightmap.cpp (section 4): the whole of proc_gen on the device when there are no cities -- the texture's pixels are the only thing that crosses the host link
    p = os.path.join(out, "heightmap.cpp")
    t = open(p).read()
    t = sub_once(t, "void heightmap_t::proc_gen() {\n\tset_16_bit_grayscale();\n\talloc();\n",
                 '#include "terra_cxx.hpp"\nextern bool use_hip_terrain;\nextern unsigned hip_terrain_calls;\nbool use_hip_proc_gen(1); // (test only: 0 keeps proc_gen\'s own body, whose build_arrays / apply_erosion then go through sections 2 and 3)\n\n'
                 "void heightmap_t::proc_gen() {\n\tset_16_bit_grayscale();\n\talloc();\n"
                 "\tif (use_hip_terrain && use_hip_proc_gen && !have_cities()) { // eval loop, min, erosion, z range and from_floats in one call\n"
                 "\t\t++hip_terrain_calls;\n"
                 "\t\tfloat min_z(0), dz(0);\n"
                 "\t\tterra_cxx::heightmap_proc_gen(width, height, erosion_iters_tt, get_data(), min_z, dz);\n"
                 "\t\tset_mesh_height_scales_for_zval_range(min_z, dz/255.0);\n"
                 "\t\treturn;\n"
                 "\t}\n", "proc_gen")
    open(p, "w").write(t)
    print(f"engine_patch: {out} ready ({len(os.listdir(out))} files)")


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2])
