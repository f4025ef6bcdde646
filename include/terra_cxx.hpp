// terra_cxx.hpp -- engine-side C++ mirror of the reference's call surface over the C ABI (include/terra.h).
//
// Same names, argument meaning and blocking behaviour as the 3DWorld interfaces they stand in for, so the callers
// (tile_t::create_zvals, heightmap_t::proc_gen, gen_mesh, voxel_manager) stay untouched:
//   mesh_xy_grid_cache_t  src/mesh.h:22-45        (build_arrays / enable_glaciate / eval_index / clear_context / free_cshader)
//   apply_erosion         src/function_registry.h:354, src/erosion.cpp:14
//   create_procedural     src/voxels.cpp:278      (fill part)
// Header only; link with -lterra_hip.  The reference asserts on misuse; so does this header (terra_status < 0 -> assert + message).
#pragma once
#include "terra.h"
#include <vector>
#include <cassert>
#include <cstdio>
#include <cstdlib>

namespace terra_cxx {

inline void check(int rc, char const *what) {
	if (rc < 0) {std::fprintf(stderr, "terra: %s failed (%d): %s\n", what, rc, terra_last_error()); assert(!"terra call failed"); std::abort();}
}

// process-wide context, like the engine's process-wide globals; created on first use on GPU 0 (TERRA_DEVICE overrides)
inline terra_ctx *default_ctx() {
	static terra_ctx *ctx = [] {
		terra_ctx *c = nullptr;
		char const *dev = std::getenv("TERRA_DEVICE");
		check(terra_create(&c, dev ? std::atoi(dev) : 0), "terra_create");
		return c;
	}();
	return ctx;
}

// Call once after the engine has loaded its config and derived its globals (after gen_scene()/estimate_zminmax()):
// copies the config-file values and the derived globals across the boundary.  Alternatively terra_init_scene() derives them itself.
inline void set_engine_state(terra_config const &cfg, terra_state const &st) {
	check(terra_set_config(default_ctx(), &cfg), "terra_set_config");
	check(terra_set_state(default_ctx(), &st), "terra_set_state");
}

class mesh_xy_grid_cache_t { // src/mesh.h:22-45
	terra_gen *gen = nullptr;
	unsigned cur_nx = 0, cur_ny = 0;
	terra_gen *handle() {if (!gen) {check(terra_gen_create(default_ctx(), &gen), "terra_gen_create");} return gen;}
public:
	mesh_xy_grid_cache_t() = default;
	mesh_xy_grid_cache_t(mesh_xy_grid_cache_t const &) = delete;
	mesh_xy_grid_cache_t &operator=(mesh_xy_grid_cache_t const &) = delete;
	~mesh_xy_grid_cache_t() {clear_context();}
	// returns 1 when values are available, 0 when no_wait and the job was only launched (call again with the same arguments)
	bool build_arrays(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, bool cache_values=0, bool force_sine_mode=0, bool no_wait=0) {
		assert(nx > 0 && ny > 0);
		unsigned const flags = (cache_values ? TERRA_GEN_CACHE_VALUES : 0u) | (force_sine_mode ? TERRA_GEN_FORCE_SINE : 0u) | (no_wait ? TERRA_GEN_NO_WAIT : 0u);
		int const rc = terra_gen_build_arrays(handle(), x0, y0, dx, dy, nx, ny, flags, 0);
		check(rc, "build_arrays");
		cur_nx = nx; cur_ny = ny;
		return rc != 0;
	}
	void enable_glaciate() {check(terra_gen_enable_glaciate(handle()), "enable_glaciate");}
	// min_start_sin > start_eval_sin (tile_t::create_texture passes 50, src/tiled_mesh.cpp:1114; grass.cpp:819-820): the library evaluates the grid once more from
	// that term on the first such call and serves the following ones from it
	float eval_index(unsigned x, unsigned y, int min_start_sin=0, bool use_cache=1) const {
		assert(x < cur_nx && y < cur_ny);
		return terra_gen_eval_index(gen, x, y, min_start_sin, use_cache ? 1 : 0);
	}
	float const *device_values() const {return terra_gen_device_values(gen);} // extension: the grid stays in HBM for erosion / normals
	void clear_context() {if (gen) {terra_gen_destroy(gen); gen = nullptr;}}
	void free_cshader() {} // nothing GL-side to release
};

// src/erosion.cpp:14 -- in place on a host buffer, blocking, silent no-op when disabled
inline void apply_erosion(float *heightmap, int xsize, int ysize, float min_zval, unsigned num_iters) {
	check(terra_apply_erosion(default_ctx(), heightmap, xsize, ysize, min_zval, num_iters), "apply_erosion");
}

// heightmap_t::proc_gen (src/heightmap.cpp:130-151) as one call: the texture's 16-bit pixels + {min_z, dz} for set_mesh_height_scales_for_zval_range(min_z, dz/255)
inline void heightmap_proc_gen(unsigned width, unsigned height, unsigned erosion_iters, unsigned char *pixels16, float &min_z, float &dz) {
	float range[2] = {0.0f, 0.0f};
	check(terra_heightmap_proc_gen(default_ctx(), width, height, erosion_iters, pixels16, range), "heightmap_proc_gen");
	min_z = range[0]; dz = range[1];
}

// tile_t::create_zvals for a batch of tiles (src/tiled_mesh.cpp:467-546): zvals n*130*130, stats n, normals n*129*129*4 (optional)
// ---- eval_mesh_sin_terms (src/mesh_gen.cpp:797-805): point query, evaluated on the host
inline float eval_mesh_sin_terms(float xv, float yv) {
	float z = 0.0f;
	check(terra_eval_mesh_sin_terms(default_ctx(), xv, yv, &z), "eval_mesh_sin_terms");
	return z;
}
// ---- eval_mesh_sin_terms_scaled (src/mesh_gen.cpp:807-813) and get_exact_zval (:816-847): the all-modes point queries; a batch at a time where the caller has one
// (biome parameters of a tile's corners src/tiled_mesh.cpp:332-338, voxel terrain src/voxels.cpp:434), else one point
inline void eval_mesh_sin_terms_scaled(float const *xy, unsigned n, float xy_scale, float *out) {
	check(terra_eval_points(default_ctx(), xy, n, TERRA_POINTS_SCALED, xy_scale, 0, 0, 0, out), "eval_mesh_sin_terms_scaled");
}
inline float eval_mesh_sin_terms_scaled(float xval, float yval, float xy_scale) {
	float const xy[2] = {xval, yval}; float z = 0.0f;
	eval_mesh_sin_terms_scaled(xy, 1, xy_scale, &z);
	return z;
}
inline void get_exact_zval(float const *xy, unsigned n, float *out, bool no_xyoff = false, int xoff2 = 0, int yoff2 = 0) { // xoff2 / yoff2: the reference's scroll-offset globals
	check(terra_eval_points(default_ctx(), xy, n, TERRA_POINTS_EXACT, 1.0f, no_xyoff ? 1 : 0, xoff2, yoff2, out), "get_exact_zval");
}
inline float get_exact_zval(float xval, float yval, bool no_xyoff = false, int xoff2 = 0, int yoff2 = 0) {
	float const xy[2] = {xval, yval}; float z = 0.0f;
	get_exact_zval(xy, 1, &z, no_xyoff, xoff2, yoff2);
	return z;
}
// ---- glaciate() over the ground mesh (src/mesh_gen.cpp:388-404): in place on a device buffer, returns zbottom / ztop
inline void glaciate_mesh_dev(float *d_mesh, unsigned nx, unsigned ny, int xoff2, int yoff2, float &zbottom, float &ztop) {
	float zz[2] = {0.0f, 0.0f};
	check(terra_glaciate_mesh_dev(default_ctx(), d_mesh, nx, ny, xoff2, yoff2, zz), "glaciate");
	zbottom = zz[0]; ztop = zz[1];
}

inline void tiles_create_zvals(int const *tile_xy, unsigned n, unsigned erosion_iters_tt, float *zvals, terra_tile_stats *stats, unsigned char *normals=nullptr, float *min_normal_z=nullptr) {
	check(terra_tiles_create_zvals(default_ctx(), tile_xy, n, erosion_iters_tt, zvals, stats, normals, min_normal_z), "tiles_create_zvals");
}

// ---- tile_t::upload_normal_texture (src/tiled_mesh.cpp:865-880) and the sub-block / water-bbox loop of create_zvals (:517-541) over zvals the engine already has
inline void tiles_upload_normal_texture(int const *tile_xy, unsigned n, float const *zvals, terra_tile_stats *stats, unsigned char *normals, float *min_normal_z=nullptr) {
	check(terra_tiles_post(default_ctx(), tile_xy, n, zvals, stats, normals, min_normal_z), "upload_normal_texture");
}

// ---- tile_t::calc_shadows_for_light (src/tiled_mesh.cpp:664-692) for a batch and one light: smask [n][130][130] gets the MESH_SHADOW bits
inline void tiles_mesh_shadows(int const *tile_xy, unsigned n, float const *zvals, float const lpos[3], unsigned char *smask) {
	check(terra_tiles_mesh_shadows(default_ctx(), tile_xy, n, zvals, lpos, smask), "calc_mesh_shadows");
}
// ---- terrain_hmap_manager: serve tiles from a heightmap texture that is already on the device (src/heightmap.cpp:385-407, src/mesh_gen.cpp:125-131)
inline void use_heightmap_texture(unsigned char const *d_pixels, int width, int height, int ncolors, float min_z, float dz) {
	check(terra_hmap_set_dev(default_ctx(), d_pixels, width, height, ncolors), "terrain_hmap_manager");
	if (d_pixels) {check(terra_set_mesh_height_scales_for_zval_range(default_ctx(), min_z, dz), "set_mesh_height_scales_for_zval_range");}
}
// ---- heightmap_t::postprocess_height (src/heightmap.cpp:117-128) on the loaded image: `data` = texture_t::data (width*height*ncolors bytes, host), eroded in place with
// erosion_iters_tt droplets; mesh_file_scale / mesh_file_tz as the config line `mh_filename <png> <scale> <tz>` set them.  Asserts like the reference when a value leaves [0, 256)
inline void heightmap_postprocess_height(unsigned char *data, unsigned width, unsigned height, int ncolors, unsigned erosion_iters_tt, float mesh_file_scale, float mesh_file_tz) {
	if (erosion_iters_tt == 0) return; // no erosion or cities => no need to update height values
	terra_ctx *ctx = default_ctx();
	size_t const bytes = (size_t)width*height*(size_t)ncolors;
	void *d_pixels = nullptr;
	check(terra_set_mesh_file_scale(ctx, mesh_file_scale, mesh_file_tz), "mh_filename scale");
	check(terra_malloc(ctx, &d_pixels, bytes), "postprocess_height");
	int rc = terra_memcpy_h2d(ctx, d_pixels, data, bytes);
	unsigned bad = 0;
	if (rc == 0) {rc = terra_heightmap_postprocess_dev(ctx, (unsigned char *)d_pixels, width, height, ncolors, erosion_iters_tt, nullptr, &bad);}
	if (rc == 0) {rc = terra_memcpy_d2h(ctx, data, d_pixels, bytes);}
	terra_free(ctx, d_pixels);
	check(rc, "postprocess_height");
	assert(bad == 0); // assert(v >= 0.0 && v < 256.0), src/heightmap.cpp:210
	(void)bad;
}
// ---- heightmap_t::write_png / texture_t::load_png for grayscale heightmaps (src/image_io.cpp:493-605)
inline void write_heightmap_png(char const *fn, unsigned char const *pixels, unsigned width, unsigned height, int ncolors) {
	check(terra_heightmap_write_png(fn, pixels, width, height, ncolors), "write_png");
}

// ---- read_mesh / write_mesh (src/mesh_gen.cpp:895-965): the ground mesh's text file.  Same return value as the reference (0: the file is missing, short or of another size);
// mesh_height is the engine's float ** matrix of MESH_Y_SIZE rows.  After read_mesh copy zmin / zmax / zmax_est / water_plane_z back from terra_get_state().
inline bool read_mesh(char const *filename, float zmm, float **mesh_height, unsigned mesh_x_size, unsigned mesh_y_size, float &zbottom, float &ztop) {
	if (filename == nullptr) return 0;
	std::vector<float> m((size_t)mesh_x_size*mesh_y_size);
	float zz[2];
	if (terra_read_mesh(default_ctx(), filename, zmm, m.data(), mesh_x_size, mesh_y_size, zz) != TERRA_OK) {std::fprintf(stderr, "read_mesh: %s\n", terra_last_error()); return 0;}
	for (unsigned i = 0; i < mesh_y_size; ++i) {for (unsigned j = 0; j < mesh_x_size; ++j) {mesh_height[i][j] = m[(size_t)i*mesh_x_size + j];}}
	zbottom = zz[0]; ztop = zz[1];
	return 1;
}
inline bool write_mesh(char const *filename, float const *const *mesh_height, unsigned mesh_x_size, unsigned mesh_y_size) {
	if (filename == nullptr || mesh_height == nullptr) return 0;
	std::vector<float> m((size_t)mesh_x_size*mesh_y_size);
	for (unsigned i = 0; i < mesh_y_size; ++i) {for (unsigned j = 0; j < mesh_x_size; ++j) {m[(size_t)i*mesh_x_size + j] = mesh_height[i][j];}}
	return terra_write_mesh(filename, m.data(), mesh_x_size, mesh_y_size) == TERRA_OK;
}

// ---- tile_t::calc_mesh_ao_lighting (src/tiled_mesh.cpp:586-661) for a batch: zvals [n][130][130] -> ao_lighting [n][129][129]
inline void tiles_ao_lighting(int const *tile_xy, unsigned n, float const *zvals, unsigned char *ao) {
	check(terra_tiles_ao_lighting(default_ctx(), tile_xy, n, zvals, ao), "calc_mesh_ao_lighting");
}
// ---- tile_t::create_texture's weights (src/tiled_mesh.cpp:1071-1240, terrain-only branch) for a batch: zvals [n][130][130] -> mesh_weight_data
// [n][129][129][4], grass_blocks [n][32][32] (may be null), has_any_grass [n] (may be null); set_landscape = the globals it reads (vegetation, ...)
inline void set_landscape(terra_landscape const &params) {check(terra_set_landscape(default_ctx(), &params), "set_landscape");}
inline void tiles_create_weights(int const *tile_xy, unsigned n, float const *zvals, unsigned char *mesh_weight_data, terra_grass_block *grass_blocks, unsigned char *has_any_grass) {
	check(terra_tiles_create_weights(default_ctx(), tile_xy, n, zvals, mesh_weight_data, grass_blocks, has_any_grass), "create_texture");
}
// tile_t::update_terrain_params (src/tiled_mesh.cpp:321-343): params [n][2][2]{veg, grass, dirt}
inline void tiles_terrain_params(int const *tile_xy, unsigned n, float *params) {check(terra_tiles_terrain_params(default_ctx(), tile_xy, n, params), "update_terrain_params");}
// voxel_manager::create_procedural fill (src/voxels.cpp:278-346): `vals` is the voxel_grid<float> storage, z fastest
inline void voxel_create_procedural(std::vector<float> &vals, unsigned nx, unsigned ny, unsigned nz, float const lo_pos[3], float const vsz[3], float const offset[3],
	float mag, float freq, bool normalize_to_1, int rseed1, int rseed2, int gen_mode, float zscale)
{
	vals.resize((size_t)nx*ny*nz);
	check(terra_voxel_fill(default_ctx(), vals.data(), nx, ny, nz, lo_pos, vsz, offset, mag, freq, rseed1, rseed2, gen_mode, zscale, normalize_to_1 ? 1 : 0), "voxel_create_procedural");
}

} // namespace terra_cxx
