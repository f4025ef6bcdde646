// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Link shim that lets the reference's OWN translation units
//     /root/reference/src/mesh_gen.cpp, erosion.cpp, upsurface.cpp, visibility.cpp, Math3d.cpp, heightmap.cpp, tiled_mesh.cpp, Textures.cpp   (+ vendored glm 0.9.9.1 headers)
// be compiled unmodified, in place, into oracle/_ref/liboracle_ref.so (recipe: oracle/Makefile).
// No reference source is copied: this file only (1) DEFINES the process globals those TUs declare
// `extern` (they live in 3DWorld.cpp / display_world.cpp / Universe.cpp / grass.cpp, which cannot be
// built without OpenGL), (2) STUBS the GL/IO entry points they reference but that the CPU path never
// calls (tiled_mesh.cpp and Textures.cpp are mostly renderer: -ffunction-sections + --gc-sections drop everything the exported
// harness does not reach, the link then needs a dozen libGL no-ops instead of hundreds of stubs), and (3) exports a plain C harness ("ref_*") that ctypes can drive.
// The tile rows are the reference's own members: tile_t::create_zvals, calc_mesh_ao_lighting, create_texture (+ add_grass_block_at, update_terrain_params),
// upload_normal_texture / get_norm, and get_tids / update_lttex_ix / gen_tex_height_tables / get_bare_ls_tid from Textures.cpp.
//
// The handful of functions that live in TUs we cannot build are restated here, each with its citation:
//   set_scene_constants   src/matrix_ops.cpp:59-84
//   rgen_core_t::randd    src/gen_object.cpp:377-381
//   the voxel fill loop   src/voxels.cpp:312-345       (voxels.o has >100 unrelated externals)
//   texture_t::set_16_bit_grayscale  src/image_io.cpp:493-496 (for heightmap.cpp)
//   write_map_mode_heightmap_image / get_heightmap_z_range  src/map_view.cpp:399-442 (driver over the reference's own setup_height_gen_cached / eval_index / interpolate_height)

// The tile functions under test are private members of tile_t (src/tiled_mesh.h:156-331): this harness -- and only it -- sees the reference's class
// definitions with every member public.  Access specifiers do not change g++'s object layout, so the objects are the ones tiled_mesh.o works on.
#define timer_t stdlib_timer_t // as src/3DWorld.h:7-8 does before its own system includes
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <assert.h>
#include <vector>
#include <memory>
#include <deque>
#include <algorithm>
#include <set>
#include <map>
#include <iostream>
#include <fstream>
#include <string>
#include <sstream>
#include <iterator>
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <list>
#include <array>
#include <cfloat>
#include <omp.h>
#define private public
#define protected public
#include "3DWorld.h"
#include "mesh.h"
#include "textures.h"
#include "heightmap.h"
#include "shaders.h"
#include "upsurface.h"
#include "sinf.h"
#include "tiled_mesh.h"
#undef private
#undef protected
#include <glm/gtc/noise.hpp>
#include <omp.h>
#include <cfloat>

#define REF_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------------
// (1) globals normally defined in 3DWorld.cpp / matrix_ops.cpp / Textures.cpp / display_world.cpp
// ---------------------------------------------------------------------------------------------
int MESH_X_SIZE(128), MESH_Y_SIZE(128), MESH_Z_SIZE(0);
float X_SCENE_SIZE(4.0), Y_SCENE_SIZE(4.0), Z_SCENE_SIZE(4.0);
int MESH_SIZE[3] = {0}, MAX_XY_SIZE(0), XY_MULT_SIZE(0), XY_SUM_SIZE(0), MAX_RUN_DIST(0), I_TIMESCALE(0);
float SCENE_SIZE[3] = {0}, MESH_HEIGHT(0), XY_SCENE_SIZE(0), TWO_XSS(0), TWO_YSS(0);
float DX_VAL(0), DY_VAL(0), HALF_DXY(0), DX_VAL_INV(0), DY_VAL_INV(0), DZ_VAL(0), dxdy(0), CLOUD_CEILING(0), LARGE_ZVAL(0);
double c_radius(0), c_theta(0), c_phi(0);
int camera_mode(0), world_mode(WMODE_INF_TERRAIN), do_read_mesh(0), read_heightmap(0), read_landscape(0), invert_mh_image(0);
int xoff(0), yoff(0), xoff2(0), yoff2(0), rand_gen_index(0), mesh_rgen_index(0), mesh_scale_change(0), mesh_seed(0), scrolling(0), display_mode(0);
point camera_origin, surface_pos, mesh_origin, camera_pos;
bool combined_gu(0);
float custom_glaciate_exp(0.0), disabled_mesh_z(FAR_DISTANCE), erode_amount(1.0), init_temperature(DEF_TEMPERATURE), temperature(DEF_TEMPERATURE), univ_temp(0.0);
float mesh_file_scale(1.0), mesh_file_tz(0.0), relh_adj_tex(0.0), read_mesh_zmm(0.0), water_h_off(0.0), water_h_off_rel(0.0), water_plane_z(0.0);
unsigned erosion_iters(0), erosion_iters_tt(0);
rand_gen_t global_rand_gen;
unsigned char **mesh_draw = NULL;
float **mesh_height = NULL;
char *mesh_file(nullptr), *mh_filename(nullptr), *mh_filename_tt(nullptr);
float ocean_wave_height(0.0); // reference default is DEF_OCEAN_WAVE_HEIGHT; harness sets it explicitly

extern float zmin, zmax, zmax_est, glaciate_exp, mesh_scale, mesh_scale_z, mesh_scale_z_inv, mesh_height_scale;
extern int start_eval_sin, GLACIATE, mesh_gen_mode, mesh_gen_shape, mesh_freq_filter;
extern float sinTable[][5];
extern float MESH_START_MAG, MESH_START_FREQ, MESH_MAG_MULT, MESH_FREQ_MULT;
extern hmap_params_t hmap_params;
extern ttex lttex_dirt[];
extern float h_dirt[NTEX_DIRT], clip_hd1; // src/Textures.cpp (compiled in place)
void gen_tex_height_tables();             // src/Textures.cpp:1757-1761
int get_bare_ls_tid(float zval);          // src/Textures.cpp:1284-1287

// ---------------------------------------------------------------------------------------------
// (2) stubs for GL / IO symbols referenced by the TUs but never reached on the CPU path
// ---------------------------------------------------------------------------------------------
static void ref_unreachable(char const *what) {fprintf(stderr, "oracle ref_shim: unexpected call to %s\n", what); abort();}
// GL "compute" job: there is no GL here.  The stubs make a launched job come back with NO cached values, so that
// mesh_xy_grid_cache_t::eval_index falls through to the reference's own CPU restatement of the GPU modes,
// get_noise_zval(xval, yval, gen_mode) (src/mesh_gen.cpp:759-765,734-751) -- the function the engine itself uses for
// exact height queries in modes 3/4 (src/mesh_gen.cpp:807-813), evaluated at eval_index's CPU coordinates.
void compute_shader_t::begin() {}
void compute_shader_t::end_shader() {}
void compute_shader_t::setup_and_run(unsigned &, bool, bool, bool) {is_running = 1;}
void compute_shader_t::prep_for_read_pixels(bool) {}
void compute_shader_t::read_float_vals(vector<float> &vals, bool, bool) {vals.clear(); is_running = 0;}
void shader_t::set_prefix(char const *, unsigned) {}
void shader_t::enable() {}
void shader_t::disable() {}
bool shader_t::add_uniform_float(char const *const, float) const {return 0;}
void texture_t::load(int, bool, bool, bool) {ref_unreachable("texture_t::load");}
void texture_t::set_16_bit_grayscale() {ncolors = 2; is_16_bit_gray = 1;} // src/image_io.cpp:493-496
int texture_t::write_to_png(string const &) const {ref_unreachable("texture_t::write_to_png"); return 0;}
// city generation and the heightmap output file name are outside the path
bool have_cities() {return 0;}
void gen_cities(float *, unsigned, unsigned) {}
string hmap_out_fn;
unsigned hmap_filter_width(0);
void get_heightmap_z_range(vector<float> const &heights, float &min_z, float &max_z) { // src/map_view.cpp:399-407 (GL-bound file)
	min_z = FLT_MAX; max_z = -FLT_MAX;
	for (unsigned i = 0; i < heights.size(); ++i) {min_eq(min_z, heights[i]); max_eq(max_z, heights[i]);}
}
void checked_fclose(FILE *fp) {if (fp) fclose(fp);}
bool open_file(FILE *&fp, char const *const fn, string const &, char const *const mode) {fp = fopen(fn, mode); return (fp != nullptr);}
void gen_scene(int, int, int, int, int) {}
void regen_lightmap() {}
void update_cpos() {}
float int_mesh_zval_pt_off(point const &, int, int, bool) {return 0.0;}
void register_timing_value(const char *, int, bool) {}
extern "C" int glutGet(unsigned) {return 0;}

// src/gen_object.cpp:377-381
double rgen_core_t::randd() {
	double rand_num;
	randome_int(rand_num);
	return rand_num/2147483563.;
}


// src/matrix_ops.cpp:59-84 (only the constants the hot path reads)
static void ref_set_scene_constants() {
	MESH_SIZE[0]  = MESH_X_SIZE; MESH_SIZE[1] = MESH_Y_SIZE; MESH_SIZE[2] = MESH_Z_SIZE;
	SCENE_SIZE[0] = X_SCENE_SIZE; SCENE_SIZE[1] = Y_SCENE_SIZE; SCENE_SIZE[2] = Z_SCENE_SIZE;
	MAX_XY_SIZE   = max(MESH_X_SIZE, MESH_Y_SIZE);
	XY_MULT_SIZE  = MESH_X_SIZE*MESH_Y_SIZE;
	XY_SUM_SIZE   = MESH_X_SIZE + MESH_Y_SIZE;
	MESH_HEIGHT   = 0.10f*Z_SCENE_SIZE;
	XY_SCENE_SIZE = 0.5f*(X_SCENE_SIZE + Y_SCENE_SIZE);
	TWO_XSS       = 2.0f*X_SCENE_SIZE;
	TWO_YSS       = 2.0f*Y_SCENE_SIZE;
	DX_VAL        = TWO_XSS/(float)MESH_X_SIZE;
	DY_VAL        = TWO_YSS/(float)MESH_Y_SIZE;
	HALF_DXY      = 0.5f*(DX_VAL + DY_VAL);
	DX_VAL_INV    = 1.0f/DX_VAL;
	DY_VAL_INV    = 1.0f/DY_VAL;
	DZ_VAL        = float(2.0f*Z_SCENE_SIZE)/(float)max(MESH_Z_SIZE, 1);
	dxdy          = DX_VAL*DY_VAL;
	MAX_RUN_DIST  = min(MESH_X_SIZE, MESH_Y_SIZE)/2;
	CLOUD_CEILING = CLOUD_CEILING0*Z_SCENE_SIZE;
	LARGE_ZVAL    = 100.0f*CLOUD_CEILING;
}

// reference functions (defined in the reference TUs) the harness calls
void create_sin_table();
void compute_scale();
void gen_rand_sine_table_entries(float scaled_height);
void estimate_zminmax(bool using_eq);
void set_zmax_est(float zval);
void gen_mesh(int surface_type, int keep_sin_table, int update_zvals);
void init_terrain_mesh();
float get_water_z_height();
float gen_noise(float xv, float yv, int mode, int shape);
float get_noise_zval(float xval, float yval, int mode, int shape);
void gen_rx_ry(float &rx, float &ry);
float eval_mesh_sin_terms(float xv, float yv);
void apply_erosion(float *heightmap, int xsize, int ysize, float min_zval, unsigned num_iters);

// ---------------------------------------------------------------------------------------------
// (3) C harness
// ---------------------------------------------------------------------------------------------
struct ref_config_t { // all scalars a config file would set for this path (src/3DWorld.cpp:1763-2110)
	int mesh_x, mesh_y;                 // mesh_size
	float scene_x, scene_y, scene_z;    // scene_size
	float mesh_height, mesh_scale;      // mesh_height (-> mesh_height_scale), mesh_scale
	int mesh_seed, mesh_freq_filter, mesh_gen_mode, mesh_gen_shape, glaciate;
	float custom_glaciate_exp;
	float hmap[14];                     // hmap_params_t in declaration order (src/mesh.h:84-88)
	float erode_amount, water_h_off, water_h_off_rel, relh_adj_tex, ocean_wave_height;
	float start_mag, start_freq, mag_mult, freq_mult;
};

struct ref_state_t { // everything derived, for injection into the system under test
	float sinTable[90][5];
	int start_eval_sin;
	float MESH_HEIGHT, DX_VAL, DY_VAL, DX_VAL_INV, DY_VAL_INV, HALF_DXY, dxdy, XY_SCENE_SIZE;
	float mesh_scale, mesh_scale_z_inv, mesh_height_scale;
	float zmax_est, zmin, zmax, water_plane_z, glaciate_exp, clip_hd1, relh_adj_tex;
	float rx, ry;
};

static vector<float> mesh_height_store;
static vector<float*> mesh_height_rows;

REF_API int ref_num_threads() {return omp_get_max_threads();}
REF_API void ref_set_num_threads(int n) {omp_set_num_threads(n);}

// Mirrors main(): create_sin_table(); set_scene_constants(); load config; init_terrain_mesh(); gen_scene()->gen_mesh()
// (src/3DWorld.cpp:2393-2460, src/build_world.cpp:628). gen_mesh() generates the 128^2 ground mesh, which is what
// seeds zmin/zmax before estimate_zminmax() -> zmax_est (src/mesh_gen.cpp:337-343,447-485).
#ifdef TERRA_ENGINE_LOOP
// ---- engine in the loop (oracle/Makefile target `engine`): this build of the harness links the PATCHED mesh_gen.cpp / erosion.cpp of oracle/engine_patch.py (INTEGRATION.md
// sections 2, 3) against a terra library.  terra_sync_globals is INTEGRATION.md section 1 verbatim: the engine hands its config values and derived globals over once after
// gen_scene(); ref_set_use_hip_terrain is the config key.  Everything else the tests call is the reference's own code: heightmap_t::proc_gen, tile_t::create_zvals ...
#include "terra_cxx.hpp"
extern bool use_hip_terrain;
extern float MESH_START_MAG, MESH_START_FREQ, MESH_MAG_MULT, MESH_FREQ_MULT;
void gen_rx_ry(float &rx, float &ry);
static void terra_sync_globals() {
	terra_config c = {};
	c.mesh_x = MESH_X_SIZE; c.mesh_y = MESH_Y_SIZE; c.scene_x = X_SCENE_SIZE; c.scene_y = Y_SCENE_SIZE; c.scene_z = Z_SCENE_SIZE;
	c.mesh_height = mesh_height_scale; c.mesh_scale = mesh_scale; c.mesh_seed = mesh_seed; c.mesh_freq_filter = mesh_freq_filter;
	c.mesh_gen_mode = mesh_gen_mode; c.mesh_gen_shape = mesh_gen_shape; c.glaciate = GLACIATE; c.custom_glaciate_exp = custom_glaciate_exp;
	memcpy(c.hmap, &hmap_params, sizeof(c.hmap));   // 14 floats, src/mesh.h:84-88
	c.erode_amount = erode_amount; c.water_h_off = water_h_off; c.water_h_off_rel = water_h_off_rel; c.relh_adj_tex = relh_adj_tex; c.ocean_wave_height = ocean_wave_height;
	c.start_mag = MESH_START_MAG; c.start_freq = MESH_START_FREQ; c.mag_mult = MESH_MAG_MULT; c.freq_mult = MESH_FREQ_MULT;
	terra_state s = {};
	memcpy(s.sinTable, sinTable, sizeof(s.sinTable)); s.start_eval_sin = start_eval_sin;
	s.MESH_HEIGHT = MESH_HEIGHT; s.DX_VAL = DX_VAL; s.DY_VAL = DY_VAL; s.DX_VAL_INV = DX_VAL_INV; s.DY_VAL_INV = DY_VAL_INV; s.HALF_DXY = HALF_DXY; s.dxdy = dxdy; s.XY_SCENE_SIZE = XY_SCENE_SIZE;
	s.mesh_scale = mesh_scale; s.mesh_scale_z_inv = mesh_scale_z_inv; s.mesh_height_scale = mesh_height_scale;
	s.zmax_est = zmax_est; s.zmin = zmin; s.zmax = zmax; s.water_plane_z = water_plane_z; s.glaciate_exp = glaciate_exp; s.clip_hd1 = clip_hd1; s.relh_adj_tex = relh_adj_tex;
	gen_rx_ry(s.rx, s.ry);
	terra_cxx::set_engine_state(c, s);
}
static void ref_init_engine(ref_config_t const *c);
REF_API void ref_init(ref_config_t const *c) { // the engine derives its globals with its own CPU path (gen_scene), THEN hands them over
	bool const hip(use_hip_terrain);
	use_hip_terrain = 0;
	ref_init_engine(c);
	use_hip_terrain = hip;
	if (hip) {terra_sync_globals();}
}
REF_API void ref_set_use_hip_terrain(int v) {use_hip_terrain = (v != 0); if (use_hip_terrain) {terra_sync_globals();}}
REF_API int  ref_get_use_hip_terrain() {return use_hip_terrain ? 1 : 0;}
extern unsigned hip_terrain_calls;
extern bool use_hip_proc_gen;
REF_API void ref_set_use_hip_proc_gen(int v) {use_hip_proc_gen = (v != 0);}
REF_API unsigned ref_hip_terrain_calls() {return hip_terrain_calls;} // build_arrays + apply_erosion calls that went through include/terra.h
static void ref_init_engine(ref_config_t const *c) {
#else
REF_API void ref_init(ref_config_t const *c) {
#endif
	MESH_X_SIZE = c->mesh_x; MESH_Y_SIZE = c->mesh_y;
	X_SCENE_SIZE = c->scene_x; Y_SCENE_SIZE = c->scene_y; Z_SCENE_SIZE = c->scene_z;
	create_sin_table();
	ref_set_scene_constants();
	mesh_height_scale = c->mesh_height; mesh_scale = c->mesh_scale;
	mesh_scale_z = 1.0; mesh_scale_z_inv = 1.0; // a config-file mesh_scale leaves these at 1; only the runtime update_mesh() (src/mesh_gen.cpp:862-874) changes them
	mesh_seed = c->mesh_seed; mesh_freq_filter = c->mesh_freq_filter; mesh_gen_mode = c->mesh_gen_mode; mesh_gen_shape = c->mesh_gen_shape;
	GLACIATE = c->glaciate; custom_glaciate_exp = c->custom_glaciate_exp;
	memcpy(&hmap_params, c->hmap, 14*sizeof(float));
	erode_amount = c->erode_amount; water_h_off = c->water_h_off; water_h_off_rel = c->water_h_off_rel; relh_adj_tex = c->relh_adj_tex;
	ocean_wave_height = c->ocean_wave_height;
	MESH_START_MAG = c->start_mag; MESH_START_FREQ = c->start_freq; MESH_MAG_MULT = c->mag_mult; MESH_FREQ_MULT = c->freq_mult;
	erosion_iters = 0; // ground-mode erosion off during init; harness calls apply_erosion explicitly
	mesh_height_store.assign(size_t(MESH_X_SIZE)*MESH_Y_SIZE, 0.0f);
	mesh_height_rows.resize(MESH_Y_SIZE);
	for (int i = 0; i < MESH_Y_SIZE; ++i) {mesh_height_rows[i] = mesh_height_store.data() + size_t(i)*MESH_X_SIZE;}
	mesh_height = mesh_height_rows.data();
	init_terrain_mesh(); // lttex_dirt zvals (src/mesh_gen.cpp:407-431)
	gen_mesh(0, 0, 1);   // sine table, ground mesh, zmax_est, water_plane_z, glaciate_exp
	gen_tex_height_tables(); // after glaciate_exp is known (gen_mesh->gen_terrain_map->glaciate sets it)
}

REF_API void ref_get_state(ref_state_t *s) {
	memcpy(s->sinTable, sinTable, sizeof(s->sinTable));
	s->start_eval_sin = start_eval_sin;
	s->MESH_HEIGHT = MESH_HEIGHT; s->DX_VAL = DX_VAL; s->DY_VAL = DY_VAL; s->DX_VAL_INV = DX_VAL_INV; s->DY_VAL_INV = DY_VAL_INV;
	s->HALF_DXY = HALF_DXY; s->dxdy = dxdy; s->XY_SCENE_SIZE = XY_SCENE_SIZE;
	s->mesh_scale = mesh_scale; s->mesh_scale_z_inv = mesh_scale_z_inv; s->mesh_height_scale = mesh_height_scale;
	s->zmax_est = zmax_est; s->zmin = zmin; s->zmax = zmax; s->water_plane_z = water_plane_z; s->glaciate_exp = glaciate_exp;
	s->clip_hd1 = clip_hd1; s->relh_adj_tex = relh_adj_tex;
	gen_rx_ry(s->rx, s->ry);
}

#ifdef TERRA_ENGINE_LOOP
#define TERRA_RESYNC if (use_hip_terrain) {terra_sync_globals();}
#else
#define TERRA_RESYNC
#endif
REF_API void ref_set_zmax_est(float v) {set_zmax_est(v); zmin = -zmax_est; zmax = zmax_est; water_plane_z = get_water_z_height(); TERRA_RESYNC}
REF_API void ref_set_water_plane_z(float v) {water_plane_z = v; TERRA_RESYNC}
REF_API void ref_set_mode(int mode, int shape) {mesh_gen_mode = mode; mesh_gen_shape = shape; TERRA_RESYNC}
REF_API void ref_set_start_eval_sin(int v) {start_eval_sin = v; TERRA_RESYNC}
REF_API void ref_set_erode_amount(float v) {erode_amount = v; TERRA_RESYNC}
REF_API void ref_get_ground_mesh(float *out) {memcpy(out, mesh_height_store.data(), mesh_height_store.size()*sizeof(float));}
REF_API float ref_sin_table(int i) {return sin_table[i];}
// read_mesh / write_mesh themselves (src/mesh_gen.cpp:895-965); zbottom / ztop are what set_zvals left
bool read_mesh(const char *filename, float zmm);
bool write_mesh(const char *filename);
extern float zbottom, ztop;
REF_API int ref_read_mesh(const char *filename, float zmm, float *zbottom_ztop) {
	bool const ok(read_mesh(filename, zmm));
	if (ok && zbottom_ztop) {zbottom_ztop[0] = zbottom; zbottom_ztop[1] = ztop;}
	if (ok) {TERRA_RESYNC}
	return ok;
}
REF_API int ref_write_mesh(const char *filename) {return write_mesh(filename);}
REF_API void ref_set_ground_mesh(const float *in) {memcpy(mesh_height_store.data(), in, mesh_height_store.size()*sizeof(float));}

// mesh_xy_grid_cache_t::build_arrays + enable_glaciate + the caller's eval_index double loop
// (src/heightmap.cpp:135-143, src/tiled_mesh.cpp:455-464,495-514)
REF_API void ref_gen_grid(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, int glaciate, int cache_values, int min_start_sin, float *out) {
	mesh_xy_grid_cache_t height_gen;
	height_gen.build_arrays(x0, y0, dx, dy, nx, ny, (cache_values != 0));
	if (glaciate) {height_gen.enable_glaciate();}
#pragma omp parallel for schedule(static,1)
	for (int y = 0; y < (int)ny; ++y) {
		for (unsigned x = 0; x < nx; ++x) {out[size_t(y)*nx + x] = height_gen.eval_index(x, y, min_start_sin);}
	}
}

REF_API void ref_gen_grid_ex(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, int glaciate, int cache_values, int force_sine_mode, int min_start_sin, int use_cache, float *out) {
	mesh_xy_grid_cache_t height_gen;
	height_gen.build_arrays(x0, y0, dx, dy, nx, ny, (cache_values != 0), (force_sine_mode != 0));
	if (glaciate) {height_gen.enable_glaciate();}
#pragma omp parallel for schedule(static,1)
	for (int y = 0; y < (int)ny; ++y) {
		for (unsigned x = 0; x < nx; ++x) {out[size_t(y)*nx + x] = height_gen.eval_index(x, y, min_start_sin, (use_cache != 0));}
	}
}

// the same over a rectangle of the grid only (the generator object is built for the whole nx x ny grid, eval_index is called inside the rectangle)
REF_API void ref_gen_grid_rect(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, int glaciate, int min_start_sin, unsigned rx0, unsigned ry0, unsigned rw, unsigned rh, float *out) {
	mesh_xy_grid_cache_t height_gen;
	height_gen.build_arrays(x0, y0, dx, dy, nx, ny);
	if (glaciate) {height_gen.enable_glaciate();}
#pragma omp parallel for schedule(static,1)
	for (int y = 0; y < (int)rh; ++y) {
		for (unsigned x = 0; x < rw; ++x) {out[size_t(y)*rw + x] = height_gen.eval_index(rx0 + x, ry0 + y, min_start_sin);}
	}
}

REF_API void ref_apply_erosion(float *hmap, int xsize, int ysize, float min_zval, unsigned iters) {apply_erosion(hmap, xsize, ysize, min_zval, iters);}
REF_API float ref_get_noise_zval(float x, float y, int mode, int shape) {return get_noise_zval(x, y, mode, shape);}
REF_API float ref_gen_noise(float x, float y, int mode, int shape) {return gen_noise(x, y, mode, shape);}
REF_API float ref_eval_mesh_sin_terms(float x, float y) {return eval_mesh_sin_terms(x, y);}
float eval_mesh_sin_terms_scaled(float xval, float yval, float xy_scale); // src/mesh_gen.cpp:807
float get_exact_zval(float xval, float yval, bool no_xyoff);               // src/mesh_gen.cpp:816
REF_API void ref_eval_points(float const *xy, unsigned n, int exact, float xy_scale, int no_xyoff, int xo2, int yo2, float *out) { // the reference's own functions, its own scroll-offset globals
	int const sx(xoff2), sy(yoff2);
	xoff2 = xo2; yoff2 = yo2;
	for (unsigned i = 0; i < n; ++i) {out[i] = (exact ? get_exact_zval(xy[2*i], xy[2*i+1], (no_xyoff != 0)) : eval_mesh_sin_terms_scaled(xy[2*i], xy[2*i+1], xy_scale));}
	xoff2 = sx; yoff2 = sy;
}
REF_API float ref_glm_simplex2(float x, float y) {return glm::simplex(glm::vec2(x, y));}
REF_API float ref_glm_perlin2 (float x, float y) {return glm::perlin (glm::vec2(x, y));}
REF_API float ref_glm_simplex3(float x, float y, float z) {return glm::simplex(glm::vec3(x, y, z));}
REF_API float ref_glm_perlin3 (float x, float y, float z) {return glm::perlin (glm::vec3(x, y, z));}
REF_API int   ref_get_bare_ls_tid_is_rock(float z) {return (get_bare_ls_tid(z) == ROCK_TEX);}
REF_API float ref_get_max_sea_level() {return (get_water_z_height() + ocean_wave_height);} // src/tiled_mesh.cpp:141

// RNG streams (src/rand_gen.h:20-35,63-79)
REF_API void ref_rand_ints(long s1, long s2, int n, int *out) {rand_gen_t r; r.set_state(s1, s2); for (int i = 0; i < n; ++i) {out[i] = r.rand();}}
REF_API void ref_rand_floats(long s1, long s2, int n, float *out) {rand_gen_t r; r.set_state(s1, s2); for (int i = 0; i < n; ++i) {out[i] = r.rand_float();}}
REF_API void ref_rand_uniforms(long s1, long s2, float a, float b, int n, float *out) {rand_gen_t r; r.set_state(s1, s2); for (int i = 0; i < n; ++i) {out[i] = r.rand_uniform(a, b);}}

// tile_t::create_zvals driver (src/tiled_mesh.cpp:467-546) for tile (tx,ty), size=128: zvals[130*130], sub_zmin/zmax[4][4], water bbox
struct ref_tile_stats_t {float sub_zmin[16], sub_zmax[16], mzmin, mzmax, radius; int wx1, wy1, wx2, wy2;};

// ---- tiles from a heightmap texture: the reference's own terrain_hmap_manager_t / heightmap_t (src/heightmap.cpp, compiled in place) over an image the
// harness copies in; modify_height_value is the override of tiled_terrain_hmap_manager_t (src/tiled_mesh.cpp:259-266, a GL-bound file) minus its tile bookkeeping
float scale_mh_texture_val(float val);
void set_mesh_height_scales_for_zval_range(float min_z, float dz);
float const SHIM_HMAP_DETAIL_SCALE = HMAP_DETAIL_SCALE, SHIM_HMAP_DETAIL_MAG = HMAP_DETAIL_MAG; // src/heightmap.h:8-9
struct shim_hmap_manager_t : public terrain_hmap_manager_t {
	void set_image(unsigned char const *pixels, int width, int height, int ncolors) {
		hmap.free_data();
		if (pixels == nullptr) return;
		hmap = heightmap_t(0, ((ncolors == 2) ? 8 : 7), width, height, "@harness", 0);
		hmap.ncolors = 1;
		if (ncolors == 2) {hmap.set_16_bit_grayscale();}
		hmap.alloc();
		memcpy(hmap.get_data(), pixels, hmap.num_bytes());
	}
	void get_image(unsigned char *out) const {memcpy(out, hmap.get_data(), hmap.num_bytes());}
	heightmap_t &image() {return hmap;}
	void clear_mods() {mod_map.clear(); brush_vect.clear();}
	unsigned num_mods() const {return mod_map.size();}
	unsigned num_brushes() const {return brush_vect.size();}
	void get_mods(mod_elem_t *mods, hmap_brush_t *brushes) const {
		unsigned n(0);
		for (tex_mod_map_t::const_iterator i = mod_map.begin(); i != mod_map.end(); ++i) {mods[n++] = mod_elem_t(*i);}
		for (unsigned i = 0; i < brush_vect.size(); ++i) {brushes[i] = brush_vect[i];}
	}
};
// The image lives in the reference's OWN manager object, `terrain_hmap_manager` (src/tiled_mesh.cpp:271; its class tiled_terrain_hmap_manager_t is local to that
// file and derives from terrain_hmap_manager_t without adding state the harness touches), so that tile_t::create_zvals / calc_mesh_ao_lighting sample it
// themselves.  The helper subclass above only adds accessors (no data members, the virtual call goes to the reference's override).
extern terrain_hmap_manager_t terrain_hmap_manager;
#define shim_hmap (*reinterpret_cast<shim_hmap_manager_t *>(&terrain_hmap_manager))
REF_API void ref_hmap_set(unsigned char const *pixels, int width, int height, int ncolors) {shim_hmap.set_image(pixels, width, height, ncolors);}
REF_API void ref_hmap_get(unsigned char *out) {shim_hmap.get_image(out);}
REF_API void ref_set_mesh_height_scales_for_zval_range(float min_z, float dz) {set_mesh_height_scales_for_zval_range(min_z, dz);}
REF_API float ref_get_clamped_height(int x, int y) {return shim_hmap.get_clamped_height(x, y);}
REF_API float ref_hmap_interpolate_height(float x, float y) {return shim_hmap.interpolate_height(x, y);}
REF_API float ref_hmap_get_nearest_height(float x, float y) {return shim_hmap.get_nearest_height(x, y);}
// rest of row f4: brushes, the mod map and its file (src/heightmap.cpp:36-58,216-308,414-440).  hmap_brush_t::apply's "omp parallel for" races on texels several
// brush points map to; the order-free result (same-sign saturating adds commute) is what one thread produces
struct ref_hmap_brush_t {int x, y; unsigned radius; int delta; short shape;};
struct ref_hmap_mod_t {unsigned short x, y; int delta;};
REF_API void ref_hmap_apply_brush(ref_hmap_brush_t const *b, int step_sz, unsigned num_steps) {
	int const nt(omp_get_max_threads());
	omp_set_num_threads(1);
	shim_hmap.apply_brush(tex_mod_map_manager_t::hmap_brush_t(b->x, b->y, b->delta, b->radius, b->shape), step_sz, num_steps);
	omp_set_num_threads(nt);
}
REF_API void ref_hmap_apply_mods(ref_hmap_mod_t const *mods, unsigned n) { // add_mod (combine per texel) + apply_cur_mod_map
	shim_hmap.clear_mods();
	for (unsigned i = 0; i < n; ++i) {shim_hmap.add_mod(tex_mod_map_manager_t::mod_elem_t(mods[i].x, mods[i].y, mods[i].delta));}
	shim_hmap.apply_cur_mod_map();
}
REF_API int ref_hmap_write_mod(char const *fn, ref_hmap_mod_t const *mods, unsigned n, ref_hmap_brush_t const *brushes, unsigned nb) {
	static_assert(sizeof(ref_hmap_brush_t) == sizeof(tex_mod_map_manager_t::hmap_brush_t) && sizeof(ref_hmap_mod_t) == sizeof(tex_mod_map_manager_t::mod_elem_t), "layout");
	shim_hmap.clear_mods();
	for (unsigned i = 0; i < n; ++i) {shim_hmap.add_mod(tex_mod_map_manager_t::mod_elem_t(mods[i].x, mods[i].y, mods[i].delta));}
	for (unsigned i = 0; i < nb; ++i) {shim_hmap.add_brush(tex_mod_map_manager_t::hmap_brush_t(brushes[i].x, brushes[i].y, brushes[i].delta, brushes[i].radius, brushes[i].shape));}
	return shim_hmap.write_mod(fn);
}
REF_API int ref_hmap_read_mod(char const *fn, ref_hmap_mod_t *mods, unsigned *n, ref_hmap_brush_t *brushes, unsigned *nb) { // mods / brushes null: counts only
	if (!shim_hmap.read_mod(fn)) return 0;
	*n = shim_hmap.num_mods(); *nb = shim_hmap.num_brushes();
	if (mods && brushes) {shim_hmap.get_mods((tex_mod_map_manager_t::mod_elem_t *)mods, (tex_mod_map_manager_t::hmap_brush_t *)brushes);}
	return 1;
}
REF_API int ref_hmap_read_and_apply_mod(char const *fn) {
	int const nt(omp_get_max_threads());
	omp_set_num_threads(1);
	int const ret(shim_hmap.read_and_apply_mod(fn));
	omp_set_num_threads(nt);
	return ret;
}
// heightmap_t::proc_gen itself (src/heightmap.cpp:130-151): width x height 16-bit map with erosion_iters_tt droplets; returns the pixels and mesh_file_scale / tz
extern float mesh_file_scale, mesh_file_tz;
REF_API void ref_heightmap_proc_gen(int width, int height, unsigned iters, unsigned char *pixels, float *file_scale_tz) {
	unsigned const prev(erosion_iters_tt);
	erosion_iters_tt = iters;
	heightmap_t hm(0, 8, width, height, "@tt_heightmap", 0);
	hm.proc_gen();
	memcpy(pixels, hm.get_data(), hm.num_bytes());
	hm.free_data();
	file_scale_tz[0] = mesh_file_scale; file_scale_tz[1] = mesh_file_tz;
	erosion_iters_tt = prev;
}

// rest of row a12: the reference's own heightmap_t::to_floats / from_floats (private members: this file is compiled with -fno-access-control) and
// postprocess_height (src/heightmap.cpp:117-128,191-215) on an image the harness copies in; mesh_file_scale / mesh_file_tz as the config line
// `mh_filename <png> <scale> <tz>` sets them (src/3DWorld.cpp:2205).  from_floats / postprocess_height assert on values outside [0, 256): the harness counts
// them first and then lets the member run only when there are none (the count is what the oracle and the library report).
REF_API void ref_set_mesh_file_scale(float scale, float tz) {mesh_file_scale = scale; mesh_file_tz = tz;}
static void shim_load_image(heightmap_t &hm, unsigned char const *pixels, int ncolors) {
	hm.ncolors = 1;
	if (ncolors == 2) {hm.set_16_bit_grayscale();}
	hm.alloc();
	memcpy(hm.get_data(), pixels, hm.num_bytes());
}
REF_API void ref_heightmap_to_floats(unsigned char const *pixels, int width, int height, int ncolors, float *vals) {
	heightmap_t hm(0, ((ncolors == 2) ? 8 : 7), width, height, "@harness", 0);
	shim_load_image(hm, pixels, ncolors);
	vector<float> v;
	hm.to_floats(v);
	memcpy(vals, v.data(), v.size()*sizeof(float));
	hm.free_data();
}
static unsigned shim_count_out_of_range(float const *vals, size_t n) {
	float const val_div(1.0/get_mh_texture_mult()), val_add(get_mh_texture_add());
	unsigned bad(0);
	for (size_t i = 0; i < n; ++i) {float const v((vals[i] - val_add)*val_div); if (!(v >= 0.0 && v < 256.0)) {++bad;}}
	return bad;
}
REF_API unsigned ref_heightmap_from_floats(float const *vals, int width, int height, int ncolors, unsigned char *pixels) {
	size_t const n(size_t(width)*height);
	unsigned const bad(shim_count_out_of_range(vals, n));
	if (bad) return bad; // the member would assert
	heightmap_t hm(0, ((ncolors == 2) ? 8 : 7), width, height, "@harness", 0);
	hm.ncolors = 1;
	if (ncolors == 2) {hm.set_16_bit_grayscale();}
	hm.alloc();
	vector<float> v(vals, vals + n);
	hm.from_floats(v);
	memcpy(pixels, hm.get_data(), hm.num_bytes());
	hm.free_data();
	return 0;
}
REF_API unsigned ref_heightmap_postprocess(unsigned char *pixels, int width, int height, int ncolors, unsigned iters_tt) {
	unsigned const prev(erosion_iters_tt);
	int const nt(omp_get_max_threads());
	omp_set_num_threads(1); // apply_erosion's OpenMP loop is a data race: the serial droplet order is the only defined result
	erosion_iters_tt = iters_tt;
	heightmap_t hm(0, ((ncolors == 2) ? 8 : 7), width, height, "@harness", 0);
	shim_load_image(hm, pixels, ncolors);
	unsigned bad(0);
	{ // dry run of to_floats + run_erosion to see whether from_floats would assert
		vector<float> v;
		hm.to_floats(v);
		if (iters_tt > 0) {hm.run_erosion(v);}
		bad = shim_count_out_of_range(v.data(), v.size());
	}
	if (bad == 0) {hm.postprocess_height(); memcpy(pixels, hm.get_data(), hm.num_bytes());}
	hm.free_data();
	erosion_iters_tt = prev;
	omp_set_num_threads(nt);
	return bad;
}

// ---------------------------------------------------------------------------------------------
// (4) the tile functions: the reference's OWN tile_t members (src/tiled_mesh.cpp compiled in place; get_tids / update_lttex_ix / gen_tex_height_tables
//     from src/Textures.cpp compiled in place).  The harness builds a tile_t, calls the member, copies the member data out.  GL entry points the members
//     end with (texture uploads) are no-ops below; glTexImage2D hands the uploaded bytes to the harness (that is how the normal map, which
//     upload_normal_texture builds in a local vector, comes back).  Globals of other subsystems the members read are defined with the reference's defaults.
// ---------------------------------------------------------------------------------------------
bool enable_tiled_mesh_ao(0);       // src/3DWorld.cpp:73,1778
bool add_city_grass(0), water_is_lava(0); // src/3DWorld.cpp
int DISABLE_WATER(0);
float vegetation(1.0), biome_x_offset(0.0);
unsigned grass_density(0), num_rnd_grass_blocks(16); // src/grass.cpp:14
extern unsigned erosion_iters_tt;
extern bool enable_terrain_env;     // src/tiled_mesh.cpp:82
REF_API void ref_set_tiled_mesh_ao(int v) {enable_tiled_mesh_ao = (v != 0);}

// cities / tunnels / buildings / the disabled-mesh mask: other subsystems' inputs of create_zvals / create_texture, absent here (terrain-only branch)
int  check_city_contains_overlaps(cube_t const &) {return 0;}
bool check_mesh_disable(point const &, float) {return 0;}
bool check_inside_city(point const &, float) {return 0;}
bool city_has_grass_at(point const &, float, cube_t &) {return 0;}
bool tile_contains_tunnel(cube_t const &) {return 0;}
bool no_grass_under_buildings() {return 0;}
bool check_buildings_no_grass(point const &) {return 0;}
cube_t get_city_grass_bcube_at(cube_t const &) {return cube_t();}
void get_city_grass_coll_cubes(cube_t const &, vect_cube_t &, vect_cube_t &) {}
void get_building_grass_coll_cubes(cube_t const &, vect_cube_t &) {}
// GL: uploads are no-ops; the last 2-D upload's pixels are kept for the harness
static void const *ref_last_upload = nullptr; static int ref_last_upload_w = 0, ref_last_upload_h = 0;
extern "C" {
}
static vector<unsigned char> ref_upload_copy; // RGBA8 uploads only (what the members under test upload)
static void ref_keep_upload(GLsizei w, GLsizei h, void const *pixels) {ref_last_upload = pixels; ref_last_upload_w = w; ref_last_upload_h = h; if (pixels) {ref_upload_copy.assign((unsigned char const *)pixels, (unsigned char const *)pixels + (size_t)4*w*h);}}
extern "C" {
void glTexImage2D(GLenum, GLint, GLint, GLsizei w, GLsizei h, GLint, GLenum, GLenum, const void *pixels) {ref_keep_upload(w, h, pixels);}
void glTexSubImage2D(GLenum, GLint, GLint, GLint, GLsizei w, GLsizei h, GLenum, GLenum, const void *pixels) {ref_keep_upload(w, h, pixels);}
}
// setup_texture / bind_2d_texture are the reference's own (src/Textures.cpp); what they call in libGL does nothing here
extern "C" {
void glGenTextures(GLsizei n, GLuint *t) {for (GLsizei i = 0; i < n; ++i) {t[i] = 1;}}
void glBindTexture(GLenum, GLuint) {}
void glDeleteTextures(GLsizei, const GLuint *) {}
GLboolean glIsTexture(GLuint) {return 1;}
void glPixelStorei(GLenum, GLint) {}
void glTexParameterf(GLenum, GLenum, GLfloat) {}
void glTexParameteri(GLenum, GLenum, GLint) {}
int gluScaleImage(unsigned, int, int, unsigned, const void *, int, int, unsigned, void *) {ref_unreachable("gluScaleImage"); return 0;}
const unsigned char *gluErrorString(unsigned) {return (const unsigned char *)"";}
}
int omp_get_thread_num_3dw() {return omp_get_thread_num();} // src/3DWorld.cpp
bool check_gl_error(unsigned) {return 0;}                   // src/gl_ext_arb.cpp: there is no GL context to ask

struct ref_landscape_t {float vegetation, temperature, biome_x_offset, mesh_scale_z; int water_is_lava, disable_water, enable_terrain_env; unsigned grass_density, num_rnd_grass_blocks;};
struct ref_grass_block_t {unsigned ix; float zmin, zmax;}; // tile_t::grass_block_t (src/tiled_mesh.h:186)
REF_API void ref_set_landscape(ref_landscape_t const *p) {
	vegetation = p->vegetation; temperature = p->temperature; biome_x_offset = p->biome_x_offset; mesh_scale_z = p->mesh_scale_z;
	water_is_lava = (p->water_is_lava != 0); DISABLE_WATER = p->disable_water; enable_terrain_env = (p->enable_terrain_env != 0);
	grass_density = p->grass_density; num_rnd_grass_blocks = p->num_rnd_grass_blocks;
	init_terrain_mesh(); // calls gen_tex_height_tables()
}

// a tile_t whose heavy members (trees, scenery, clouds: other subsystems, constructors in translation units that are not built) are never constructed:
// zero-filled storage -- what a value-initialised tile_t of empty containers is -- with the scalar members set as tile_t::tile_t(size, x, y) sets them
// (src/tiled_mesh.cpp:302-314) and the default member initialisers of the members the functions under test read
struct ref_tile_box_t {
	tile_t *t;
	ref_tile_box_t(int tx, int ty) {
		t = (tile_t *)calloc(1, sizeof(tile_t));
		unsigned const size(128);
		t->size = size; t->stride = size + 1; t->zvsize = size + 2;
		t->x1 = tx*(int)size; t->y1 = ty*(int)size; t->x2 = t->x1 + size; t->y2 = t->y1 + size;
		t->wx1 = t->x2; t->wy1 = t->y2; t->wx2 = t->x1; t->wy2 = t->y1; // start denormalized
		t->mesh_off.set_from_xyoff2(); // tile_offset_t(xoff-xoff2, yoff-yoff2) with all four 0
		t->xstart = get_xval(t->x1 + t->mesh_off.dxoff); t->ystart = get_yval(t->y1 + t->mesh_off.dyoff);
		t->radius = t->calc_radius();
		t->mzmin = t->mzmax = t->ptzmax = t->dtzmax = 0.0; // get_camera_pos().z: overwritten by create_zvals
		t->base_tsize = 512; // NORM_TEXELS (src/tiled_mesh.cpp)
		t->sun_shadows_invalid = t->moon_shadows_invalid = t->recalc_tree_grass_weights = 1;
		for (unsigned i = 0; i < 4; ++i) {tile_t::terrain_params_t &p(t->params[i>>1][i&1]); p.hoff = 0.0; p.hscale = 1.0; p.veg = 1.0; p.grass = 1.0; p.dirt = 0.0;}
	}
	~ref_tile_box_t() { // only the containers the harness / the functions under test filled
		typedef vector<float> vf; typedef vector<unsigned char> vu;
		t->zvals.~vf(); t->ao_zvals.~vf(); t->mesh_weight_data.~vu(); t->weight_data.~vu(); t->ao_lighting.~vu();
		t->grass_blocks.~vector<tile_t::grass_block_t>();
		free(t);
	}
	void set_zvals(float const *z) {t->zvals.assign(z, z + t->zvsize*t->zvsize);}
};

REF_API void ref_tile_create_zvals(int tx, int ty, unsigned iters_tt, float *zvals, ref_tile_stats_t *st) {
	unsigned const prev(erosion_iters_tt);
	erosion_iters_tt = iters_tt;
	ref_tile_box_t b(tx, ty);
	mesh_xy_grid_cache_t height_gen;
	bool const ok(b.t->create_zvals(height_gen, 0)); // tile_t::create_zvals itself (src/tiled_mesh.cpp:467-546)
	erosion_iters_tt = prev;
	if (!ok) {ref_unreachable("tile_t::create_zvals returned 0 without no_wait");}
	memcpy(zvals, b.t->zvals.data(), b.t->zvals.size()*sizeof(float));
	if (st) {
		for (unsigned i = 0; i < 16; ++i) {st->sub_zmin[i] = b.t->sub_zmin[i>>2][i&3]; st->sub_zmax[i] = b.t->sub_zmax[i>>2][i&3];}
		st->mzmin = b.t->mzmin; st->mzmax = b.t->mzmax; st->radius = b.t->radius;
		st->wx1 = b.t->wx1; st->wy1 = b.t->wy1; st->wx2 = b.t->wx2; st->wy2 = b.t->wy2;
	}
}
// tile_t::calc_mesh_ao_lighting itself (src/tiled_mesh.cpp:586-661) on a tile whose zvals are the caller's (eroded or not).  With enable_tiled_mesh_ao and a GL
// noise mode the engine's create_zvals has left the 201^2 context in ao_zvals: reproduced by running create_zvals first, exactly as the engine does.
REF_API void ref_tile_ao_lighting(int tx, int ty, float const *zvals, unsigned char *ao) {
	ref_tile_box_t b(tx, ty);
	if (enable_tiled_mesh_ao && !using_tiled_terrain_hmap_tex() && mesh_gen_mode >= MGEN_SIMPLEX_GPU) {
		unsigned const prev(erosion_iters_tt);
		erosion_iters_tt = 0;
		mesh_xy_grid_cache_t height_gen;
		b.t->create_zvals(height_gen, 0);
		erosion_iters_tt = prev;
	}
	b.set_zvals(zvals);
	b.t->calc_mesh_ao_lighting();
	memcpy(ao, b.t->ao_lighting.data(), b.t->ao_lighting.size());
}
// tile_t::update_terrain_params itself (src/tiled_mesh.cpp:321-343): out[yp][xp] = {veg, grass, dirt}
REF_API void ref_tile_terrain_params(int tx, int ty, float *out) {
	ref_tile_box_t b(tx, ty);
	if (enable_terrain_env) {b.t->update_terrain_params();}
	for (unsigned i = 0; i < 4; ++i) {tile_t::terrain_params_t const &p(b.t->params[i>>1][i&1]); out[3*i] = p.veg; out[3*i+1] = p.grass; out[3*i+2] = p.dirt;}
}
// tile_t::create_texture itself (src/tiled_mesh.cpp:1071-1240, with get_tids / update_lttex_ix from src/Textures.cpp:1289-1316 and add_grass_block_at :1354-1371)
extern vector<texture_t> textures; // src/Textures.cpp:177: the renderer's texture table, filled by load_textures() in the engine
REF_API void ref_tile_create_weights(int tx, int ty, float const *zvals, unsigned char *mesh_weight_data, ref_grass_block_t *grass_blocks, int *has_any_grass_out) {
	// create_or_update_weight_tex (the tail of create_texture) averages the landscape textures' colours into avg_mesh_tex_color, a render-only value nobody compares:
	// give it default entries to read instead of the image files the engine loads
	if (textures.size() < 256) {textures.resize(256);} // the ids of the built-in texture enum (src/3DWorld.h:1380-1394) are far below 256
	ref_tile_box_t b(tx, ty);
	b.set_zvals(zvals);
	if (enable_terrain_env) {b.t->update_terrain_params();} // what create_zvals does before (src/tiled_mesh.cpp:471)
	int const nt(omp_get_max_threads());
	mesh_xy_grid_cache_t height_gen;
	b.t->create_texture(height_gen);
	omp_set_num_threads(nt);
	memcpy(mesh_weight_data, b.t->mesh_weight_data.data(), b.t->mesh_weight_data.size());
	if (grass_blocks) {
		unsigned const n(32*32);
		memset(grass_blocks, 0, n*sizeof(ref_grass_block_t));
		for (unsigned i = 0; i < b.t->grass_blocks.size() && i < n; ++i) {grass_blocks[i].ix = b.t->grass_blocks[i].ix; grass_blocks[i].zmin = b.t->grass_blocks[i].zmin; grass_blocks[i].zmax = b.t->grass_blocks[i].zmax;}
	}
	if (has_any_grass_out) {*has_any_grass_out = b.t->has_any_grass;}
}
// tile_t::upload_normal_texture itself (src/tiled_mesh.cpp:865-880, get_norm src/tiled_mesh.h:281-284): the RGBA bytes it hands to GL + min_normal_z
REF_API float ref_tile_normals(float const *zvals, unsigned char *rgba /*129*129*4*/) {
	ref_tile_box_t b(0, 0);
	b.set_zvals(zvals);
	ref_last_upload = nullptr;
	b.t->upload_normal_texture(0);
	// upload_normal_texture's vector is gone by now; glTexImage2D was called while it was alive -- copy there instead
	memcpy(rgba, ref_upload_copy.data(), 4*129*129);
	return b.t->min_normal_z;
}

// ---- row f2: mesh shadows.  calc_mesh_shadows / mesh_shadow_gen are the reference's own code (src/visibility.cpp:411-517, do_line_clip from src/Math3d.cpp:1070);
// the chaining of sh_in / sh_out between adjacent tiles restates tile_t::calc_shadows_for_light (src/tiled_mesh.cpp:664-692).
void calc_mesh_shadows(unsigned l, point const &lpos, float const *const mh, unsigned char *smask, int xsize, int ysize,
	float const *sh_in_x, float const *sh_in_y, float *sh_out_x, float *sh_out_y);
REF_API void ref_calc_mesh_shadows(float lx, float ly, float lz, float const *mh, unsigned char *smask, int xsize, int ysize,
	float const *sh_in_x, float const *sh_in_y, float *sh_out_x, float *sh_out_y)
{
	// mesh_shadow_gen::run() asks for two threads explicitly ("#pragma omp parallel sections num_threads(2)": the X and the Y sweeps), which race on
	// smask / sh_out.  Called from one thread of an ACTIVE (two-thread) parallel region with max-active-levels = 1, the nested region is
	// serialised and its sections run in order: the single-threaded semantics.  (A one-thread outer region would not count as active.)
	omp_set_max_active_levels(1);
#pragma omp parallel num_threads(2)
	{
		if (omp_get_thread_num() == 0) {calc_mesh_shadows(LIGHT_SUN, point(lx, ly, lz), mh, smask, xsize, ysize, sh_in_x, sh_in_y, sh_out_x, sh_out_y);}
	}
}
// a batch of tiles: every tile pulls its inputs from the neighbours toward the light that are part of the batch (computed first, like the recursion in :683)
REF_API void ref_tiles_mesh_shadows(int const *tile_xy, unsigned n, float const *zvals, float lx, float ly, float lz, unsigned char *smask) {
	unsigned const zvsize(130);
	point const lpos(lx, ly, lz);
	int const sx((lpos.x < 0.0) ? -1 : 1), sy((lpos.y < 0.0) ? -1 : 1); // toward the light source
	std::map<std::pair<int,int>, unsigned> ix;
	for (unsigned i = 0; i < n; ++i) {ix[std::make_pair(tile_xy[2*i], tile_xy[2*i+1])] = i;}
	vector<vector<float>> sh_out[2];
	sh_out[0].resize(n); sh_out[1].resize(n);
	vector<char> done(n, 0);
	std::function<void(unsigned)> run = [&](unsigned i) {
		if (done[i]) return;
		done[i] = 1;
		float const *sh_in[2] = {0, 0};
		int const adj[2][2] = {{tile_xy[2*i] + sx, tile_xy[2*i+1]}, {tile_xy[2*i], tile_xy[2*i+1] + sy}};
		for (unsigned d = 0; d < 2; ++d) {
			sh_out[!d][i].assign(zvsize, MESH_MIN_Z);
			auto it(ix.find(std::make_pair(adj[d][0], adj[d][1])));
			if (it == ix.end()) continue; // no adjacent tile
			run(it->second); // recursive call on adjacent tile
			sh_in[!d] = sh_out[!d][it->second].data();
		}
		calc_mesh_shadows(LIGHT_SUN, lpos, zvals + (size_t)i*zvsize*zvsize, smask + (size_t)i*zvsize*zvsize, zvsize, zvsize, sh_in[0], sh_in[1], sh_out[0][i].data(), sh_out[1][i].data());
	};
	omp_set_max_active_levels(1); // see ref_calc_mesh_shadows
#pragma omp parallel num_threads(2)
	{
		if (omp_get_thread_num() == 0) {for (unsigned i = 0; i < n; ++i) {run(i);}}
	}
}

// tile_t::upload_normal_texture CPU part (src/tiled_mesh.cpp:865-880, src/tiled_mesh.h:281-284); returns min_normal_z
// heightmap_t::proc_gen tail (src/heightmap.cpp:146-150,205-215; src/Textures.cpp:1889-1893; src/mesh_gen.cpp:120-131):
// z-range -> 16-bit quantise. out = 2 bytes per pixel, [lo, hi].
REF_API void ref_quantize16(float const *vals, size_t n, unsigned char *out, float *min_z_out, float *dz_out) {
	float min_z(vals[0]), max_z(vals[0]);
	for (size_t i = 0; i < n; ++i) {min_z = min(min_z, vals[i]); max_z = max(max_z, vals[i]);}
	float const dz(max(TOLERANCE, (max_z - min_z)));
	float const READ_MESH_H_SCALE(0.0008);
	// set_mesh_height_scales_for_zval_range(min_z, dz/255.0)
	float const dzs(dz/255.0);
	float const file_scale(dzs/(READ_MESH_H_SCALE*mesh_height_scale*mesh_scale_z_inv)), file_tz(min_z/mesh_scale_z_inv);
	float const mult(READ_MESH_H_SCALE*mesh_height_scale*file_scale*mesh_scale_z_inv), add(file_tz*mesh_scale_z_inv); // get_mh_texture_mult/add
	float const val_div(1.0/mult), val_add(add);
	for (size_t i = 0; i < n; ++i) {
		float v((vals[i] - val_add)*val_div);
		unsigned char const high_bits(v);
		out[(i<<1)+1] = high_bits;
		out[i<<1]     = (unsigned char)(256.0f*(v - float(high_bits)));
	}
	*min_z_out = min_z; *dz_out = dz;
}

// write_map_mode_heightmap_image (src/map_view.cpp:409-442, a GL-bound file) from the image origin on: driver over the reference's build_arrays /
// eval_index / terrain_hmap_manager_t::interpolate_height / get_heightmap_z_range / write_pixel_16_bits; setup_height_gen_cached = src/tiled_mesh.cpp:452-457,
// get_mesh_height = src/map_view.cpp:97-105.  pixels: 2 bytes per pixel; min_z_dz = {min_z, dz}
void setup_height_gen_cached(mesh_xy_grid_cache_t &height_gen, float x0, float y0, float dx, float dy, unsigned nx, unsigned ny);
bool using_tiled_terrain_hmap_tex();
bool using_hmap_with_detail();
REF_API void ref_export_heightmap(float xstart, float ystart, int width, int height, unsigned char *pixels, float *min_z_dz) {
	texture_t texture(0, 6, width, height, 0, 2, 0, "heightmap.png"); // two bytes per pixel grayscale
	texture.set_16_bit_grayscale();
	texture.alloc();
	vector<float> heights(texture.num_pixels());
	mesh_xy_grid_cache_t height_gen;
	setup_height_gen_cached(height_gen, xstart, ystart, DX_VAL, DY_VAL, width, height); // the reference's own (src/tiled_mesh.cpp:452-457)
	float const xscale(DX_VAL), yscale(DY_VAL);
#pragma omp parallel for schedule(static,1)
	for (int i = 0; i < height; ++i) {
		int const off(width*(height - i - 1)); // invert yval
		for (int j = 0; j < width; ++j) {
			float zval;
			if (using_tiled_terrain_hmap_tex()) {
				zval = shim_hmap.interpolate_height((xstart + X_SCENE_SIZE + j*xscale)*DX_VAL_INV, (ystart + Y_SCENE_SIZE + i*yscale)*DY_VAL_INV);
				if (using_hmap_with_detail()) {zval += HMAP_DETAIL_MAG*height_gen.eval_index(j, i);}
			}
			else {zval = height_gen.eval_index(j, i);}
			heights[off + j] = zval;
		}
	}
	float min_z(0), max_z(0);
	get_heightmap_z_range(heights, min_z, max_z);
	float const dz(max(TOLERANCE, (max_z - min_z))), height_scale(255.0/dz); // prevent divide-by-zero
	for (unsigned i = 0; i < heights.size(); ++i) {texture.write_pixel_16_bits(i, (heights[i] - min_z)*height_scale);}
	memcpy(pixels, texture.get_data(), texture.num_bytes());
	texture.free_data();
	min_z_dz[0] = min_z; min_z_dz[1] = dz;
}

// voxel_manager::create_procedural fill loop (src/voxels.cpp:278-345) around the real noise_gen_3d (src/upsurface.cpp:16-70).
// out is z-fastest: ix = z + (x + y*nx)*nz  (src/voxels.h:141-144)
REF_API void ref_voxel_fill(float *out, unsigned nx, unsigned ny, unsigned nz, float const lo_pos[3], float const vsz[3], float const offset[3],
	float mag, float freq, int rseed1, int rseed2, int gen_mode, float zscale, int normalize_to_1)
{
	unsigned const xyz_num[3] = {nx, ny, nz};
	vector<float> xyz_vals[3];
	noise_gen_3d ngen;
	float rx(0.0), ry(0.0);
	point const lo(lo_pos[0], lo_pos[1], lo_pos[2]);
	vector3d const off(offset[0], offset[1], offset[2]), step(vsz[0], vsz[1], vsz[2]);

	if (gen_mode == MGEN_SINE) {
		ngen.set_rand_seeds(rseed1, rseed2);
		ngen.gen_sines(mag, freq);
		ngen.gen_xyz_vals((lo + off), step, xyz_num, xyz_vals);
	}
	else {gen_rx_ry(rx, ry);}
#pragma omp parallel for schedule(static,1)
	for (int y = 0; y < (int)ny; ++y) {
		for (unsigned x = 0; x < nx; ++x) {
			for (unsigned z = 0; z < nz; ++z) {
				float val(0.0);
				if (gen_mode == MGEN_SINE) {val = ngen.get_val(x, y, z, xyz_vals);}
				else {
					point const pos(point((x*step.x + lo.x), (y*step.y + lo.y), (z*step.z + lo.z)) + off); // get_pt_at (src/voxels.h:126) + offset
					glm::vec3 const v(pos.x, pos.y, pos.z);
					float nmag(mag), nfreq(0.25*freq);
					float const lacunarity(1.92), gain(0.5);
					for (int n = 0; n < max(1, (5 - mesh_freq_filter)); ++n) { // MAX_FREQ_BINS = 5 (src/upsurface.h)
						glm::vec3 const nv(nfreq*v + glm::vec3(rx, ry, rx-ry));
						val   += nmag*((gen_mode == MGEN_PERLIN) ? glm::perlin(nv) : glm::simplex(nv));
						nmag  *= gain;
						nfreq *= lacunarity;
					}
				}
				val += z*zscale;
				if (normalize_to_1) {val = CLIP_TO_pm1(val);}
				out[z + (x + size_t(y)*nx)*nz] = val;
			}
		}
	}
}
REF_API void ref_voxel_rdata(int rseed1, int rseed2, float mag, float freq, float *rdata /*420*/) {
	noise_gen_3d ngen;
	ngen.set_rand_seeds(rseed1, rseed2);
	ngen.gen_sines(mag, freq);
	memcpy(rdata, ngen.rdata, sizeof(ngen.rdata));
}
