// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Link shim that lets the reference's OWN translation units
//     /root/reference/src/mesh_gen.cpp, erosion.cpp, upsurface.cpp, visibility.cpp, Math3d.cpp, heightmap.cpp   (+ vendored glm 0.9.9.1 headers)
// be compiled unmodified, in place, into oracle/_ref/liboracle_ref.so (recipe: oracle/Makefile).
// No reference source is copied: this file only (1) DEFINES the process globals those TUs declare
// `extern` (they live in 3DWorld.cpp / Textures.cpp / display_world.cpp / Universe.cpp, which cannot be
// built without OpenGL), (2) STUBS the GL/IO entry points they reference but that the CPU path never
// calls, and (3) exports a plain C harness ("ref_*") that ctypes can drive.
//
// The handful of functions that live in TUs we cannot build are restated here, each with its citation:
//   set_scene_constants   src/matrix_ops.cpp:59-84
//   get_bare_ls_tid       src/Textures.cpp:1284-1287
//   gen_tex_height_tables src/Textures.cpp:1757-1761
//   rgen_core_t::randd    src/gen_object.cpp:377-381
//   the voxel fill loop   src/voxels.cpp:312-345       (voxels.o has >100 unrelated externals)
//   tile_t::create_zvals  src/tiled_mesh.cpp:467-546   (driver only; generator + erosion are the real TUs)
//   tile_t::get_norm / upload_normal_texture  src/tiled_mesh.h:281-284, src/tiled_mesh.cpp:865-880
//   texture_t::alloc / free_client_mem / set_16_bit_grayscale / write_pixel_16_bits  src/Textures.cpp:486-517,1889-1893, src/image_io.cpp:493-496 (for heightmap.cpp)
//   tile_t::calc_mesh_ao_lighting, create_texture, update_terrain_params; get_tids; write_map_mode_heightmap_image  (drivers, see each)

#include "3DWorld.h"
#include <map>
#include <functional>
#include "mesh.h"
#include "textures.h"
#include "heightmap.h"
#include "shaders.h"
#include "upsurface.h"
#include "sinf.h"
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
float h_dirt[NTEX_DIRT], clip_hd1;
float ocean_wave_height(0.0); // reference default is DEF_OCEAN_WAVE_HEIGHT; harness sets it explicitly

extern float zmin, zmax, zmax_est, glaciate_exp, mesh_scale, mesh_scale_z, mesh_scale_z_inv, mesh_height_scale;
extern int start_eval_sin, GLACIATE, mesh_gen_mode, mesh_gen_shape, mesh_freq_filter;
extern float sinTable[][5];
extern float MESH_START_MAG, MESH_START_FREQ, MESH_MAG_MULT, MESH_FREQ_MULT;
extern hmap_params_t hmap_params;
extern ttex lttex_dirt[];

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
void texture_t::resize(int, int) {ref_unreachable("texture_t::resize");}
void texture_t::gl_delete() {}
// texture_t client-memory management and 16-bit pixel writer, needed by the reference's heightmap.cpp (src/Textures.cpp:486-490,512-517,1889-1893; src/image_io.cpp:493-496)
void texture_t::alloc() {free_data(); data = new unsigned char[num_bytes()];}
void texture_t::free_client_mem() {
	if (orig_data    != data) {delete [] orig_data;}
	if (colored_data != data) {delete [] colored_data;}
	delete [] data;
	data = orig_data = colored_data = NULL;
}
void texture_t::set_16_bit_grayscale() {ncolors = 2; is_16_bit_gray = 1;}
void texture_t::write_pixel_16_bits(unsigned ix, float val) {
	unsigned char const high_bits(val); // high bits - truncate
	data[(ix<<1)+1] = high_bits;
	data[ix<<1]     = (unsigned char)(256.0f*(val - float(high_bits))); // low bits - remainder
}
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
void free_texture(unsigned &tid) {tid = 0;}
void checked_fclose(FILE *fp) {if (fp) fclose(fp);}
bool open_file(FILE *&fp, char const *const fn, string const &, char const *const mode) {fp = fopen(fn, mode); return (fp != nullptr);}
void gen_scene(int, int, int, int, int) {}
void regen_lightmap() {}
void update_cpos() {}
float int_mesh_zval_pt_off(point const &, int, int, bool) {return 0.0;}
bool using_hmap_with_detail() {return 0;}
bool using_tiled_terrain_hmap_tex() {return 0;}
float get_tiled_terrain_height_tex(float, float, bool) {return 0.0;}
void register_timing_value(const char *, int, bool) {}
extern "C" int glutGet(unsigned) {return 0;}

// src/gen_object.cpp:377-381
double rgen_core_t::randd() {
	double rand_num;
	randome_int(rand_num);
	return rand_num/2147483563.;
}

// src/Textures.cpp:1757-1761
void gen_tex_height_tables() {
	for (unsigned i = 0; i < NTEX_DIRT; ++i) {h_dirt[i] = pow(lttex_dirt[i].zval, glaciate_exp);}
	clip_hd1 = (0.90*h_dirt[1] + 0.10*h_dirt[0]);
}
// src/Textures.cpp:1284-1287
int get_bare_ls_tid(float zval) {
	float const relh(relh_adj_tex + (zval - zmin)/(zmax - zmin));
	return ((relh > clip_hd1) ? (int)ROCK_TEX : (int)DIRT_TEX); // rock or dirt
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
REF_API void ref_init(ref_config_t const *c) {
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

REF_API void ref_set_zmax_est(float v) {set_zmax_est(v); zmin = -zmax_est; zmax = zmax_est; water_plane_z = get_water_z_height();}
REF_API void ref_set_water_plane_z(float v) {water_plane_z = v;}
REF_API void ref_set_mode(int mode, int shape) {mesh_gen_mode = mode; mesh_gen_shape = shape;}
REF_API void ref_set_start_eval_sin(int v) {start_eval_sin = v;}
REF_API void ref_set_erode_amount(float v) {erode_amount = v;}
REF_API void ref_get_ground_mesh(float *out) {memcpy(out, mesh_height_store.data(), mesh_height_store.size()*sizeof(float));}
REF_API float ref_sin_table(int i) {return sin_table[i];}

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

REF_API void ref_apply_erosion(float *hmap, int xsize, int ysize, float min_zval, unsigned iters) {apply_erosion(hmap, xsize, ysize, min_zval, iters);}
REF_API float ref_get_noise_zval(float x, float y, int mode, int shape) {return get_noise_zval(x, y, mode, shape);}
REF_API float ref_gen_noise(float x, float y, int mode, int shape) {return gen_noise(x, y, mode, shape);}
REF_API float ref_eval_mesh_sin_terms(float x, float y) {return eval_mesh_sin_terms(x, y);}
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
	virtual bool modify_height_value(int x, int y, hmap_val_t val, bool is_delta, float fract_x, float fract_y, bool allow_wrap=1) {
		int clamped_x(x), clamped_y(y);
		if (!clamp_xy(clamped_x, clamped_y, fract_x, fract_y, allow_wrap)) return 0;
		assert(clamped_x >= 0 && clamped_y >= 0);
		modify_height(tex_mod_map_manager_t::mod_elem_t(clamped_x, clamped_y, val), is_delta);
		return 1;
	}
	void clear_mods() {mod_map.clear(); brush_vect.clear();}
	unsigned num_mods() const {return mod_map.size();}
	unsigned num_brushes() const {return brush_vect.size();}
	void get_mods(mod_elem_t *mods, hmap_brush_t *brushes) const {
		unsigned n(0);
		for (tex_mod_map_t::const_iterator i = mod_map.begin(); i != mod_map.end(); ++i) {mods[n++] = mod_elem_t(*i);}
		for (unsigned i = 0; i < brush_vect.size(); ++i) {brushes[i] = brush_vect[i];}
	}
};
static shim_hmap_manager_t shim_hmap;
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
static bool shim_using_hmap() {return shim_hmap.enabled();}                             // using_tiled_terrain_hmap_tex (src/tiled_mesh.cpp:273)
static bool shim_using_hmap_with_detail() {return (shim_using_hmap() && mesh_scale < 0.75);} // src/tiled_mesh.cpp:274
static float shim_get_xy_scale() { // src/tiled_mesh.cpp:447-451
	bool const add_detail(shim_using_hmap_with_detail());
	if (!add_detail && shim_using_hmap()) return 0.0;
	return (add_detail ? SHIM_HMAP_DETAIL_SCALE : 1.0);
}

// enable_tiled_mesh_ao (src/3DWorld.cpp:73,1778; scene_config/config.txt:81 turns it on): a config-file flag read by the tile code
static bool shim_enable_tiled_mesh_ao(0);
REF_API void ref_set_tiled_mesh_ao(int v) {shim_enable_tiled_mesh_ao = (v != 0);}
unsigned const SHIM_NUM_AO_DIRS = 8, SHIM_NUM_AO_STEPS = 8, SHIM_AO_RAY_LEN = SHIM_NUM_AO_STEPS*(SHIM_NUM_AO_STEPS+1)/2; // src/tiled_mesh.cpp:41-43

REF_API void ref_tile_create_zvals(int tx, int ty, unsigned iters_tt, float *zvals, ref_tile_stats_t *st) {
	unsigned const size(128), stride(size+1), zvsize(stride+1);
	int const x1(tx*size), y1(ty*size), x2(x1 + size), y2(y1 + size);
	int wx1(x2), wy1(y2), wx2(x1), wy2(y1); // start denormalized (src/tiled_mesh.cpp:308)
	mesh_xy_grid_cache_t height_gen;
	float mzmin(FAR_DISTANCE), mzmax(-FAR_DISTANCE);
	unsigned const block_size(zvsize/4), context_sz(stride + 2*SHIM_AO_RAY_LEN);
	float const wpz_max(ref_get_max_sea_level());
	if (shim_using_hmap()) { // src/tiled_mesh.cpp:499-503
		bool const add_detail(shim_using_hmap_with_detail());
		float const xy_scale(shim_get_xy_scale());
		if (xy_scale != 0.0) {height_gen.build_arrays((x1 - MESH_X_SIZE/2), (y1 - MESH_Y_SIZE/2), xy_scale*DX_VAL, xy_scale*DY_VAL, zvsize, zvsize, 0, 0, 0); height_gen.enable_glaciate();}
#pragma omp parallel for schedule(static,1)
		for (int y = 0; y < (int)zvsize; ++y) {
			for (unsigned x = 0; x < zvsize; ++x) {
				float &zval(zvals[y*zvsize + x]);
				zval = ref_get_clamped_height((x1 + x), (y1 + y));
				if (add_detail) {zval += SHIM_HMAP_DETAIL_MAG*height_gen.eval_index(x, y);}
			}
		}
		iters_tt = 0; // heightmap is eroded during load (:515)
	}
	else if (shim_enable_tiled_mesh_ao && mesh_gen_mode >= MGEN_SIMPLEX_GPU) { // AO + GPU noise: the zvals are clipped from the 201^2 AO context (src/tiled_mesh.cpp:478-488,505)
		height_gen.build_arrays(((x1 - (int)SHIM_AO_RAY_LEN) - MESH_X_SIZE/2), ((y1 - (int)SHIM_AO_RAY_LEN) - MESH_Y_SIZE/2), DX_VAL, DY_VAL, context_sz, context_sz, 0, 0, 0);
		height_gen.enable_glaciate();
#pragma omp parallel for schedule(static,1)
		for (int y = 0; y < (int)zvsize; ++y) {
			for (unsigned x = 0; x < zvsize; ++x) {zvals[y*zvsize + x] = height_gen.eval_index(x + SHIM_AO_RAY_LEN, y + SHIM_AO_RAY_LEN);}
		}
	}
	else {
		height_gen.build_arrays((x1 - MESH_X_SIZE/2), (y1 - MESH_Y_SIZE/2), DX_VAL, DY_VAL, zvsize, zvsize, 0, 0, 0); // setup_height_gen_async, xy_scale=1
		height_gen.enable_glaciate();
#pragma omp parallel for schedule(static,1)
		for (int y = 0; y < (int)zvsize; ++y) {
			for (unsigned x = 0; x < zvsize; ++x) {zvals[y*zvsize + x] = height_gen.eval_index(x, y);}
		}
	}
	apply_erosion(zvals, zvsize, zvsize, zmin, iters_tt);

	for (unsigned yy = 0; yy < 4; ++yy) {
		for (unsigned xx = 0; xx < 4; ++xx) {
			unsigned const x_end((xx+1)*block_size), y_end((yy+1)*block_size);
			float &szmin(st->sub_zmin[yy*4+xx]), &szmax(st->sub_zmax[yy*4+xx]);
			szmin = FAR_DISTANCE; szmax = -FAR_DISTANCE;
			for (unsigned y = yy*block_size; y <= y_end; ++y) {
				for (unsigned x = xx*block_size; x <= x_end; ++x) {
					float const z(zvals[y*zvsize + x]);
					szmin = min(szmin, z); szmax = max(szmax, z);
					if (z < wpz_max) {
						wx1 = min(wx1, x1+int(x)); wy1 = min(wy1, y1+int(y));
						wx2 = max(wx2, x1+int(x)); wy2 = max(wy2, y1+int(y));
					}
				}
			}
			mzmin = min(mzmin, szmin);
			mzmax = max(mzmax, szmax);
		}
	}
	st->mzmin = mzmin; st->mzmax = mzmax;
	st->radius = 0.5*sqrt((DX_VAL*DX_VAL + DY_VAL*DY_VAL)*size*size + (mzmax - mzmin)*(mzmax - mzmin));
	st->wx1 = wx1; st->wy1 = wy1; st->wx2 = wx2; st->wy2 = wy2;
}

// tile_t::calc_mesh_ao_lighting driver (src/tiled_mesh.cpp:586-661) for tile (tx,ty): zvals[130*130] (as create_zvals left them) -> ao[129*129]
REF_API void ref_tile_ao_lighting(int tx, int ty, float const *zvals, unsigned char *ao) {
	unsigned const size(128), stride(size+1), zvsize(stride+1), context_sz(stride + 2*SHIM_AO_RAY_LEN);
	int const x1(tx*size), y1(ty*size);
	int ao_dirs[SHIM_NUM_AO_DIRS][2];
	unsigned ix(0);
	for (int y = -1; y <= 1; ++y) {
		for (int x = -1; x <= 1; ++x) {
			if (x != 0 || y != 0) {ao_dirs[ix][0] = x; ao_dirs[ix][1] = y; ++ix;}
		}
	}
	bool const using_hmap(shim_using_hmap()), add_detail(shim_using_hmap_with_detail());
	bool const use_ao_zvals(!using_hmap && shim_enable_tiled_mesh_ao && mesh_gen_mode >= MGEN_SIMPLEX_GPU); // ao_zvals kept by create_zvals: the whole context, interior included
	vector<float> czv(context_sz*context_sz);
	mesh_xy_grid_cache_t height_gen;
	float const xy_scale(shim_get_xy_scale());
	if (xy_scale != 0.0) {
		height_gen.build_arrays(((x1 - (int)SHIM_AO_RAY_LEN) - MESH_X_SIZE/2), ((y1 - (int)SHIM_AO_RAY_LEN) - MESH_Y_SIZE/2), xy_scale*DX_VAL, xy_scale*DY_VAL, context_sz, context_sz, 0, 0, 0);
		height_gen.enable_glaciate();
	}
	float const dz(0.5*HALF_DXY);
#pragma omp parallel for schedule(static,1)
	for (int y = 0; y < (int)context_sz; ++y) {
		for (int x = 0; x < (int)context_sz; ++x) {
			int const xv(x - (int)SHIM_AO_RAY_LEN), yv(y - (int)SHIM_AO_RAY_LEN);
			float &zv(czv[y*context_sz + x]);
			if (!use_ao_zvals && xv >= 0 && yv >= 0 && xv < (int)zvsize && yv < (int)zvsize) {zv = zvals[yv*zvsize + xv];}
			else if (using_hmap) {
				zv = ref_get_clamped_height((x1 + xv), (y1 + yv));
				if (add_detail) {zv += SHIM_HMAP_DETAIL_MAG*height_gen.eval_index(x, y);}
			}
			else {zv = height_gen.eval_index(x, y);}
		}
	}
#pragma omp parallel for schedule(static,1)
	for (int y = 0; y < (int)stride; ++y) {
		for (int x = 0; x < (int)stride; ++x) {
			unsigned atten(0);
			for (unsigned d = 0; d < SHIM_NUM_AO_DIRS; ++d) {
				float z0(zvals[y*zvsize + x]);
				int stepx(ao_dirs[d][0]), stepy(ao_dirs[d][1]), vx(x), vy(y);
				for (unsigned s = 0; s < SHIM_NUM_AO_STEPS; ++s) {
					vx += stepx; vy += stepy;
					z0 += dz;
					stepx += ao_dirs[d][0]; stepy += ao_dirs[d][1]; // linear increase
					int const xv(vx + (int)SHIM_AO_RAY_LEN), yv(vy + (int)SHIM_AO_RAY_LEN);
					if (czv[yv*context_sz + xv] > z0) {atten += (SHIM_NUM_AO_STEPS - s); break;}
				}
			}
			float const ao_scale(1.0 - float(atten)/float(SHIM_NUM_AO_DIRS*SHIM_NUM_AO_STEPS));
			ao[y*stride + x] = (unsigned char)(255.0*ao_scale);
		}
	}
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
// ---- f3: landscape weights texture.  tile_t::create_texture / update_terrain_params live in tiled_mesh.cpp and get_tids / update_lttex_ix in Textures.cpp
// (GL-bound translation units that cannot be built here), so this is a driver in the reference's own types and macros (vector3d, CLIP_TO_01, ttex ids)
// around the reference functions that ARE compiled: build_arrays / eval_index (the noise field), eval_mesh_sin_terms (the biome parameters),
// get_water_z_height, init_terrain_mesh (lttex_dirt), sthresh.  It pins those inputs; the blend logic itself is a second restatement.
struct ref_landscape_t {float vegetation, temperature, biome_x_offset, mesh_scale_z; int water_is_lava, disable_water, enable_terrain_env; unsigned grass_density, num_rnd_grass_blocks;};
struct ref_grass_block_t {unsigned ix; float zmin, zmax;}; // tile_t::grass_block_t (src/tiled_mesh.h:186)
static ref_landscape_t shim_ls = {1.0, DEF_TEMPERATURE, 0.0, 1.0, 0, 0, 1, 0, 16};
extern float sthresh[2][2];
REF_API void ref_set_landscape(ref_landscape_t const *p) {
	shim_ls = *p; temperature = p->temperature; mesh_scale_z = p->mesh_scale_z;
	init_terrain_mesh(); // calls gen_tex_height_tables()
}
static void shim_update_lttex_ix(int &ix) { // src/Textures.cpp:1289-1292
	if ((shim_ls.water_is_lava || shim_ls.disable_water == 2) && lttex_dirt[ix].id == SNOW_TEX) {--ix;}
	if (shim_ls.vegetation == 0.0 && lttex_dirt[ix].id == GROUND_TEX) {++ix;}
}
static void shim_get_tids(float relh, int &k1, int &k2, float *t=nullptr) { // src/Textures.cpp:1294-1316
	float const TEXTURE_SMOOTH = 0.01;
	for (k1 = 0; k1 < NTEX_DIRT-1; ++k1) {if (relh < h_dirt[k1]) break;}
	if (k1 < NTEX_DIRT-1 && (h_dirt[k1] - relh) < TEXTURE_SMOOTH) {
		if (t) {*t = 1.0 - (h_dirt[k1] - relh)/TEXTURE_SMOOTH;}
		k2 = k1+1;
		shim_update_lttex_ix(k1);
		shim_update_lttex_ix(k2);
	}
	else {
		shim_update_lttex_ix(k1);
		k2 = k1;
	}
}
struct shim_terrain_params_t {float veg=1.0, grass=1.0, dirt=0.0;};
static void shim_update_terrain_params(int x1, int y1, int x2, int y2, shim_terrain_params_t params[2][2]) { // src/tiled_mesh.cpp:321-343
	float const dirt_mult(1.0), veg_mult(5.0);
	float const xv1(get_xval(x1)), xv2(xv1 + (x2-x1)*DX_VAL), yv1(get_yval(y1)), yv2(yv1 + (y2-y1)*DY_VAL);
	for (unsigned yp = 0; yp < 2; ++yp) {
		for (unsigned xp = 0; xp < 2; ++xp) {
			shim_terrain_params_t &param(params[yp][xp]);
			float const xv(mesh_scale*(xp ? xv2 : xv1) + shim_ls.biome_x_offset), yv(mesh_scale*(yp ? yv2 : yv1));
			float const veg_val(eval_mesh_sin_terms(veg_mult*xv, veg_mult*yv));
			param.veg   = CLIP_TO_01(5.000f*(veg_val + 1.5f));
			param.grass = CLIP_TO_01(100.0f*(veg_val + 3.0f));
			param.dirt  = CLIP_TO_01(5.0f*(eval_mesh_sin_terms(dirt_mult*xv, dirt_mult*yv) + 1.0f));
		}
	}
}
REF_API void ref_tile_terrain_params(int tx, int ty, float *out) {
	shim_terrain_params_t params[2][2];
	if (shim_ls.enable_terrain_env) {shim_update_terrain_params(tx*128, ty*128, tx*128 + 128, ty*128 + 128, params);}
	for (unsigned i = 0; i < 4; ++i) {shim_terrain_params_t const &p(params[i >> 1][i & 1]); out[3*i] = p.veg; out[3*i+1] = p.grass; out[3*i+2] = p.dirt;}
}
#define SHIM_BILINEAR_INTERP(arr, var, x, y) (y*(x*arr[1][1].var + (1.0f-x)*arr[1][0].var) + (1.0f-y)*(x*arr[0][1].var + (1.0f-x)*arr[0][0].var)) // src/tiled_mesh.cpp:189
REF_API void ref_tile_create_weights(int tx, int ty, float const *zvals, unsigned char *mesh_weight_data, ref_grass_block_t *grass_blocks, int *has_any_grass_out) {
	unsigned const size(128), stride(size+1), zvsize(stride+1), tsize(stride), grass_block_sz(4), grass_block_dim(1+(size-1)/grass_block_sz);
	int const x1(tx*size), y1(ty*size);
	int sand_tex_ix(-1), dirt_tex_ix(-1), grass_tex_ix(-1), rock_tex_ix(-1), snow_tex_ix(-1);
	for (unsigned i = 0; i < NTEX_DIRT; ++i) { // get_texture_ixs (src/tiled_mesh.cpp:1049-1062)
		switch (lttex_dirt[i].id) {
		case SAND_TEX:   sand_tex_ix  = i; break;
		case DIRT_TEX:   dirt_tex_ix  = i; break;
		case GROUND_TEX: grass_tex_ix = i; break;
		case ROCK_TEX:   rock_tex_ix  = i; break;
		case SNOW_TEX:   snow_tex_ix  = i; break;
		}
	}
	shim_terrain_params_t params[2][2];
	if (shim_ls.enable_terrain_env) {shim_update_terrain_params(x1, y1, x1 + size, y1 + size, params);}
	bool has_any_grass(0);
	bool const gen_grass_map(shim_ls.grass_density > 0 && shim_ls.vegetation > 0.0);
	if (grass_blocks) {for (unsigned i = 0; i < grass_block_dim*grass_block_dim; ++i) {grass_blocks[i] = ref_grass_block_t{0, 0.0, 0.0};}}
	float const vegetation(shim_ls.vegetation);
	float const xy_mult(1.0/float(size)), water_level(get_water_z_height());
	float const MESH_NOISE_SCALE = 0.003;
	float const MESH_NOISE_FREQ  = 80.0;
	float const dz_inv(1.0f/(zmax - zmin));
	float const noise_scale(((mesh_gen_shape == 2) ? 2.0 : 1.0)*MESH_NOISE_SCALE*mesh_scale_z);
	float const steep_mult_grass(1.0f/(sthresh[0][1] - sthresh[0][0]));
	float const steep_mult_snow (1.0f/(sthresh[1][1] - sthresh[1][0]));
	float const steep_mult_rock (1.0f/(0.8f*sthresh[0][0] - 0.5f*sthresh[0][0]));
	float const vnz_scale((mesh_gen_mode == MGEN_DWARP_GPU) ? SQRT2 : 1.0);
	int k1, k2, k3, k4;
	mesh_xy_grid_cache_t height_gen;
	height_gen.build_arrays((x1 - MESH_X_SIZE/2), (y1 - MESH_Y_SIZE/2), MESH_NOISE_FREQ*DX_VAL, MESH_NOISE_FREQ*DY_VAL, tsize, tsize, 0, 1); // force_sine_mode=1
	vector<float> rand_vals(tsize*tsize);
	for (unsigned y = 0; y < tsize; ++y) {
		for (unsigned x = 0; x < tsize; ++x) {rand_vals[y*tsize + x] = noise_scale*height_gen.eval_index(x, y, 50);}
	}
	for (unsigned y = 0; y < tsize; ++y) {
		float const yv(float(y)*xy_mult);
		for (unsigned x = 0; x < tsize; ++x) {
			unsigned const ix_val(y*tsize + x), off(4*ix_val), ix(y*zvsize + x);
			float weights[NTEX_DIRT] = {};
			float const mh00(zvals[ix]), mh01(zvals[ix+1]), mh10(zvals[ix+zvsize]), mh11(zvals[ix+zvsize+1]);
			float const mhmin(min(min(mh00, mh01), min(mh10, mh11))), mhmax(max(max(mh00, mh01), max(mh10, mh11)));
			float const rand_offset(rand_vals[y*tsize + x]);
			float const relh1(relh_adj_tex + (mhmin - zmin)*dz_inv + rand_offset), relh2(relh_adj_tex + (mhmax - zmin)*dz_inv + rand_offset);
			shim_get_tids(relh1, k1, k2);
			shim_get_tids(relh2, k3, k4);
			bool const same_tid(k1 == k4);
			float t(0.0);
			k2 = k4;
			if (!same_tid) {
				float const relh(relh_adj_tex + (mh00 - zmin)*dz_inv);
				shim_get_tids(relh, k1, k2, &t);
			}
			float weight_scale(1.0);
			bool const grass(lttex_dirt[k1].id == GROUND_TEX || lttex_dirt[k2].id == GROUND_TEX), snow(lttex_dirt[k2].id == SNOW_TEX);
			has_any_grass |= grass;
			if (grass || snow) {
				float const *const sti(sthresh[snow]);
				vector3d const normal(DY_VAL*(zvals[ix] - zvals[ix + 1]), DX_VAL*(zvals[ix] - zvals[ix + zvsize]), dxdy); // get_norm_not_normalized (src/tiled_mesh.h:281)
				float vnz(vnz_scale*normal.z/normal.mag());
				if (grass && vnz > sti[1]) {vnz = CLIP_TO_01(1.0f + 20.0f*rand_offset);}
				if (vnz < sti[1]) {
					if (grass) {
						float rock_weight((lttex_dirt[k1].id == GROUND_TEX || lttex_dirt[k2].id == ROCK_TEX) ? t : 0.0);
						float const steepness(1.0 - CLIP_TO_01((vnz - 0.5f*sti[0])*steep_mult_rock));
						rock_weight  = rock_weight*(1.0 - steepness) + steepness;
						weight_scale = CLIP_TO_01((vnz - sti[0])*steep_mult_grass);
						weights[rock_tex_ix] += (1.0 - weight_scale)*rock_weight;
						weights[dirt_tex_ix] += (1.0 - weight_scale)*(1.0 - rock_weight);
					}
					else {
						weight_scale = CLIP_TO_01(2.0f*(vnz - sti[0])*steep_mult_snow);
						weights[rock_tex_ix] += 1.0 - weight_scale;
					}
				}
			}
			weights[k2] += weight_scale*t;
			weights[k1] += weight_scale*(1.0 - t);
			float const xv(float(x)*xy_mult);
			if (vegetation > 0.0) {
				float const dirt_scale(SHIM_BILINEAR_INTERP(params, dirt, xv, yv));
				if (dirt_scale < 1.0) {
					weights[sand_tex_ix] += (1.0 - dirt_scale)*weights[dirt_tex_ix];
					weights[dirt_tex_ix] *= dirt_scale;
				}
			}
			if (grass) {
				float grass_scale((mhmin < water_level) ? 0.0f : SHIM_BILINEAR_INTERP(params, grass, xv, yv));
				if (grass_scale < 1.0) {
					float const gscale(CLIP_TO_01(2.5f*(grass_scale - 0.5f) + 0.5f));
					weights[sand_tex_ix ] += (1.0 - gscale)*weights[grass_tex_ix];
					weights[grass_tex_ix] *= gscale;
				}
				if (grass_scale > 0.0 && grass_blocks && gen_grass_map && x < size && y < size) { // add_grass_block_at (src/tiled_mesh.cpp:1354-1371)
					ref_grass_block_t &gb(grass_blocks[(y/grass_block_sz)*grass_block_dim + x/grass_block_sz]);
					if (gb.ix == 0) {
						gb.ix   = (((x1 + x) + 1567*(y1 + y)) % shim_ls.num_rnd_grass_blocks) + 1;
						gb.zmin = mhmin;
						gb.zmax = mhmax;
					}
					else {
						min_eq(gb.zmin, mhmin);
						max_eq(gb.zmax, mhmax);
					}
				}
			}
			for (unsigned i = 0; i < NTEX_DIRT-1; ++i) {
				mesh_weight_data[off+i] = ((weights[i] <= 0.01) ? 0 : ((weights[i] >= 0.99) ? 255 : (unsigned char)(255.0*weights[i])));
			}
		}
	}
	(void)snow_tex_ix;
	if (has_any_grass_out) {*has_any_grass_out = has_any_grass;}
}

REF_API float ref_tile_normals(float const *zvals, unsigned char *rgba /*129*129*4*/) {
	unsigned const stride(129), zvsize(130);
	float min_normal_z(1.0);
	memset(rgba, 0, 4*stride*stride);
	for (unsigned y = 0; y < stride; ++y) {
		for (unsigned x = 0; x < stride; ++x) {
			unsigned const ix(y*stride + x), ix2(y*zvsize + x), ix_off(4*ix);
			vector3d const norm(vector3d(DY_VAL*(zvals[ix2] - zvals[ix2 + 1]), DX_VAL*(zvals[ix2] - zvals[ix2 + zvsize]), dxdy).get_norm());
			min_normal_z = min(min_normal_z, norm.z);
			UNROLL_3X(rgba[ix_off+i_] = (unsigned char)(127.0*(norm[i_] + 1.0));)
		}
	}
	return min_normal_z;
}

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
REF_API void ref_export_heightmap(float xstart, float ystart, int width, int height, unsigned char *pixels, float *min_z_dz) {
	texture_t texture(0, 6, width, height, 0, 2, 0, "heightmap.png"); // two bytes per pixel grayscale
	texture.set_16_bit_grayscale();
	texture.alloc();
	vector<float> heights(texture.num_pixels());
	mesh_xy_grid_cache_t height_gen;
	float const xy_scale(shim_get_xy_scale());
	if (xy_scale != 0.0) {
		height_gen.build_arrays(xstart/DX_VAL, ystart/DY_VAL, xy_scale*DX_VAL, xy_scale*DY_VAL, width, height, 1); // cache_values=1
		height_gen.enable_glaciate();
	}
	float const xscale(DX_VAL), yscale(DY_VAL);
#pragma omp parallel for schedule(static,1)
	for (int i = 0; i < height; ++i) {
		int const off(width*(height - i - 1)); // invert yval
		for (int j = 0; j < width; ++j) {
			float zval;
			if (shim_using_hmap()) {
				zval = shim_hmap.interpolate_height((xstart + X_SCENE_SIZE + j*xscale)*DX_VAL_INV, (ystart + Y_SCENE_SIZE + i*yscale)*DY_VAL_INV);
				if (shim_using_hmap_with_detail()) {zval += HMAP_DETAIL_MAG*height_gen.eval_index(j, i);}
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
