/* oracle/terra_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the 3DWorld procedural-terrain hot path, used as the CPU
 * checker for the HIP library.  Nothing under oracle/ is linked into, imported by, or called from the product
 * (3dworld_amd/, libterra_hip.so); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY PIN: this restatement is checked bit-for-bit against the reference's own translation units compiled
 * in place (oracle/_ref/liboracle_ref.so, recipe oracle/Makefile) by tests/test_oracle_vs_ref.py whenever
 * /root/reference is present, and against tests/golden/ fixtures generated from that build
 * (tests/golden/make_golden.py) everywhere else.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef TERRA_ORACLE_H
#define TERRA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_F_TABLE_SIZE 90 /* NUM_FREQ_COMP(9) * N_RAND_SIN2(10), src/mesh_gen.cpp:14,16,30 */

enum {ORC_MGEN_SINE = 0, ORC_MGEN_SIMPLEX, ORC_MGEN_PERLIN, ORC_MGEN_SIMPLEX_GPU, ORC_MGEN_DWARP_GPU}; /* src/3DWorld.h:1399 */

/* what a config file sets (src/3DWorld.cpp:1763-2110); same layout as ref_config_t in oracle/ref_shim.cpp */
typedef struct {
	int mesh_x, mesh_y;
	float scene_x, scene_y, scene_z;
	float mesh_height, mesh_scale;
	int mesh_seed, mesh_freq_filter, mesh_gen_mode, mesh_gen_shape, glaciate;
	float custom_glaciate_exp;
	float hmap[14]; /* hmap_params_t, src/mesh.h:84-88: plat_bot plat_h plat_s plat_max crat_h crat_s crack_lo crack_hi crack_d sine_mag sine_freq sine_bias volcano_width volcano_height */
	float erode_amount, water_h_off, water_h_off_rel, relh_adj_tex, ocean_wave_height;
	float start_mag, start_freq, mag_mult, freq_mult;
} orc_config_t;

/* derived state; same layout as ref_state_t in oracle/ref_shim.cpp */
typedef struct {
	float sinTable[ORC_F_TABLE_SIZE][5];
	int start_eval_sin;
	float MESH_HEIGHT, DX_VAL, DY_VAL, DX_VAL_INV, DY_VAL_INV, HALF_DXY, dxdy, XY_SCENE_SIZE;
	float mesh_scale, mesh_scale_z_inv, mesh_height_scale;
	float zmax_est, zmin, zmax, water_plane_z, glaciate_exp, clip_hd1, relh_adj_tex;
	float rx, ry;
} orc_state_t;

typedef struct {float sub_zmin[16], sub_zmax[16], mzmin, mzmax, radius; int wx1, wy1, wx2, wy2;} orc_tile_stats_t;

/* per-droplet trace statistics (instrumentation only) */
typedef struct {uint64_t steps, erode_steps, deposit_steps, ocean_stops, pit_stops, nan_droplets; uint32_t max_steps;} orc_erosion_stats_t;

void  orc_init(orc_config_t const *c);
void  orc_get_state(orc_state_t *s);
void  orc_set_zmax_est(float v);
void  orc_set_water_plane_z(float v);
void  orc_set_mode(int mode, int shape);
void  orc_set_start_eval_sin(int v);
void  orc_set_erode_amount(float v);
void  orc_get_ground_mesh(float *out);
float orc_sin_table(int i);
int   orc_num_threads(void);
void  orc_set_num_threads(int n);

/* 1: gen_grid / tiles evaluate the product's TOLERANCE mode (terra_oracle.c: g_fused) instead of the reference's arithmetic; 0 (default): the reference's */
void  orc_set_fused(int on);
void  orc_gen_grid(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, int glaciate, int cache_values, int min_start_sin, float *out);
uint64_t orc_apply_erosion_trace(float *hmap, int xsize, int ysize, float min_zval, unsigned iters, uint32_t *cells, uint64_t cap, uint64_t *offsets);
void  orc_gen_grid_ex(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, int glaciate, int cache_values, int force_sine_mode, int min_start_sin, int use_cache, float *out);
void  orc_gen_grid_rect(float x0, float y0, float dx, float dy, unsigned nx, unsigned ny, int glaciate, int min_start_sin, unsigned rx0, unsigned ry0, unsigned rw, unsigned rh, float *out);
void  orc_apply_erosion(float *hmap, int xsize, int ysize, float min_zval, unsigned iters);
void  orc_apply_erosion_stats(float *hmap, int xsize, int ysize, float min_zval, unsigned iters, orc_erosion_stats_t *st, uint32_t *steps_per_droplet);
float orc_get_noise_zval(float x, float y, int mode, int shape);
float orc_gen_noise(float x, float y, int mode, int shape);
float orc_eval_mesh_sin_terms(float x, float y);
/* eval_mesh_sin_terms_scaled (exact = 0) / get_exact_zval (exact = 1) for n points xy[2 i], xy[2 i + 1] (src/mesh_gen.cpp:807-847) */
void  orc_eval_points(float const *xy, unsigned n, int exact, float xy_scale, int no_xyoff, int xoff2, int yoff2, float *out);
float orc_glm_simplex2(float x, float y);
float orc_glm_perlin2(float x, float y);
float orc_glm_simplex3(float x, float y, float z);
float orc_glm_perlin3(float x, float y, float z);
int   orc_get_bare_ls_tid_is_rock(float z);
float orc_get_max_sea_level(void);
void  orc_rand_ints(long s1, long s2, int n, int *out);
void  orc_rand_floats(long s1, long s2, int n, float *out);
void  orc_rand_uniforms(long s1, long s2, float a, float b, int n, float *out);
void  orc_tile_create_zvals(int tx, int ty, unsigned iters_tt, float *zvals, orc_tile_stats_t *st);
void  orc_set_tiled_mesh_ao(int v);
void  orc_hmap_set(unsigned char const *pixels, int width, int height, int ncolors); /* NULL: back to procedural tiles */
void  orc_set_mesh_height_scales_for_zval_range(float min_z, float dz);
float orc_get_clamped_height(int x, int y);
void  orc_hmap_get(unsigned char *out); /* the current image (after brushes / mods) */
float orc_hmap_interpolate_height(float x, float y);
float orc_hmap_get_nearest_height(float x, float y);
/* rest of row f4: tex_mod_map_manager_t::hmap_brush_t / mod_elem_t with the reference's layouts (src/heightmap.h:59-81), the .mod file and the exporter */
typedef struct orc_hmap_brush_t {int x, y; unsigned radius; int delta; short shape;} orc_hmap_brush_t;
typedef struct orc_hmap_mod_t {unsigned short x, y; int delta;} orc_hmap_mod_t;
void  orc_hmap_apply_brush(orc_hmap_brush_t const *b, int step_sz, unsigned num_steps);
void  orc_hmap_apply_mods(orc_hmap_mod_t const *mods, unsigned n);
int   orc_hmap_write_mod(char const *fn, orc_hmap_mod_t const *mods, unsigned n, orc_hmap_brush_t const *brushes, unsigned nb);
int   orc_hmap_read_mod(char const *fn, orc_hmap_mod_t *mods, unsigned *n, orc_hmap_brush_t *brushes, unsigned *nb);
int   orc_hmap_read_and_apply_mod(char const *fn);
void  orc_heightmap_proc_gen(int width, int height, unsigned iters, unsigned char *pixels, float *file_scale_tz);
/* rest of row a12: the loaded-heightmap path (src/heightmap.cpp:117-128,191-215; config `mh_filename <png> <scale> <tz>`, src/3DWorld.cpp:2205) */
void  orc_set_mesh_file_scale(float mesh_file_scale, float mesh_file_tz);
int   orc_read_mesh(const char *filename, float zmm, float *zbottom_ztop); /* read_mesh, src/mesh_gen.cpp:895-933 */
int   orc_write_mesh(const char *filename);                                /* write_mesh, :936-965 */
void  orc_set_ground_mesh(const float *in);
void  orc_heightmap_to_floats(unsigned char const *pixels, int width, int height, int ncolors, float *vals);
unsigned orc_heightmap_from_floats(float const *vals, int width, int height, int ncolors, unsigned char *pixels);   /* -> values outside [0, 256) */
unsigned orc_heightmap_postprocess(unsigned char *pixels, int width, int height, int ncolors, unsigned iters_tt);    /* in place; -> values outside [0, 256) */
void  orc_export_heightmap(float xstart, float ystart, int width, int height, unsigned char *pixels, float *min_z_dz);
void  orc_tile_ao_lighting(int tx, int ty, float const *zvals, unsigned char *ao);
void  orc_calc_mesh_shadows(float lx, float ly, float lz, float const *mh, unsigned char *smask, int xsize, int ysize, float const *sh_in_x, float const *sh_in_y, float *sh_out_x, float *sh_out_y);
void  orc_tiles_mesh_shadows(int const *tile_xy, unsigned n, float const *zvals, float lx, float ly, float lz, unsigned char *smask);
/* f3: the globals tile_t::create_texture / update_terrain_params read beyond orc_config_t (defaults = the reference's) */
typedef struct orc_landscape_t {
	float vegetation, temperature, biome_x_offset, mesh_scale_z; /* 1, DEF_TEMPERATURE = 20, 0, 1 */
	int32_t water_is_lava, disable_water /* DISABLE_WATER */, enable_terrain_env /* 1 */;
	uint32_t grass_density /* 0: no grass blocks */, num_rnd_grass_blocks /* 16 */;
} orc_landscape_t;
typedef struct orc_grass_block_t {uint32_t ix; float zmin, zmax;} orc_grass_block_t; /* tile_t::grass_block_t, src/tiled_mesh.h:186 */
void  orc_set_landscape(orc_landscape_t const *p);
void  orc_tile_terrain_params(int tx, int ty, float *out12);
void  orc_tile_create_weights(int tx, int ty, float const *zvals, unsigned char *weights_rgba, orc_grass_block_t *blocks, int *has_any_grass);
float orc_tile_normals(float const *zvals, unsigned char *rgba);
void  orc_quantize16(float const *vals, size_t n, unsigned char *out, float *min_z_out, float *dz_out);
void  orc_voxel_fill(float *out, unsigned nx, unsigned ny, unsigned nz, float const lo_pos[3], float const vsz[3], float const offset[3],
	float mag, float freq, int rseed1, int rseed2, int gen_mode, float zscale, int normalize_to_1);
void  orc_voxel_rdata(int rseed1, int rseed2, float mag, float freq, float *rdata);

#ifdef __cplusplus
}
#endif
#endif
