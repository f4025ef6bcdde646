/* terra.h -- C ABI of libterra_hip.so: MI355X (gfx950) procedural-terrain hot path of 3DWorld.
 *
 * Drop-in boundary.  3DWorld has no plugin/FFI layer; the seam is the three C++ call surfaces
 * below plus the voxel fill.  Every entry point names the reference interface it replaces (paths relative to the
 * 3DWorld tree).  All pointers are plain host or device pointers, all sizes plain integers; nothing here depends on
 * torch, OpenGL or the engine's headers.  The engine keeps its process globals; they cross the boundary explicitly
 * as terra_config (what the config file sets) or terra_state (already-derived globals).
 *
 *   reference interface                                                     replaced by
 *   ----------------------------------------------------------------------  -------------------------------------------
 *   create_sin_table / gen_rand_sine_table_entries / compute_scale /        terra_init_scene
 *     estimate_zminmax / set_zvals / init_terrain_mesh / gen_tex_height_tables
 *     (src/mesh_gen.cpp:72-81,213-254,407-431,447-512,544-548; src/Textures.cpp:1757-1761)
 *   mesh_xy_grid_cache_t::build_arrays / enable_glaciate / eval_index       terra_gen_* handle, terra_gen_grid[_dev]
 *     (src/mesh.h:22-45, src/mesh_gen.cpp:588-650,754-792)
 *   apply_erosion(float*,int,int,float,unsigned)                            terra_apply_erosion[_dev]
 *     (src/function_registry.h:354, src/erosion.cpp:14-164)
 *   tile_t::create_zvals + get_norm/upload_normal_texture CPU part          terra_tiles_create_zvals[_dev]
 *     (src/tiled_mesh.h:277,281-284; src/tiled_mesh.cpp:467-546,865-880)
 *   heightmap_t::proc_gen / run_erosion / from_floats                       terra_heightmap_proc_gen[_dev], terra_quantize16_dev
 *     (src/heightmap.cpp:130-215, src/Textures.cpp:1889-1893)
 *   heightmap_t::to_floats / postprocess_height (loaded heightmaps)         terra_heightmap_to_floats_dev, terra_heightmap_from_floats_dev,
 *     (src/heightmap.cpp:117-128,191-203,351)                                 terra_heightmap_postprocess_dev, terra_set_mesh_file_scale
 *   voxel_manager::create_procedural                                        terra_voxel_fill[_dev]
 *     (src/voxels.cpp:278-346, src/upsurface.cpp:16-70)
 *
 * Error behaviour: the reference asserts; this library returns TERRA_OK (0) or a negative terra_status and keeps a
 * thread-local message (terra_last_error).  There is NO CPU fall-back: without a usable HIP device terra_create fails.
 *
 * Threading: one terra_ctx per host thread / GPU (one process per GPU in multi-GPU runs).  All *_dev work is enqueued
 * on the context's HIP stream (its own, or the caller's via terra_set_stream) and is asynchronous unless noted.
 * The terra_gen_* handles are the one exception (the reference calls eval_index from its OpenMP workers): handles of one context may be
 * used from several threads at once, terra_gen_eval_index on a collected grid takes no lock at all; what may NOT overlap is (a) any other
 * terra_* call on the same context with a terra_gen_* call that launches work (build_arrays, enable_glaciate, collect, an eval_index that
 * needs another first sine term), and (b) terra_gen_build_arrays with terra_gen_eval_index on the SAME handle (build_arrays is main-thread
 * only in the reference too, src/mesh.h:40).
 */
#ifndef TERRA_H
#define TERRA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TERRA_F_TABLE_SIZE 90 /* F_TABLE_SIZE, src/mesh_gen.cpp:30 */

typedef enum {TERRA_OK = 0, TERRA_ERR_ARG = -1, TERRA_ERR_HIP = -2, TERRA_ERR_STATE = -3, TERRA_ERR_NODEVICE = -4, TERRA_ERR_LIMIT = -5} terra_status;

/* mesh_gen_mode values (src/3DWorld.h:1399) */
enum {TERRA_MGEN_SINE = 0, TERRA_MGEN_SIMPLEX = 1, TERRA_MGEN_PERLIN = 2, TERRA_MGEN_SIMPLEX_GPU = 3, TERRA_MGEN_DWARP_GPU = 4};

/* What the engine's config file sets for this path (keyword -> global binding in src/3DWorld.cpp:1763-2110). */
typedef struct terra_config {
	int32_t mesh_x, mesh_y;                  /* mesh_size            -> MESH_X_SIZE, MESH_Y_SIZE */
	float scene_x, scene_y, scene_z;         /* scene_size           -> X/Y/Z_SCENE_SIZE */
	float mesh_height, mesh_scale;           /* mesh_height (-> mesh_height_scale), mesh_scale */
	int32_t mesh_seed, mesh_freq_filter, mesh_gen_mode, mesh_gen_shape, glaciate; /* mesh_seed, mesh_freq_filter, mesh_gen_mode, mesh_gen_shape, GLACIATE */
	float custom_glaciate_exp;               /* custom_glaciate_exp (0 = cubic) */
	float hmap[14];                          /* hmap_params_t in declaration order, src/mesh.h:84-88 */
	float erode_amount, water_h_off, water_h_off_rel, relh_adj_tex, ocean_wave_height;
	float start_mag, start_freq, mag_mult, freq_mult; /* mesh_start_mag/freq, mesh_mag/freq_mult */
} terra_config;

/* The derived globals of the reference; terra_get_state exports them, terra_set_state injects an engine's own values. */
typedef struct terra_state {
	float sinTable[TERRA_F_TABLE_SIZE][5];   /* {mag, y-phase, x-phase, y-freq, x-freq}, src/mesh_gen.cpp:247-251 */
	int32_t start_eval_sin;
	float MESH_HEIGHT, DX_VAL, DY_VAL, DX_VAL_INV, DY_VAL_INV, HALF_DXY, dxdy, XY_SCENE_SIZE;
	float mesh_scale, mesh_scale_z_inv, mesh_height_scale;
	float zmax_est, zmin, zmax, water_plane_z, glaciate_exp, clip_hd1, relh_adj_tex;
	float rx, ry;                            /* gen_rx_ry(), src/mesh_gen.cpp:581-586 */
} terra_state;

/* tile_t outputs of create_zvals (src/tiled_mesh.cpp:517-541): 4x4 sub-block z range, tile z range, radius, water bbox (ints, bit-exact) */
typedef struct terra_tile_stats {
	float sub_zmin[16], sub_zmax[16], mzmin, mzmax, radius;
	int32_t wx1, wy1, wx2, wy2;
} terra_tile_stats;

/* counters of the last terra_apply_erosion*_dev call (diagnostics / bench) */
typedef struct terra_erosion_report {
	uint32_t droplets, windows /* ring generations = ceil(droplets / slots) */, rounds, traces, serial_fallbacks, nan_droplets;
	uint64_t steps;          /* droplet steps of the final (committed) traces */
	uint64_t traced_steps;   /* droplet steps actually simulated, re-traces included */
	uint64_t retraces_same;  /* re-traces that reproduced the published version bit for bit (the conflict that caused them was one of blocks, not of cells read) */
	uint64_t checkpoint_resumes;     /* re-traces that started from a checkpoint of the droplet's previous trace instead of from its spawn */
	uint64_t checkpoint_steps_saved; /* steps those re-traces did not have to repeat */
	/* where the time goes (speculative scheduler only).  Everything from here down to the end is collected only with TERRA_ERO_DIAG=1 in the environment
	 * (0 otherwise): the counters live in one cache line that every trace of the chip would have to update */
	uint64_t window_shifts;  /* times the 32 x 32 LDS window was moved */
	uint64_t critical_steps; /* sum over the rounds of the most steps any one trace made in the round: the scheduler's serial chain, in droplet steps */
	uint64_t critical_shifts;/* the same for window moves */
	/* device time in 10 ns ticks, summed over all traces: a trace's whole wave body / before its first step / inside window moves / after its last step;
	 * clk_critical: the longest wave body of each round, summed over the rounds */
	uint64_t clk_wave, clk_init, clk_shift, clk_tail, clk_critical;
	uint64_t clk_shift_flush, clk_shift_prep, clk_shift_load; /* parts of clk_shift: write-back of the cells that leave / block flags (candidate versions per block); clk_shift_load is 0 since the grid loads are in flight during the look-ups and are not timed apart: the rest of clk_shift is loads + look-ups + filling the window */
	uint64_t crit_clk_flush, crit_clk_load, crit_clk_prep; /* the parts of crit_clk_shift */
	uint64_t crit_clk_shift, crit_clk_edge, crit_steps_own; /* of each round's longest wave body: ticks in window moves, ticks before + after its steps, its steps (multiple of 4) */
	/* the sparse scheduler (few droplets on a big map: lean traces on the grid itself, one round per conflicting droplet): droplets it committed (0: not tried; `droplets`: the
	 * whole run -- anything less: the multi-version scheduler did the rest) and the re-traces that took */
	uint64_t sparse_droplets, sparse_retraces;
	uint64_t sparse_probe_only; /* droplets of such a run that ended at their first step without writing (ocean): settled by one thread each, no trace wave */
} terra_erosion_report;

/* The globals tile_t::create_texture and tile_t::update_terrain_params read beyond terra_config; the defaults are the reference's. */
typedef struct terra_landscape {
	float vegetation;            /* config "vegetation" (src/3DWorld.cpp:109): 0 turns ground/grass into rock and disables the sand conversions */
	float temperature;           /* DEF_TEMPERATURE = 20 (src/3DWorld.h:87): above 40 the snow line rises (src/mesh_gen.cpp:423-426) */
	float biome_x_offset;        /* config "biome_x_offset" */
	float mesh_scale_z;          /* src/mesh_gen.cpp:37,873: 1 until the terrain zoom changes it */
	int32_t water_is_lava;       /* config "water_is_lava": snow -> rock */
	int32_t disable_water;       /* DISABLE_WATER: 2 also turns snow into rock (src/Textures.cpp:1290) */
	int32_t enable_terrain_env;  /* ENABLE_TERRAIN_ENV = 1 (src/tiled_mesh.h:21): biome parameters from eval_mesh_sin_terms at the tile corners; 0 = {veg 1, grass 1, dirt 0} */
	uint32_t grass_density;      /* config "grass_density": 0 = gen_grass_map() false, no grass blocks (src/tiled_mesh.cpp:126) */
	uint32_t num_rnd_grass_blocks; /* 16 (src/grass.cpp:14) */
} terra_landscape;
/* tile_t::grass_block_t (src/tiled_mesh.h:186): ix 0 = no grass in this 4x4-texel block, else 1 + index of the random grass block; z range of its texels */
typedef struct terra_grass_block {uint32_t ix; float zmin, zmax;} terra_grass_block;

typedef struct terra_ctx terra_ctx;
typedef struct terra_gen terra_gen;

/* flags of terra_gen_grid* / terra_gen_build_arrays (bool arguments of build_arrays, src/mesh.h:40) */
#define TERRA_GEN_GLACIATE     1u  /* enable_glaciate() after build_arrays() */
#define TERRA_GEN_FORCE_SINE   2u  /* force_sine_mode */
#define TERRA_GEN_NO_WAIT      4u  /* no_wait: return 0 right after launch */
#define TERRA_GEN_CACHE_VALUES 8u  /* cache_values: every cell is always evaluated on the device; the flag only decides what eval_index(.., use_cache=1) returns */
/* TOLERANCE mode (opt-in; everything else in this library is bit-identical to the reference's CPU path).  The reference binary has no fused multiply-add, so the exact
 * kernels pay a multiply AND an add per term of eval_index's sum (src/mesh_gen.cpp:777-779) and can never pass half of the chip's fp32 rate.  With this flag every a*b + c of
 * the sum and of eval_index's tail (glaciate's last step, the island term; src/mesh_gen.cpp:380-385,781-790) rounds ONCE: the value is the reference's expression tree
 * evaluated with fmaf, exactly (pinned against that restatement in the tests' checker: orc_set_fused), and within 1e-5 * zmax_est of the reference (BASELINE's bar; measured
 * max 3e-7 * zmax_est on the 16384^2 grid).  The sine tables are the exact mode's (SINF indices pinned).  The sum runs on the f32 matrix pipe (csrc/terra_fused.hpp).  It is a
 * permission, not a command: a configuration without a fused kernel (fBm modes, shapes / plateau / crater / volcano set-ups whose cells can leave the short tail) is
 * evaluated by the exact kernel.  Do not feed fused heights to apply_erosion when the droplet paths must be the reference's: erosion amplifies a last-bit difference.
 * The calls without a flags argument (tiles, voxels) take it from terra_set_option(ctx, "gen.fused", "1"). */
#define TERRA_GEN_FUSED        16u
/* The cheapest form within the same tolerance (implies TERRA_GEN_FUSED where it has no kernel of its own): the sine sum's tables are split into scaled half-precision pairs and
 * the contraction runs on the half-precision matrix pipe, 16 times the f32 pipe's rate (csrc/terra_fused.hpp: k_sine_grid_h3) -- the kernel is then bound by writing its result.
 * Within 1e-5 * zmax_est of the reference like TERRA_GEN_FUSED (measured ~1e-6), but not the value of any restatement: tested against the tolerance only.  Option "gen.fused" = "2". */
#define TERRA_GEN_FAST         32u
/* flags of terra_apply_erosion*_dev */
#define TERRA_ERODE_SERIAL        1u /* walk droplets one by one on one lane (reference order, no speculation): debugging / tiny grids */
#define TERRA_ERODE_SERIAL_WAVE   4u /* droplets one after another, each simulated by a whole wave through the LDS window (no speculation) */
#define TERRA_ERODE_MINZ_IS_MIN   2u /* caller guarantees min_zval <= every grid value (heightmap_t::run_erosion passes min(vals)): clamp only written cells */

const char *terra_last_error(void);
int  terra_device_count(void);

/* ---- context */
int  terra_create(terra_ctx **out, int device_index);
void terra_destroy(terra_ctx *ctx);
int  terra_set_stream(terra_ctx *ctx, void *hip_stream);   /* use the caller's hipStream_t (e.g. torch's current stream); NULL = own stream */
int  terra_synchronize(terra_ctx *ctx);
/* ---- options: every behaviour switch of the library (nothing in it reads the process environment).  Values are strings; an unknown key or a value outside the key's
 * range is TERRA_ERR_ARG and changes nothing.  The call drains the context's stream first.  Only "gen.fused" changes a result.
 *   "gen.fused"            "0" | "1" | "2"   every generator call of this context behaves as if TERRA_GEN_FUSED ("1") / TERRA_GEN_FAST ("2") were given (tile batches and voxel fields have no flags argument)
 *   "kernels.simple"       "0" | "1"      one-thread-per-cell cross-check kernels instead of the tiled ones (tests)
 *   "graphs"               "0" | "1"      replay the erosion rounds as hipGraphs (default 1)
 *   "sg.kc" "20"|"27"|"45", "sg.kc_tiles" "27"|"45", "sg.rowgroup" "1".."1024"      LDS chunking / tile walk of the exact sine kernel
 *   "tile_erosion"         "lds" | "window"   the whole padded tile in LDS (default) or a 32 x 32 window over a copy in HBM
 *   "weights.simple"       "0" | "1"      per-texel form of the weights-texture pass;   "shadows.levels" "0" | "1"   one launch per dependency level of the mesh shadows
 *   "ao.bands"             "1" | "0"      the AO context of a tile batch evaluated as four bands around each tile (its centre is the tile's own heights) / as whole squares
 *   "ao.whole"             "1" | "0"      the AO rays of a tile from one workgroup that holds the tile's whole 201 x 201 context in LDS / from four 33-row band workgroups
 *   "voxels.cols"          "1" | "0"      the lane-per-column form of the voxel sine field (no array of x*y products) wherever the depth is a multiple of 4 / the z-lane form everywhere
 *   "ero.sparse"           "0" | "1" | "auto"   never / always / by droplet density try the sparse erosion scheduler;   "ero.sparse_retraces" n
 *   "ero.lead" "0".."2", "ero.batch" n, "ero.fuse" 0..7, "ero.live" "0"|"1", "ero.diag" "0"|"1", "ero.ck" "steps:max"|"default", "ero.near" n|"default", "ero.mem_budget" bytes|"-1"
 *                                          scheduling knobs of the multi-version erosion scheduler (terra_set_erosion_tuning has the documented ones) */
int  terra_set_option(terra_ctx *ctx, const char *key, const char *value);
/* ---- whole grids between host and device.  The reference's callers own HOST arrays (cached_vals of build_arrays, src/mesh_gen.cpp:597-603; apply_erosion's float*,
 * src/erosion.cpp:14; heightmap_t's pixels, src/heightmap.cpp:130-151), 1 GiB at 16384^2.  Every host-pointer entry point moves arrays of >= 16 MiB in 8 MiB bands on four
 * streams at once through pinned staging (csrc/terra_xfer.hpp); an array allocated with terra_host_alloc (pinned) is the DMA target itself, no staging copy.
 * terra_download_async: d_src -> h_dst behind everything enqueued on ctx so far, without blocking the host or the context's later kernels (the next map's noise runs
 * beside the copy); h_dst is complete when terra_download_wait returns.  One context = one thread at a time, as everywhere. */
void *terra_host_alloc(size_t bytes);   /* NULL + terra_last_error() when pinned memory is exhausted */
void terra_host_free(void *p);
int  terra_download_async(terra_ctx *ctx, const void *d_src, void *h_dst, size_t bytes);
int  terra_download_wait(terra_ctx *ctx);
/* give every grow-only work buffer of the context back to the device (synchronises first; they grow again on demand).  The one that matters is the speculation ring of
 * terra_apply_erosion_dev: ~266 KiB per droplet in flight, 8.5 GiB for a 16384^2 map -- its size is also capped by the memory that is free when it has to grow. */
int  terra_release_scratch(terra_ctx *ctx);

/* ---- events: stream-level ordering between contexts (hipEventRecord / hipStreamWaitEvent; the host never blocks).  The engine keeps several generator objects in
 * flight (height_gens[8], src/tiled_mesh.h:418); here one context can produce (noise of map i, recorded) what another consumes (erosion of map i, its stream waits):
 * terra_event_record marks everything enqueued so far on ctx's stream, terra_event_wait makes everything enqueued on ctx's stream from now on wait for the last
 * record of ev.  Record before you wait (host order is the caller's business).  An event may be recorded again once its waits have been enqueued. */
typedef struct terra_event terra_event;
int  terra_event_create(terra_ctx *ctx, terra_event **out);
int  terra_event_record(terra_ctx *ctx, terra_event *ev);
int  terra_event_wait(terra_ctx *ctx, terra_event *ev);
/* the HOST waits until everything before the last record of ev has completed (hipEventSynchronize; returns at once for an event that was never recorded).  A thread that
 * drives the next heightmap can wait for the previous map's noise kernel itself -- not for the thread that launched it to wake up, notice and tell it */
int  terra_event_synchronize(terra_event *ev);
void terra_event_destroy(terra_event *ev);

/* ---- scene / globals.  terra_init_scene = main()'s start-up sequence for this path (src/3DWorld.cpp:2393-2460 -> gen_mesh). */
int  terra_init_scene(terra_ctx *ctx, const terra_config *cfg);
int  terra_set_config(terra_ctx *ctx, const terra_config *cfg);   /* store the config-file values only (engine integration: follow with terra_set_state) */
int  terra_get_state(terra_ctx *ctx, terra_state *out);
int  terra_set_state(terra_ctx *ctx, const terra_state *in);
int  terra_set_mode(terra_ctx *ctx, int mesh_gen_mode, int mesh_gen_shape);
int  terra_set_zmax_est(terra_ctx *ctx, float zmax_est);          /* set_zmax_est + zmin/zmax/water_plane_z, src/mesh_gen.cpp:162-167,494-512 */
int  terra_set_water_plane_z(terra_ctx *ctx, float water_plane_z);
int  terra_set_start_eval_sin(terra_ctx *ctx, int start_eval_sin);
int  terra_set_erode_amount(terra_ctx *ctx, float erode_amount);
float terra_get_max_sea_level(terra_ctx *ctx);                    /* src/tiled_mesh.cpp:141 */

/* ---- generator: mesh_xy_grid_cache_t (src/mesh.h:22-45).  One handle per in-flight grid, like height_gens[8] (src/tiled_mesh.h:418). */
int  terra_gen_create(terra_ctx *ctx, terra_gen **out);
void terra_gen_destroy(terra_gen *g);                              /* ~mesh_xy_grid_cache_t / clear_context */
/* build_arrays: returns 1 = results available, 0 = launched and not ready (only with TERRA_GEN_NO_WAIT), <0 = error.
 * Same async protocol as the GL path (src/mesh_gen.cpp:597-603): call again with the same arguments to collect.
 * min_start_sin: the first sine term the caller's eval_index() calls will ask for (0 when unknown): the device grid is evaluated from
 * max(start_eval_sin, min_start_sin), e.g. 50 for tile_t::create_texture's noise field (src/tiled_mesh.cpp:1099,1114). */
int  terra_gen_build_arrays(terra_gen *g, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin);
int  terra_gen_enable_glaciate(terra_gen *g);                      /* must follow build_arrays, as in the reference */
int  terra_gen_is_running(terra_gen *g);                           /* compute_shader_t::get_is_running (src/shaders.h:233) */
int  terra_gen_collect(terra_gen *g, float *host_out);            /* blocks; copies nx*ny floats (cached_vals) */
/* eval_index(x, y, min_start_sin, use_cache) (src/mesh.h:42, src/mesh_gen.cpp:754-792) on collected values.  Sine mode: the sum starts at
 * max(start_eval_sin, min_start_sin) unless use_cache && TERRA_GEN_CACHE_VALUES was given (cached values are built with min_start_sin = 0); when that is not
 * the first term the device grid was evaluated with, the grid is evaluated once more for it (kept until the next build).  fBm modes ignore both arguments. */
float terra_gen_eval_index(terra_gen *g, uint32_t x, uint32_t y, int min_start_sin, int use_cache);
const float *terra_gen_device_values(terra_gen *g);                /* device pointer to the nx*ny grid (valid until the next build) */

/* one-shot: build_arrays + [enable_glaciate] + the caller's eval_index double loop (src/heightmap.cpp:135-143, src/tiled_mesh.cpp:495-514) */
int  terra_gen_grid_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *d_out);
/* same, plus min(vals)/max(vals) folded into the grid kernel (what heightmap_t::run_erosion / get_heightmap_z_range compute next); synchronous */
int  terra_gen_grid_minmax_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *d_out, float *h_min, float *h_max);
/* the same with min / max left in DEVICE memory (d_minmax: 2 floats) and nothing read back: asynchronous.  With terra_apply_erosion_devmin_dev the whole
 * heightmap_t::proc_gen step (src/heightmap.cpp:135-169: eval loop, min(vals), apply_erosion) is enqueued without a host round trip in between */
int  terra_gen_grid_minmax_async_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *d_out, float *d_minmax);
int  terra_gen_grid(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *h_out);
/* rows [row0, row0 + nrows) of the nx x ny grid only, d_out = nrows*nx floats; bit-identical to the same rows of the full-grid call (heightmap_t::proc_gen's
 * row loop is independent per row, src/heightmap.cpp:139-143): one heightmap as row strips on several GPUs.  h_min / h_max (optional, synchronous when given):
 * min / max of the strip -- min(vals) of the whole map is the minimum over the strips (one float through ncclAllReduce(min), see bench.py --workload strips). */
int  terra_gen_grid_rows_minmax_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin,
                                    uint32_t row0, uint32_t nrows, float *d_out, float *h_min, float *h_max);
/* the same with the strip's {min, max} left in DEVICE memory (d_minmax: 2 floats), nothing read back: the call only enqueues.  One rank's part of a step of the one-grid
 * pipeline with no host round trip: the strip's noise, ncclAllReduce(min) of d_minmax[0] on the same stream, terra_apply_erosion_devmin_dev on the eroding rank
 * (tools/bench_native_onegrid.c) */
int  terra_gen_grid_rows_minmax_async_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin,
                                          uint32_t row0, uint32_t nrows, float *d_out, float *d_minmax);

/* ---- point query and ground-mode post-pass
 * eval_mesh_sin_terms (src/mesh_gen.cpp:797-805): non-separable point query used for biome parameters / collision height; evaluated on the host.
 * glaciate() (src/mesh_gen.cpp:388-404): in-place apply_glaciate + apply_mesh_sine over the MESH_X x MESH_Y ground mesh (xoff2/yoff2 = scroll offsets);
 * h_zbottom_ztop (optional) receives {zbottom, ztop}. */
int  terra_eval_mesh_sin_terms(terra_ctx *ctx, float xv, float yv, float *out);
/* The all-modes point queries, batched (n points, xy[2 i] = x, xy[2 i + 1] = y), evaluated on the device:
 *   TERRA_POINTS_SCALED  out[i] = eval_mesh_sin_terms_scaled(x, y, xy_scale) (src/mesh_gen.cpp:807-813): index-space coordinates; sine mode = the sine sum scaled and shaped,
 *                        the fBm modes go to get_noise_zval (the detail noise of heightmap tiles src/tiled_mesh.cpp:499-503, density fields)
 *   TERRA_POINTS_EXACT   out[i] = get_exact_zval(x, y, no_xyoff) (src/mesh_gen.cpp:816-847): world-space point -> index space (+ xoff2 / yoff2, the reference's scroll-offset
 *                        globals, unless no_xyoff) -> the heightmap texture (+ detail noise) when terra_hmap_set_dev gave one, else noise + apply_glaciate +
 *                        apply_mesh_sine (islands, volcano).  What collision / building placement / biome code calls (src/tiled_mesh.cpp:332-338, src/voxels.cpp:434).
 * Not covered: get_exact_zval's two branches that only read the caller's state -- the ground-mode mesh_height[][] look-up (:821-825) and the constant returned while a named
 * texture is not loaded yet (:839-843).  xy_scale is ignored by TERRA_POINTS_EXACT; no_xyoff / xoff2 / yoff2 by TERRA_POINTS_SCALED. */
#define TERRA_POINTS_SCALED 0u
#define TERRA_POINTS_EXACT  1u
int  terra_eval_points(terra_ctx *ctx, const float *xy, uint32_t n, uint32_t kind, float xy_scale, int no_xyoff, int xoff2, int yoff2, float *out);
int  terra_eval_points_dev(terra_ctx *ctx, const float *d_xy, uint32_t n, uint32_t kind, float xy_scale, int no_xyoff, int xoff2, int yoff2, float *d_out);
int  terra_glaciate_mesh_dev(terra_ctx *ctx, float *d_mesh, uint32_t nx, uint32_t ny, int xoff2, int yoff2, float *h_zbottom_ztop);

/* ---- erosion: apply_erosion (src/erosion.cpp:14).  In place; silently returns TERRA_OK when num_iters == 0 or erode_amount <= 0.
 * num_iters (and erosion_iters_tt of the tile calls) above 27 183 336 is TERRA_ERR_ARG: the reference's `int` seed 79*iter+121 of droplet iter = 27 183 336 overflows (undefined). */
int  terra_apply_erosion_dev(terra_ctx *ctx, float *d_heightmap, int xsize, int ysize, float min_zval, uint32_t num_iters, uint32_t flags);
/* min_zval read from device memory (one float, e.g. d_minmax of terra_gen_grid_minmax_async_dev -- possibly written by ANOTHER context's stream that this context's
 * stream was made to wait for) when the final clamp runs, the only use apply_erosion makes of it (src/erosion.cpp:158-162) */
int  terra_apply_erosion_devmin_dev(terra_ctx *ctx, float *d_heightmap, int xsize, int ysize, const float *d_min_zval, uint32_t num_iters, uint32_t flags);
int  terra_apply_erosion(terra_ctx *ctx, float *h_heightmap, int xsize, int ysize, float min_zval, uint32_t num_iters);
/* ---- the erosion of ONE grid whose row strips live on several GPUs (terra_dgrid below), with the sparse scheduler's read-only phases run where the
 * rows live.  apply_erosion (src/erosion.cpp:14) is one serial droplet order over the whole map, so it is the eroding rank that checks and commits -- but the first step of
 * every droplet and the trace of the ones that move read the ORIGINAL grid only, and a droplet spends its life near where it starts: rank r probes / traces the droplets
 * that start in its rows into its own HBM, the eroding rank fetches the (small) traces over xGMI instead of walking remote rows window by window.
 *   every rank r (the eroding one too), once the whole grid is written:
 *       terra_erosion_shard_trace_dev(ctx, d_grid, xsize, ysize, num_iters, row0_r, nrows_r, d_arena_r)
 *   the eroding rank, once every rank's trace call has completed (the caller's collective / event):
 *       terra_erosion_shard_finish_dev(ctx, d_grid, xsize, ysize, d_min_zval, num_iters, flags, world, self, row_end, d_arena_self, arena_stride)
 *         = terra_apply_erosion_devmin_dev on the same grid, bit for bit (the traces are the ones it would have made itself).
 * d_arena_r: terra_erosion_shard_arena_bytes() bytes of rank r's memory, all of them mapped on the eroding rank arena_stride bytes apart in rank order (one more
 * terra_dgrid whose strips are the arenas); row_end[r] = first row after rank r's strip (row_end[world - 1] = ysize); world <= 16.  Runs the sparse scheduler would not
 * take (many droplets on a small map) make the trace call a no-op and the finish call the ordinary erosion.  Not with TERRA_ERODE_SERIAL*. */
size_t terra_erosion_shard_arena_bytes(terra_ctx *ctx, uint32_t num_iters);
int  terra_erosion_shard_trace_dev(terra_ctx *ctx, float *d_heightmap, int xsize, int ysize, uint32_t num_iters, uint32_t row0, uint32_t nrows, void *d_arena);
int  terra_erosion_shard_finish_dev(terra_ctx *ctx, float *d_heightmap, int xsize, int ysize, const float *d_min_zval, uint32_t num_iters, uint32_t flags,
                                    uint32_t world, uint32_t self, const uint32_t *row_end, void *d_arena_self, size_t arena_stride);
int  terra_get_erosion_report(terra_ctx *ctx, terra_erosion_report *out);
/* tuning of the speculative scheduler (0 keeps a value): droplets in flight (ring slots; default automatic from the grid size, 0xFFFFFFFF restores that; at most
 * 2^20 is accepted, and a run uses at most (2^25 - 1)/block_list_capacity slots: version pages are addressed with 31-bit float indices),
 * log_capacity_log2: ignored (range-checked only) -- a droplet's writes are kept as one 64-cell page per 8x8 block of its footprint, there is no hashed log to size,
 * per-droplet block-list capacity = pages per droplet (16 .. 256, larger values mean 256).  A droplet whose footprint overflows it runs alone, in order, directly on
 * the grid (still exact).
 * slice_steps: while droplets wait for a slot a trace advances at most this many steps per round (default 96), except the 128 droplets next in line for the
 * commit, which trace to the end; a droplet's first trace is visible to the droplets after it from its first slice on (what it has written back so far).
 * Results never depend on any of these. */
int  terra_set_erosion_tuning(terra_ctx *ctx, uint32_t window, uint32_t log_capacity_log2, uint32_t block_list_capacity);
int  terra_set_erosion_slice_steps(terra_ctx *ctx, uint32_t slice_steps);

/* ---- whole heightmap: heightmap_t::proc_gen (src/heightmap.cpp:130-151) minus run_city_gen.
 * d_vals: width*height floats (final z); d_pixels16: optional 2 bytes per pixel {lo, hi} (from_floats/write_pixel_16_bits);
 * h_range: optional {min_z, dz} used for the 16-bit scale. */
int  terra_heightmap_proc_gen_dev(terra_ctx *ctx, uint32_t width, uint32_t height, uint32_t erosion_iters, float *d_vals, uint8_t *d_pixels16, float *h_range);
/* the same with the 16-bit pixels delivered to HOST memory (h_pixels16: 2*width*height bytes, what proc_gen leaves in the texture): the drop-in for the whole body of
 * heightmap_t::proc_gen when there are no cities -- 2 bytes per cell cross the host link instead of three float grids */
int  terra_heightmap_proc_gen(terra_ctx *ctx, uint32_t width, uint32_t height, uint32_t erosion_iters, uint8_t *h_pixels16, float *h_range);
int  terra_minmax_dev(terra_ctx *ctx, const float *d_vals, size_t n, float *h_min, float *h_max); /* synchronous */
int  terra_quantize16_dev(terra_ctx *ctx, const float *d_vals, size_t n, float min_z, float dz, uint8_t *d_pixels16);

/* ---- the loaded-heightmap path: heightmap_t::to_floats / from_floats / postprocess_height (src/heightmap.cpp:117-128,191-215), what
 * terrain_hmap_manager_t::load runs right after reading the PNG (src/heightmap.cpp:351), e.g. scene_config/config_heightmap.txt:78-87.
 * terra_set_mesh_file_scale = the two numbers of the config line `mh_filename <png> <mesh_file_scale> <mesh_file_tz>` (src/3DWorld.cpp:2205): a pixel value
 * v in [0, 256) is the height get_mh_texture_mult()*v + get_mh_texture_add() (src/mesh_gen.cpp:122-123); terra_set_mesh_height_scales_for_zval_range sets the same pair.
 * d_pixels: width*height pixels in device memory, 1 byte each or 2 = {fraction, integer}.
 * to_floats: pixels -> heights.  from_floats: heights -> pixels with the scale in force (8-bit: truncation; 16-bit: write_pixel_16_bits).
 * postprocess: to_floats -> run_erosion (apply_erosion over the whole image, min_zval = min(vals), erosion_iters_tt droplets) -> from_floats, in place;
 *   nothing happens when erosion_iters_tt == 0 (run_city_gen is another subsystem).  d_vals: width*height floats of device scratch that is left holding the
 *   eroded heights, or NULL (allocated internally).
 * h_out_of_range (optional): number of values that map outside [0, 256) -- the reference asserts on those (src/heightmap.cpp:210); with NULL such an image is
 * TERRA_ERR_STATE (the pixels are still written, with the x86 conversion's wrap-around). */
int  terra_set_mesh_file_scale(terra_ctx *ctx, float mesh_file_scale, float mesh_file_tz);
int  terra_get_mesh_file_scale(terra_ctx *ctx, float *mesh_file_scale, float *mesh_file_tz);
int  terra_heightmap_to_floats_dev(terra_ctx *ctx, const uint8_t *d_pixels, uint32_t width, uint32_t height, int ncolors, float *d_vals);
int  terra_heightmap_from_floats_dev(terra_ctx *ctx, const float *d_vals, uint32_t width, uint32_t height, int ncolors, uint8_t *d_pixels, uint32_t *h_out_of_range);
int  terra_heightmap_postprocess_dev(terra_ctx *ctx, uint8_t *d_pixels, uint32_t width, uint32_t height, int ncolors, uint32_t erosion_iters_tt, float *d_vals, uint32_t *h_out_of_range);

/* ---- the ground mesh's text file: read_mesh / write_mesh (src/mesh_gen.cpp:895-965), e.g. BASELINE config 1's `mesh_file mapx/mesh128.txt` (src/3DWorld.cpp:2206).
 * Format: "nx ny" then ny rows of nx heights (written with "%f ").  Host functions, like the PNG ones: the file is a few hundred KB and is parsed with the C library's
 * own fscanf, so every height has the bits the reference reads.
 * terra_read_mesh: the header must equal (nx, ny) = the scene's MESH_X_SIZE x MESH_Y_SIZE (the reference refuses other sizes); h_mesh[i*nx + j] = mesh_file_scale*height +
 *   mesh_file_tz (terra_set_mesh_file_scale); then calc_zminmax, set_zmax_est(zmm != 0 ? zmm : max(-zmin, zmax)) and set_zvals: the context's zmin / zmax / zmax_est /
 *   water_plane_z are what the engine's globals are after read_mesh (terra_get_state).  h_zbottom_ztop (optional) receives the mesh's own {min, max} (set_zvals' zbottom / ztop).
 *   Returns TERRA_ERR_ARG for a missing file, a short file or a size mismatch (the reference prints an error and returns 0); h_mesh may then be partly written, the state is untouched.
 * terra_write_mesh: the inverse (no scale is applied, as in the reference). */
int  terra_read_mesh(terra_ctx *ctx, const char *filename, float zmm, float *h_mesh, uint32_t nx, uint32_t ny, float *h_zbottom_ztop);
int  terra_write_mesh(const char *filename, const float *h_mesh, uint32_t nx, uint32_t ny);

/* ---- tiles: tile_t::create_zvals batch, size = 128 (zvsize 130, stride 129).
 * tile_xy: n pairs (tile x, tile y) on the HOST.  d_zvals: n*130*130 floats.  d_stats: n terra_tile_stats (optional).
 * d_normals: n*129*129*4 bytes RGBA8 with A = 0 (optional); d_min_normal_z: n floats (optional). */
int  terra_tiles_create_zvals_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, uint32_t erosion_iters_tt,
                                  float *d_zvals, terra_tile_stats *d_stats, uint8_t *d_normals, float *d_min_normal_z);
int  terra_tiles_create_zvals(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, uint32_t erosion_iters_tt,
                              float *h_zvals, terra_tile_stats *h_stats, uint8_t *h_normals, float *h_min_normal_z);
/* the stats + normals pass alone, over zvals the caller HAS (edited with the brushes, eroded elsewhere, read from a file): the sub-block / water-bbox loop of
 * tile_t::create_zvals (src/tiled_mesh.cpp:517-541) and tile_t::upload_normal_texture (src/tiled_mesh.cpp:865-880), which the engine also runs on its own whenever a
 * tile's heights changed.  d_zvals is not written.  Outputs as above (each optional). */
int  terra_tiles_post_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, terra_tile_stats *d_stats, uint8_t *d_normals, float *d_min_normal_z);
int  terra_tiles_post(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, terra_tile_stats *h_stats, uint8_t *h_normals, float *h_min_normal_z); /* host arrays */
/* self test: the droplet step takes its two square roots per step with a shortened instruction sequence (csrc/terra_erosion.hpp: sqrt_rn); this runs it over every
 * stride-th fp32 bit pattern (stride 1: all 2^32, a few ms on the GPU) against sqrtf and against the correctly rounded double-precision route; the tile normals' byte test
 * (csrc/terra_kernels.hpp: k_tile_post) relies on the hardware's reciprocal square root being within 2^-23 of the real one, checked over the same inputs.  *mismatches must be 0. */
int  terra_selftest_hot_sqrt(terra_ctx *ctx, uint32_t stride, uint64_t *mismatches);

/* ---- heightmap files, host side: 8- / 16-bit grayscale PNG exactly as the reference reads / writes them through libpng (src/image_io.cpp:493-605):
 * write: rows in memory order, 16-bit pixels {fraction, integer} -> big-endian samples (heightmap_t::write_png, src/heightmap.cpp:375-378);
 * read: file row i -> memory row height-1-i (texture_t::load_png flips), allow_two_byte_grayscale keeps 16 bits as 2 bytes per pixel, otherwise the
 * high byte only.  h_pixels may be NULL to query the size (width*height*ncolors bytes).  Non-grayscale / interlaced files are refused. */
int  terra_heightmap_write_png(const char *path, const uint8_t *h_pixels, uint32_t width, uint32_t height, int ncolors);
int  terra_heightmap_read_png(const char *path, int allow_two_byte_grayscale, uint32_t *width, uint32_t *height, int *ncolors, uint8_t *h_pixels, size_t capacity);

/* ---- tiles from a heightmap texture instead of the procedural generator: terrain_hmap_manager_t (src/heightmap.h:110-142).
 * d_pixels: width*height pixels in DEVICE memory, 1 byte each or 2 = {fraction, integer} as written by terra_quantize16_dev / write_pixel_16_bits;
 * the library keeps the pointer (no copy), NULL switches back to procedural tiles.  While set, terra_tiles_create_zvals samples
 * get_clamped_height(x1 + x, y1 + y) (nearest for mesh_scale >= 1, bilinear below, mirror-wrapped outside the image; src/heightmap.cpp:310-402), adds
 * HMAP_DETAIL_MAG * the detail noise grid when mesh_scale < 0.75, skips erosion (src/tiled_mesh.cpp:499-503,515) and the AO context does the same (:623-627).
 * terra_set_mesh_height_scales_for_zval_range = src/mesh_gen.cpp:125-131 (pixel value -> height, e.g. range[0], range[1]/255 of terra_heightmap_proc_gen_dev). */
int  terra_hmap_set_dev(terra_ctx *ctx, const uint8_t *d_pixels, int width, int height, int ncolors);
int  terra_set_mesh_height_scales_for_zval_range(terra_ctx *ctx, float min_z, float dz);

/* ---- height edits of the heightmap texture (tex_mod_map_manager_t / terrain_hmap_manager_t, src/heightmap.h:38-142, src/heightmap.cpp:27-58,99-115,216-308,
 * 414-440): brushes, the per-texel mod map and the .mod file that stores both.  The records have the reference's in-memory = on-disk layouts.
 * The *_dev calls edit the image registered with terra_hmap_set_dev in place (it must be writable device memory, at most 65536 texels per side, 16-bit
 * images 2-byte aligned; a texel is updated by a compare-and-swap on the aligned 32-bit word around it, so when the image does not start / end on a
 * 4-byte boundary the up to 3 neighbouring bytes are re-written with their own values: nothing else may modify them during an edit call); a brush
 * point (xp, yp, sub-step) lands on texel clamp_xy(xp + dx, yp + dy) (mesh_scale, mirror wrap) and adds round_fp(delta * weight(shape, dist / radius)) with
 * clamping to the pixel range, flatten brushes store delta.  Brushes are applied in list order (apply_cur_brushes); within one brush the reference's
 * OpenMP loop order is immaterial (same-signed saturating adds commute) and so is the thread order here. */
enum {TERRA_BSHAPE_CONST_SQ = 0, TERRA_BSHAPE_CNST_CIR, TERRA_BSHAPE_LINEAR, TERRA_BSHAPE_QUADRATIC, TERRA_BSHAPE_COSINE, TERRA_BSHAPE_SINE, TERRA_BSHAPE_FLAT_SQ, TERRA_BSHAPE_FLAT_CIR, TERRA_NUM_BSHAPES};
typedef struct terra_hmap_brush {int32_t x, y; uint32_t radius; int32_t delta; int16_t shape;} terra_hmap_brush; /* hmap_brush_t, src/heightmap.h:71-76 (20 bytes) */
typedef struct terra_hmap_mod {uint16_t x, y; int32_t delta;} terra_hmap_mod;                                    /* mod_elem_t, src/heightmap.h:59-64 (8 bytes) */
int  terra_hmap_apply_brushes_dev(terra_ctx *ctx, const terra_hmap_brush *brushes, uint32_t n, int step_sz, uint32_t num_steps); /* apply_brush(brush, step_sz, num_steps) for each */
int  terra_hmap_apply_mods_dev(terra_ctx *ctx, const terra_hmap_mod *mods, uint32_t n);      /* add_mod for each (deltas of one texel are summed) + apply_cur_mod_map */
int  terra_hmap_read_and_apply_mod_dev(terra_ctx *ctx, const char *path);                   /* read_and_apply_mod: mods, then brushes with step_sz = num_steps = 1 */
int  terra_hmap_write_mod(const char *path, const terra_hmap_mod *mods, uint32_t n, const terra_hmap_brush *brushes, uint32_t n_brushes); /* write_mod */
/* read_mod: counts always; records when the buffers are non-NULL and large enough (capacities in records).  Mods come back combined, in map order. */
int  terra_hmap_read_mod(const char *path, terra_hmap_mod *mods, uint32_t mods_capacity, uint32_t *n_mods, terra_hmap_brush *brushes, uint32_t brushes_capacity, uint32_t *n_brushes);
/* ---- the map-view heightmap exporter write_map_mode_heightmap_image (src/map_view.cpp:409-442) from the image origin on: width x height cells starting at
 * scene position (xstart, ystart) with the mesh spacing, rows inverted, 16-bit pixels = (h - min_z) * (255 / dz).  d_vals: width*height floats (the
 * reference's `heights`), d_pixels16: 2 bytes per pixel or NULL, h_min_z_dz: {min_z, dz} or NULL.  With a heightmap texture set the heights are sampled
 * from it (get_mesh_height, src/map_view.cpp:97-105).  terra_write_map_mode_heightmap_image = the same + the PNG file. */
int  terra_export_heightmap_dev(terra_ctx *ctx, float xstart, float ystart, uint32_t width, uint32_t height, float *d_vals, uint8_t *d_pixels16, float *h_min_z_dz);
int  terra_write_map_mode_heightmap_image(terra_ctx *ctx, const char *path, float xstart, float ystart, uint32_t width, uint32_t height);

/* ---- tile ambient-occlusion lighting: tile_t::calc_mesh_ao_lighting (src/tiled_mesh.cpp:586-661).  zvals: [n][130][130] exactly as
 * terra_tiles_create_zvals left them (eroded or not); ao: [n][129][129] bytes = (unsigned char)(255*(1 - atten/64)).  The 201 x 201 context
 * around each tile is generated internally (setup_height_gen_async(x1 - 36, y1 - 36, 201, 201)).
 * terra_set_tiled_mesh_ao = the config key enable_tiled_mesh_ao (src/3DWorld.cpp:1778; scene_config/config.txt turns it on): with mesh_gen_mode >= 3 the
 * reference clips a tile's zvals out of that context grid instead of a 130 x 130 grid of its own (src/tiled_mesh.cpp:478-488,505), which changes
 * their low bits; terra_tiles_create_zvals follows the flag. */
int  terra_set_tiled_mesh_ao(terra_ctx *ctx, int enable);
int  terra_tiles_ao_lighting_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, uint8_t *d_ao);
int  terra_tiles_ao_lighting(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, uint8_t *h_ao);

/* ---- landscape weights texture of a tile: tile_t::create_texture (src/tiled_mesh.cpp:1071-1240) with update_terrain_params (:321-343), get_tids /
 * update_lttex_ix (src/Textures.cpp:1289-1316) and add_grass_block_at (src/tiled_mesh.cpp:1354-1371).  Terrain-only branch: the city, tunnel and building
 * queries and the tree map come from subsystems outside this library (their texels are the caller's to overwrite afterwards, as the reference does).
 * zvals: [n][130][130]; weights: [n][129][129][4] bytes RGBA = {sand, dirt, grass, rock}, snow = remainder; grass_blocks: [n][32][32] or NULL;
 * has_any_grass: [n] bytes or NULL.  The second noise field (build_arrays(..., 80*DX_VAL, 80*DY_VAL, 129, 129, 0, force_sine_mode=1) + eval_index(x, y, 50))
 * and the biome parameters are generated internally.  terra_tiles_terrain_params returns those parameters: [n][2][2][3] = [yp][xp]{veg, grass, dirt}. */
int  terra_set_landscape(terra_ctx *ctx, const terra_landscape *params);
int  terra_get_landscape(terra_ctx *ctx, terra_landscape *out);
int  terra_tiles_terrain_params(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, float *h_params);
int  terra_tiles_create_weights_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, uint8_t *d_weights, terra_grass_block *d_grass_blocks, uint8_t *d_has_any_grass);
int  terra_tiles_create_weights(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, uint8_t *h_weights, terra_grass_block *h_grass_blocks, uint8_t *h_has_any_grass);

/* ---- tile mesh shadows of one directional light: tile_t::calc_shadows_for_light + calc_mesh_shadows / mesh_shadow_gen (src/tiled_mesh.cpp:664-692,
 * src/visibility.cpp:411-520).  zvals: [n][130][130]; light_pos: the light's position vector (get_light_pos(l)); smask: [n][130][130] bytes, 0 or
 * MESH_SHADOW (0x02).  Shadows cross tile borders: a tile starts its sweeps from the edge heights left by its neighbours toward the light when those are
 * part of the batch (sh_out -> sh_in), otherwise from nothing, exactly like a tile whose neighbour does not exist.  Order of evaluation = the
 * reference's single-threaded order (its two OpenMP sections race). */
int  terra_tiles_mesh_shadows_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, const float light_pos[3], uint8_t *d_smask);
int  terra_tiles_mesh_shadows(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, const float light_pos[3], uint8_t *h_smask);
/* The same for a batch that is only PART of the terrain (the rest lives on another GPU / rank): h_edge_in[i][0] = sh_in_x, h_edge_in[i][1] = sh_in_y of
 * tile i (130 floats each, host memory), used where h_edge_in_present[i][d] != 0 and the neighbour toward the light is not in the batch; h_edge_out[i][0..1]
 * receives every tile's sh_out_x / sh_out_y (MESH_MIN_Z = -1e6 where nothing was written) -- what the owner of the next tile away from the light needs.
 * All three are optional (NULL).  3dworld_amd/dist.py: tile strips over torch.distributed ranks, the border edges travel by send/recv. */
int  terra_tiles_mesh_shadows_halo_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, const float light_pos[3], uint8_t *d_smask,
                                       const float *h_edge_in, const uint8_t *h_edge_in_present, float *h_edge_out);

/* ---- voxels: voxel_manager::create_procedural fill (src/voxels.cpp:278-346).  out is z-fastest: ix = z + (x + y*nx)*nz (src/voxels.h:141-144). */
int  terra_voxel_fill_dev(terra_ctx *ctx, float *d_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo_pos[3], const float vsz[3], const float offset[3],
                          float mag, float freq, int rseed1, int rseed2, int gen_mode, float zscale, int normalize_to_1);
/* the y slab [y0, y0 + nys) of the same field: d_out = nys*nx*nz floats, bit-identical to those rows of the full-grid call (voxels are independent; the axis
 * positions are the full grid's running sums, src/upsurface.cpp:41-57): one field as y slabs on several GPUs, no collective (3dworld_amd/dist.py, bench.py) */
int  terra_voxel_fill_slab_dev(terra_ctx *ctx, float *d_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo_pos[3], const float vsz[3], const float offset[3],
                               float mag, float freq, int rseed1, int rseed2, int gen_mode, float zscale, int normalize_to_1, uint32_t y0, uint32_t nys);
int  terra_voxel_fill(terra_ctx *ctx, float *h_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo_pos[3], const float vsz[3], const float offset[3],
                      float mag, float freq, int rseed1, int rseed2, int gen_mode, float zscale, int normalize_to_1);

/* ---- several GPUs from ONE host process.  3DWorld is a single C++ process that keeps eight generator objects in flight and collects them as they finish
 * (height_gens[8], src/tiled_mesh.h:418; src/tiled_mesh.cpp:2317,2367-2416).  terra_multi = N contexts, one per entry of device_indices (an index may repeat:
 * several contexts on one GPU), each driven by its own host thread for the duration of a call.  Units are dealt out in contiguous blocks -- context i gets
 * terra_multi_partition(n_units, size, i) -- of independent work: tiles (no neighbour data, src/tiled_mesh.cpp:515), rows of one heightmap
 * (src/heightmap.cpp:139-143), y slabs of a voxel field; nothing there is collective.  In the *_dev forms the outputs are arrays of per-context device
 * pointers (entry i lives on context i's device and holds its block).  terra_multi_foreach runs fn(ctx, index, user) on every context's thread at once for
 * everything else (e.g. one heightmap region per GPU: terra_gen_grid_minmax_dev + terra_apply_erosion_dev); a negative return is the call's error.
 * terra_multi_tiles_mesh_shadows: the one pass of the tile path with a cross-tile dependency (tile_t::calc_shadows_for_light, src/tiled_mesh.cpp:664-692) --
 * tile columns in strips, one per context, rows pipelined through the strips, the border tiles' outgoing edges handed from device to device
 * (hipMemcpyPeerAsync); host zvals in ([n][130][130]), host shadow masks out ([n][130][130]), same result as terra_tiles_mesh_shadows on one context. */
typedef struct terra_multi terra_multi;
int  terra_multi_create(terra_multi **out, const int *device_indices, uint32_t n);
void terra_multi_destroy(terra_multi *m);
uint32_t terra_multi_size(const terra_multi *m);
terra_ctx *terra_multi_ctx(terra_multi *m, uint32_t i);
void terra_multi_partition(uint32_t n_units, uint32_t n_parts, uint32_t part, uint32_t *first, uint32_t *count);
int  terra_multi_foreach(terra_multi *m, int (*fn)(terra_ctx *ctx, uint32_t index, void *user), void *user);
int  terra_multi_synchronize(terra_multi *m);
int  terra_multi_init_scene(terra_multi *m, const terra_config *cfg);
int  terra_multi_tiles_create_zvals_dev(terra_multi *m, const int32_t *tile_xy, uint32_t n, uint32_t erosion_iters_tt,
                                        float *const *d_zvals, terra_tile_stats *const *d_stats, uint8_t *const *d_normals, float *const *d_min_normal_z);
int  terra_multi_tiles_create_zvals(terra_multi *m, const int32_t *tile_xy, uint32_t n, uint32_t erosion_iters_tt,
                                    float *h_zvals, terra_tile_stats *h_stats, uint8_t *h_normals, float *h_min_normal_z);
int  terra_multi_gen_grid_rows_dev(terra_multi *m, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin,
                                   float *const *d_out, float *h_min, float *h_max);
int  terra_multi_voxel_fill_dev(terra_multi *m, float *const *d_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo_pos[3], const float vsz[3], const float offset[3],
                                float mag, float freq, int rseed1, int rseed2, int gen_mode, float zscale, int normalize_to_1);
int  terra_multi_tiles_mesh_shadows(terra_multi *m, const int32_t *tile_xy, uint32_t n, const float *h_zvals, const float light_pos[3], uint8_t *h_smask);
/* the device-resident form: the terrain already lies on the GPUs as strips of tile columns.  terra_multi_shadow_layout tells where: tile i belongs to context
 * ctx_of_tile[i] and is the pos_in_ctx[i]-th tile of that context's arrays (tiles_per_ctx[s] tiles on context s: rows toward the light first); d_zvals[s] / d_smask[s] are
 * [tiles_per_ctx[s]][130][130] on context s's device.  Between strips only the border tiles' 130-float edges move, device to device: an event behind a chunk's kernels, the
 * next strip's stream waits for it and gathers the edges out of the peer's buffer in one launch (xGMI; staged copies when the devices cannot map each other). */
int  terra_multi_shadow_layout(terra_multi *m, const int32_t *tile_xy, uint32_t n, const float light_pos[3], uint32_t *ctx_of_tile, uint32_t *pos_in_ctx, uint32_t *tiles_per_ctx);
int  terra_multi_tiles_mesh_shadows_dev(terra_multi *m, const int32_t *tile_xy, uint32_t n, float *const *d_zvals, const float light_pos[3], uint8_t *const *d_smask);
/* the device-resident form of the halo arrays of terra_tiles_mesh_shadows_halo_dev: d_edge_in / d_edge_out are [n][2][130] floats in device memory
 * (d_edge_in is read where h_edge_in_present says so); what terra_multi_tiles_mesh_shadows hands from GPU to GPU */
int  terra_tiles_mesh_shadows_edges_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, const float light_pos[3], uint8_t *d_smask,
                                        const float *d_edge_in, const uint8_t *h_edge_in_present, float *d_edge_out);

/* ---- ONE grid whose row strips live on several GPUs ("erosion on one big grid").  apply_erosion works on one shared array in serial droplet order
 * (src/erosion.cpp:66-155): the droplets cannot be dealt out, the MEMORY can.  Strip i is a physical allocation on its owner's device; every rank maps all strips back
 * to back into one address range (HIP virtual memory management; a strip crosses a process boundary as a POSIX file descriptor, e.g. over a unix socket with
 * SCM_RIGHTS -- 3dworld_amd/dist.py).  Each rank then fills its own rows with terra_gen_grid_rows_minmax_dev at HBM speed, and ONE rank runs terra_apply_erosion_dev on the
 * mapped pointer: rows that live on another GPU are read and written over xGMI by the same kernels -- bit for bit the single-GPU result.
 *   rank r:  terra_dgrid_create(ctx, W, strip_bytes, r, &g); terra_dgrid_export_fd(g, &fd) -> send fd to the peers; terra_dgrid_import_fd(g, j, fd_j) for j != r;
 *            terra_dgrid_map(g, &d_grid)                 strip sizes must be multiples of terra_dgrid_granularity(ctx) (2 MiB on MI355X)
 * terra_multi_dgrid_create: the same inside one process -- strip i on context i's device, one pointer that every context's device may use. */
typedef struct terra_dgrid terra_dgrid;
size_t terra_dgrid_granularity(terra_ctx *ctx);
int  terra_dgrid_create(terra_ctx *ctx, uint32_t n_strips, const size_t *strip_bytes, uint32_t local_strip, terra_dgrid **out);
int  terra_dgrid_export_fd(terra_dgrid *g, int *fd);                    /* a new descriptor of the local strip (the caller closes it after sending) */
int  terra_dgrid_import_fd(terra_dgrid *g, uint32_t strip, int fd);     /* a peer's strip (the descriptor may be closed afterwards) */
int  terra_dgrid_map(terra_dgrid *g, void **d_base);
void terra_dgrid_destroy(terra_dgrid *g);
int  terra_multi_dgrid_create(terra_multi *m, const size_t *strip_bytes, terra_dgrid **out, void **d_base);

/* ---- plumbing for callers without a HIP runtime of their own (tests, ctypes) */
int  terra_malloc(terra_ctx *ctx, void **d_ptr, size_t bytes);
int  terra_free(terra_ctx *ctx, void *d_ptr);
int  terra_memcpy_h2d(terra_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);   /* synchronous */
int  terra_memcpy_d2h(terra_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* synchronous */
/* timing on the context's stream (HIP events): t0 = terra_timer_start; ... ; ms = terra_timer_stop (synchronises) */
int  terra_timer_start(terra_ctx *ctx);
int  terra_timer_stop(terra_ctx *ctx, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* TERRA_H */
