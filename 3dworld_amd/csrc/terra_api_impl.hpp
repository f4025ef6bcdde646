// terra_api_impl.hpp -- the extern "C" entry points of include/terra.h, written once over terra_engine<BACKEND>.
// Included by exactly one translation unit per library after it has defined `terra_backend_t`:
//   3dworld_amd/csrc/terra_hip.hip  -> libterra_hip.so  (product, HIP/gfx950)
//   tests/emul/terra_emul.cpp       -> tests/emul/libterra_emul.so (test-only host emulation of the kernel bodies)
#pragma once
#include "terra_driver.hpp"
#include "terra_png.hpp"
#include <new>
#include <mutex>
#include <atomic>

namespace terra {
static thread_local std::string g_last_error;
inline int fail(int code, char const *what) {g_last_error = what ? what : "unknown error"; return code;}
} // namespace terra

struct terra_ctx {
	terra::terra_engine<terra_backend_t> eng;
	// A context is driven by one host thread at a time (include/terra.h).  The exception is the generator handle: eval_index is const in the reference and is called
	// from its OpenMP workers (src/tiled_mesh.cpp:495, src/heightmap.cpp:139), and several handles share the context's engine (grow-only scratch, stream, the sine-table
	// upload): every terra_gen_* call that touches the engine takes this lock, so handles of one context may be used from several threads
	std::recursive_mutex eng_mtx;
};

struct terra_gen { // mesh_xy_grid_cache_t (src/mesh.h:22-45)
	terra_ctx *ctx = nullptr;
	float x0 = 0, y0 = 0, dx = 0, dy = 0;
	uint32_t nx = 0, ny = 0, flags = 0;
	bool built = false, running = false, glaciated = false;
	std::atomic<bool> collected{false}; // release-stored after cached_vals is filled: eval_index's fast path reads the grid without the lock once this is set
	float *d_vals = nullptr; size_t d_count = 0;
	std::vector<float> cached_vals;
	// eval_index(x, y, min_start_sin): the sine sum starts at max(start_eval_sin, min_start_sin) (src/mesh_gen.cpp:770); the device grid was evaluated with
	// kstart; a caller that asks for another first term gets a grid evaluated with that one (one more launch, kept per first term)
	int kstart = 0, sev = 0, gen_mode = 0;
	std::map<int, std::vector<float>> alt_vals;
	std::mutex mtx; // eval_index is const in the reference and called from its OpenMP workers (src/tiled_mesh.cpp:495, src/heightmap.cpp:139): the handle's own state (lazy read-back, alternative
	                // first terms) is serialised by this lock, the shared engine of the context by terra_ctx::eng_mtx
};

#define TERRA_TRY   try {
#define TERRA_CATCH } catch (std::invalid_argument const &e) {return terra::fail(TERRA_ERR_ARG, e.what());} \
                      catch (std::logic_error const &e) {return terra::fail(TERRA_ERR_STATE, e.what());} \
                      catch (std::bad_alloc const &) {return terra::fail(TERRA_ERR_LIMIT, "out of memory");} \
                      catch (std::exception const &e) {return terra::fail(TERRA_ERR_HIP, e.what());} \
                      return TERRA_OK;
#define TERRA_CHECK_CTX if (!ctx) return terra::fail(TERRA_ERR_ARG, "null terra_ctx");

extern "C" {

const char *terra_last_error(void) {return terra::g_last_error.c_str();}
int terra_device_count(void) {return terra_backend_t::device_count();}

int terra_create(terra_ctx **out, int device_index) {
	if (!out) return terra::fail(TERRA_ERR_ARG, "terra_create: null out pointer");
	*out = nullptr;
	int const ndev = terra_backend_t::device_count();
	if (ndev <= 0) return terra::fail(TERRA_ERR_NODEVICE, "terra_create: no usable HIP device (libterra_hip has no CPU fall-back)");
	if (device_index < 0 || device_index >= ndev) return terra::fail(TERRA_ERR_ARG, "terra_create: device index out of range");
	TERRA_TRY
		terra_ctx *c = new terra_ctx();
		try {c->eng.be.init(device_index);} catch (...) {delete c; throw;}
		*out = c;
	TERRA_CATCH
}
void terra_destroy(terra_ctx *ctx) {if (ctx) {try {ctx->eng.be.download_wait();} catch (...) {} try {ctx->eng.be.sync();} catch (...) {} delete ctx;}} // (a pending terra_download_async still reads device scratch)
int terra_set_stream(terra_ctx *ctx, void *s) {TERRA_CHECK_CTX TERRA_TRY ctx->eng.be.set_stream(s); TERRA_CATCH}
int terra_release_scratch(terra_ctx *ctx) {TERRA_CHECK_CTX TERRA_TRY ctx->eng.be.download_wait(); ctx->eng.release_scratch(); TERRA_CATCH}
int terra_synchronize(terra_ctx *ctx) {TERRA_CHECK_CTX TERRA_TRY ctx->eng.be.sync(); TERRA_CATCH}
int terra_set_option(terra_ctx *ctx, const char *key, const char *value) {
	TERRA_CHECK_CTX if (!key || !value) return terra::fail(TERRA_ERR_ARG, "terra_set_option: null key / value");
	TERRA_TRY std::lock_guard<std::recursive_mutex> lk(ctx->eng_mtx); ctx->eng.set_option(key, value); TERRA_CATCH
}
void *terra_host_alloc(size_t bytes) {void *p = terra_backend_t::host_alloc(bytes); if (!p) {terra::fail(TERRA_ERR_HIP, "terra_host_alloc: out of pinned host memory");} return p;}
void terra_host_free(void *p) {terra_backend_t::host_free(p);}
int terra_download_async(terra_ctx *ctx, const void *d_src, void *h_dst, size_t bytes) {
	TERRA_CHECK_CTX if (!d_src || !h_dst) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.be.download_async(d_src, h_dst, bytes); TERRA_CATCH
}
int terra_download_wait(terra_ctx *ctx) {TERRA_CHECK_CTX TERRA_TRY ctx->eng.be.download_wait(); TERRA_CATCH}

// ---- events: stream-level ordering between contexts
struct terra_event {void *ev = nullptr;};
int terra_event_create(terra_ctx *ctx, terra_event **out) {
	TERRA_CHECK_CTX if (!out) return terra::fail(TERRA_ERR_ARG, "null out");
	*out = nullptr;
	TERRA_TRY terra_event *e = new terra_event(); try {e->ev = ctx->eng.be.event_create();} catch (...) {delete e; throw;} *out = e; TERRA_CATCH
}
int terra_event_record(terra_ctx *ctx, terra_event *ev) {TERRA_CHECK_CTX if (!ev) return terra::fail(TERRA_ERR_ARG, "null event"); TERRA_TRY ctx->eng.be.event_record(ev->ev); TERRA_CATCH}
int terra_event_synchronize(terra_event *ev) {if (!ev) return terra::fail(TERRA_ERR_ARG, "null event"); TERRA_TRY terra_backend_t::event_synchronize(ev->ev); TERRA_CATCH}
int terra_event_wait(terra_ctx *ctx, terra_event *ev) {TERRA_CHECK_CTX if (!ev) return terra::fail(TERRA_ERR_ARG, "null event"); TERRA_TRY ctx->eng.be.event_wait(ev->ev); TERRA_CATCH}
void terra_event_destroy(terra_event *ev) {if (ev) {terra_backend_t::event_destroy(ev->ev); delete ev;}}

int terra_init_scene(terra_ctx *ctx, const terra_config *cfg) {
	TERRA_CHECK_CTX
	if (!cfg) return terra::fail(TERRA_ERR_ARG, "terra_init_scene: null config");
	TERRA_TRY ctx->eng.init_scene(*cfg); TERRA_CATCH
}
int terra_set_config(terra_ctx *ctx, const terra_config *cfg) {
	TERRA_CHECK_CTX
	if (!cfg) return terra::fail(TERRA_ERR_ARG, "terra_set_config: null config");
	TERRA_TRY ctx->eng.set_config(*cfg); TERRA_CATCH
}
int terra_get_state(terra_ctx *ctx, terra_state *out) {TERRA_CHECK_CTX if (!out) return terra::fail(TERRA_ERR_ARG, "null out"); TERRA_TRY ctx->eng.require_scene(); ctx->eng.get_state(*out); TERRA_CATCH}
int terra_set_state(terra_ctx *ctx, const terra_state *in) {TERRA_CHECK_CTX if (!in) return terra::fail(TERRA_ERR_ARG, "null state"); TERRA_TRY ctx->eng.set_state(*in); TERRA_CATCH}
int terra_set_mode(terra_ctx *ctx, int mode, int shape) {
	TERRA_CHECK_CTX
	if (mode < 0 || mode > TERRA_MGEN_DWARP_GPU || shape < 0 || shape > 2) return terra::fail(TERRA_ERR_ARG, "terra_set_mode: bad mode/shape");
	ctx->eng.mode = mode; ctx->eng.shape = shape; return TERRA_OK;
}
int terra_set_zmax_est(terra_ctx *ctx, float v) {TERRA_CHECK_CTX ctx->eng.set_zmax_est(v); ctx->eng.set_zvals(); return TERRA_OK;}
int terra_set_water_plane_z(terra_ctx *ctx, float v) {TERRA_CHECK_CTX ctx->eng.water_plane_z = v; return TERRA_OK;}
int terra_set_start_eval_sin(terra_ctx *ctx, int v) {
	TERRA_CHECK_CTX
	if (v < 0 || v > TERRA_F_TABLE_SIZE) return terra::fail(TERRA_ERR_ARG, "start_eval_sin out of range"); // assert(start_eval_sin <= F_TABLE_SIZE), src/mesh_gen.cpp:590
	ctx->eng.start_eval_sin = v; return TERRA_OK;
}
int terra_set_erode_amount(terra_ctx *ctx, float v) {TERRA_CHECK_CTX ctx->eng.erode_amount = v; return TERRA_OK;}
float terra_get_max_sea_level(terra_ctx *ctx) {return ctx ? ctx->eng.get_max_sea_level() : 0.0f;}

// ---- generator handle
int terra_gen_create(terra_ctx *ctx, terra_gen **out) {
	TERRA_CHECK_CTX
	if (!out) return terra::fail(TERRA_ERR_ARG, "null out");
	TERRA_TRY *out = new terra_gen(); (*out)->ctx = ctx; TERRA_CATCH
}
void terra_gen_destroy(terra_gen *g) {
	if (!g) return;
	try {if (g->d_vals) {std::lock_guard<std::recursive_mutex> eng_lock(g->ctx->eng_mtx); g->ctx->eng.be.sync(); g->ctx->eng.be.free(g->d_vals);}} catch (...) {}
	delete g;
}
static void terra_gen_do_collect(terra_gen *g) { // caller holds g->mtx
	if (g->collected.load(std::memory_order_acquire)) return;
	std::lock_guard<std::recursive_mutex> eng_lock(g->ctx->eng_mtx);
	g->cached_vals.resize((size_t)g->nx*g->ny);
	g->ctx->eng.be.d2h(g->cached_vals.data(), g->d_vals, g->cached_vals.size()*sizeof(float)); // blocks on the stream, like read_float_vals (src/shaders.cpp:1196-1235)
	g->running = false; g->collected.store(true, std::memory_order_release);
}
int terra_gen_build_arrays(terra_gen *g, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin) {
	if (!g) return terra::fail(TERRA_ERR_ARG, "null terra_gen");
	try {
		std::lock_guard<std::mutex> lock(g->mtx);
		std::lock_guard<std::recursive_mutex> eng_lock(g->ctx->eng_mtx);
		bool const no_wait = (flags & TERRA_GEN_NO_WAIT) != 0;
		uint32_t const key = flags & (TERRA_GEN_GLACIATE | TERRA_GEN_FORCE_SINE | TERRA_GEN_FUSED | TERRA_GEN_FAST);
		int const sev = g->ctx->eng.start_eval_sin, kstart = terra::imax(sev, min_start_sin);
		bool const same = g->built && g->x0 == x0 && g->y0 == y0 && g->dx == dx && g->dy == dy && g->nx == nx && g->ny == ny && (g->flags & (TERRA_GEN_GLACIATE | TERRA_GEN_FORCE_SINE | TERRA_GEN_FUSED | TERRA_GEN_FAST)) == key && g->kstart == kstart;
		bool const was_running = g->running && same;
		if (!was_running) { // launch the job (run_gpu_simplex, src/mesh_gen.cpp:652-681)
			size_t const count = (size_t)nx*ny;
			if (count == 0) return terra::fail(TERRA_ERR_ARG, "build_arrays: nx, ny must be > 0");
			g->collected.store(false, std::memory_order_release); // before the geometry changes: eval_index's lock-free path is only valid while this is set
			if (count > g->d_count) {if (g->d_vals) {g->ctx->eng.be.sync(); g->ctx->eng.be.free(g->d_vals);} g->d_vals = (float *)g->ctx->eng.be.alloc(count*sizeof(float)); g->d_count = count;}
			g->x0 = x0; g->y0 = y0; g->dx = dx; g->dy = dy; g->nx = nx; g->ny = ny; g->flags = flags;
			g->ctx->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, g->d_vals);
			g->built = true; g->running = true; g->collected.store(false, std::memory_order_release); g->glaciated = (flags & TERRA_GEN_GLACIATE) != 0;
			g->kstart = kstart; g->sev = sev; g->gen_mode = (flags & TERRA_GEN_FORCE_SINE) ? (int)terra::MGEN_SINE : g->ctx->eng.mode;
			g->cached_vals.clear(); g->alt_vals.clear();
		}
		if (no_wait && !was_running) return 0; // just started, results not yet available
		terra_gen_do_collect(g);
		return 1;
	}
	catch (std::invalid_argument const &e) {return terra::fail(TERRA_ERR_ARG, e.what());}
	catch (std::logic_error const &e) {return terra::fail(TERRA_ERR_STATE, e.what());}
	catch (std::exception const &e) {return terra::fail(TERRA_ERR_HIP, e.what());}
}
int terra_gen_enable_glaciate(terra_gen *g) {
	if (!g) return terra::fail(TERRA_ERR_ARG, "null terra_gen");
	TERRA_TRY
		std::lock_guard<std::mutex> lock(g->mtx); // the handle's state is read under its lock: a build_arrays on another thread may be rewriting it
		if (!g->built) throw std::logic_error("enable_glaciate: build_arrays() must have been called first"); // assert(cur_nx > 0 && cur_ny > 0), src/mesh_gen.cpp:644
		if (g->glaciated) return TERRA_OK;
		std::lock_guard<std::recursive_mutex> eng_lock(g->ctx->eng_mtx);
		// not fused at build time: re-evaluate with the glaciate epilogue (pure per-cell function, identical values)
		g->ctx->eng.gen_grid_dev(g->x0, g->y0, g->dx, g->dy, g->nx, g->ny, g->flags | TERRA_GEN_GLACIATE, g->kstart, g->d_vals);
		g->flags |= TERRA_GEN_GLACIATE; g->glaciated = true; g->running = true; g->collected.store(false, std::memory_order_release); g->alt_vals.clear();
	TERRA_CATCH
}
int terra_gen_is_running(terra_gen *g) {return (g && g->running) ? 1 : 0;}
int terra_gen_collect(terra_gen *g, float *host_out) {
	if (!g || !host_out) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY
		std::lock_guard<std::mutex> lock(g->mtx);
		if (!g->built) throw std::logic_error("collect: nothing was built");
		terra_gen_do_collect(g); memcpy(host_out, g->cached_vals.data(), g->cached_vals.size()*sizeof(float));
	TERRA_CATCH
}
float terra_gen_eval_index(terra_gen *g, uint32_t x, uint32_t y, int min_start_sin, int use_cache) {
	if (!g) {terra::fail(TERRA_ERR_ARG, "eval_index: null terra_gen"); return 0.0f;}
	try {
		// fast path, lock-free: the reference's OpenMP workers all call eval_index on one collected grid (src/tiled_mesh.cpp:495, src/heightmap.cpp:139).  Once `collected`
		// is set (release) the handle's geometry and cached_vals do not change until the next build_arrays, which the caller must not overlap with eval_index on the
		// same handle (the reference's build_arrays is main-thread only, src/mesh.h:40) -- concurrent builds of OTHER handles are fine
		if (g->collected.load(std::memory_order_acquire)) {
			int const want0 = (g->gen_mode == (int)terra::MGEN_SINE) ? ((use_cache && (g->flags & TERRA_GEN_CACHE_VALUES)) ? g->sev : terra::imax(g->sev, min_start_sin)) : g->kstart;
			if (want0 == g->kstart) {
				if (x >= g->nx || y >= g->ny) {terra::fail(TERRA_ERR_ARG, "eval_index: out of range"); return 0.0f;}
				return g->cached_vals[(size_t)y*g->nx + x];
			}
		}
		std::lock_guard<std::mutex> lock(g->mtx);
		if (!g->built || x >= g->nx || y >= g->ny) {terra::fail(TERRA_ERR_ARG, "eval_index: out of range"); return 0.0f;} // assert(x < cur_nx && y < cur_ny), src/mesh_gen.cpp:756
		// which first sine term the reference would use (src/mesh_gen.cpp:759-770): cached values (cache_values at build time, filled with min_start_sin = 0)
		// win when use_cache is set; the fBm modes have no sine terms
		int want = g->kstart;
		if (g->gen_mode == (int)terra::MGEN_SINE) {want = (use_cache && (g->flags & TERRA_GEN_CACHE_VALUES)) ? g->sev : terra::imax(g->sev, min_start_sin);}
		if (want == g->kstart) {terra_gen_do_collect(g); return g->cached_vals[(size_t)y*g->nx + x];}
		std::vector<float> &alt = g->alt_vals[want];
		if (alt.empty()) { // same grid from another first term: one more launch, kept for the following calls
			std::lock_guard<std::recursive_mutex> eng_lock(g->ctx->eng_mtx);
			terra_backend_t &be = g->ctx->eng.be;
			size_t const count = (size_t)g->nx*g->ny;
			float *d = (float *)be.alloc(count*sizeof(float));
			std::vector<float> tmp(count);
			try {
				g->ctx->eng.gen_grid_dev(g->x0, g->y0, g->dx, g->dy, g->nx, g->ny, g->flags & (TERRA_GEN_GLACIATE | TERRA_GEN_FORCE_SINE | TERRA_GEN_FUSED | TERRA_GEN_FAST), want, d); // first term = max(start_eval_sin, want)
				be.d2h(tmp.data(), d, count*sizeof(float));
			} catch (...) {be.free(d); g->alt_vals.erase(want); throw;}
			be.free(d);
			alt.swap(tmp);
		}
		return alt[(size_t)y*g->nx + x];
	}
	catch (std::exception const &e) {terra::fail(TERRA_ERR_HIP, e.what()); return 0.0f;}
}
const float *terra_gen_device_values(terra_gen *g) {return (g && g->built) ? g->d_vals : nullptr;}

int terra_gen_grid_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *d_out) {
	TERRA_CHECK_CTX if (!d_out) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY ctx->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, d_out); TERRA_CATCH
}
int terra_gen_grid_minmax_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *d_out, float *h_min, float *h_max) {
	TERRA_CHECK_CTX if (!d_out) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY float mm[2]; ctx->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, d_out, mm); if (h_min) *h_min = mm[0]; if (h_max) *h_max = mm[1]; TERRA_CATCH
}
int terra_gen_grid_minmax_async_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *d_out, float *d_minmax) {
	TERRA_CHECK_CTX if (!d_out || !d_minmax) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY ctx->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, d_out, nullptr, 0, 0xFFFFFFFFu, d_minmax); TERRA_CATCH
}
int terra_gen_grid_rows_minmax_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, uint32_t row0, uint32_t nrows, float *d_out, float *h_min, float *h_max) {
	TERRA_CHECK_CTX if (!d_out) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY
		float mm[2];
		ctx->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, d_out, (h_min || h_max) ? mm : nullptr, row0, nrows);
		if (h_min) *h_min = mm[0];
		if (h_max) *h_max = mm[1];
	TERRA_CATCH
}
int terra_gen_grid_rows_minmax_async_dev(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, uint32_t row0, uint32_t nrows, float *d_out, float *d_minmax) {
	TERRA_CHECK_CTX if (!d_out || !d_minmax) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY ctx->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, d_out, nullptr, row0, nrows, d_minmax); TERRA_CATCH
}
int terra_gen_grid(terra_ctx *ctx, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *h_out) {
	TERRA_CHECK_CTX if (!h_out) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY
		size_t const bytes = (size_t)nx*ny*sizeof(float);
		if (bytes == 0) throw std::invalid_argument("build_arrays: nx, ny must be > 0");
		float *d = ctx->eng.host_grid_scratch(bytes);
		ctx->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, d); ctx->eng.be.d2h(h_out, d, bytes);
	TERRA_CATCH
}

int terra_eval_points(terra_ctx *ctx, const float *xy, uint32_t n, uint32_t kind, float xy_scale, int no_xyoff, int xoff2, int yoff2, float *out) {
	TERRA_CHECK_CTX if (n && (!xy || !out)) return terra::fail(TERRA_ERR_ARG, "terra_eval_points: null pointer");
	TERRA_TRY std::lock_guard<std::recursive_mutex> lk(ctx->eng_mtx); ctx->eng.eval_points(xy, n, kind, xy_scale, no_xyoff, xoff2, yoff2, out); TERRA_CATCH
}
int terra_eval_points_dev(terra_ctx *ctx, const float *d_xy, uint32_t n, uint32_t kind, float xy_scale, int no_xyoff, int xoff2, int yoff2, float *d_out) {
	TERRA_CHECK_CTX if (n && (!d_xy || !d_out)) return terra::fail(TERRA_ERR_ARG, "terra_eval_points_dev: null pointer");
	TERRA_TRY std::lock_guard<std::recursive_mutex> lk(ctx->eng_mtx); ctx->eng.eval_points_dev(d_xy, n, kind, xy_scale, no_xyoff, xoff2, yoff2, d_out); TERRA_CATCH
}
int terra_eval_mesh_sin_terms(terra_ctx *ctx, float xv, float yv, float *out) {
	TERRA_CHECK_CTX if (!out) return terra::fail(TERRA_ERR_ARG, "null out");
	TERRA_TRY *out = ctx->eng.eval_mesh_sin_terms(xv, yv); TERRA_CATCH
}
int terra_glaciate_mesh_dev(terra_ctx *ctx, float *d_mesh, uint32_t nx, uint32_t ny, int xoff2, int yoff2, float *h_zbottom_ztop) {
	TERRA_CHECK_CTX if (!d_mesh) return terra::fail(TERRA_ERR_ARG, "null mesh");
	TERRA_TRY ctx->eng.glaciate_mesh_dev(d_mesh, nx, ny, xoff2, yoff2, h_zbottom_ztop); TERRA_CATCH
}

// ---- erosion
int terra_apply_erosion_dev(terra_ctx *ctx, float *d, int xs, int ys, float min_zval, uint32_t iters, uint32_t flags) {
	TERRA_CHECK_CTX if (!d) return terra::fail(TERRA_ERR_ARG, "null heightmap");
	TERRA_TRY ctx->eng.apply_erosion_dev(d, xs, ys, min_zval, iters, flags); TERRA_CATCH
}
int terra_apply_erosion_devmin_dev(terra_ctx *ctx, float *d, int xs, int ys, const float *d_min_zval, uint32_t iters, uint32_t flags) {
	TERRA_CHECK_CTX if (!d || !d_min_zval) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.apply_erosion_dev(d, xs, ys, 0.0f, iters, flags, d_min_zval); TERRA_CATCH
}
size_t terra_erosion_shard_arena_bytes(terra_ctx *ctx, uint32_t iters) {
	if (!ctx) return 0;
	return ctx->eng.sparse_arena_bytes(iters);
}
int terra_erosion_shard_trace_dev(terra_ctx *ctx, float *d, int xs, int ys, uint32_t iters, uint32_t row0, uint32_t nrows, void *d_arena) {
	TERRA_CHECK_CTX if (!d || !d_arena) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (ys <= 0 || row0 > (uint32_t)ys || nrows > (uint32_t)ys - row0) return terra::fail(TERRA_ERR_ARG, "terra_erosion_shard_trace_dev: rows outside the grid");
	TERRA_TRY
		terra::sparse_shard_t sh{};
		sh.phase = 1; sh.row0 = row0; sh.row1 = row0 + nrows; sh.arena = (uint8_t *)d_arena;
		ctx->eng.apply_erosion_dev(d, xs, ys, 0.0f, iters, 0, nullptr, &sh);
	TERRA_CATCH
}
int terra_erosion_shard_finish_dev(terra_ctx *ctx, float *d, int xs, int ys, const float *d_min_zval, uint32_t iters, uint32_t flags, uint32_t world, uint32_t self, const uint32_t *row_end, void *d_arena_self, size_t arena_stride) {
	TERRA_CHECK_CTX if (!d || !d_min_zval || !row_end || !d_arena_self) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (world == 0 || world > terra::SPARSE_SHARD_MAX_WORLD || self >= world) return terra::fail(TERRA_ERR_ARG, "terra_erosion_shard_finish_dev: world must be 1 .. 16 and self below it");
	if (world > 1 && arena_stride < ctx->eng.sparse_arena_bytes(iters)) return terra::fail(TERRA_ERR_ARG, "terra_erosion_shard_finish_dev: arena_stride is smaller than an arena (terra_erosion_shard_arena_bytes)");
	for (uint32_t r = 0; r < world; ++r) {
		if ((r && row_end[r] < row_end[r - 1]) || (r + 1 == world && (ys <= 0 || row_end[r] != (uint32_t)ys))) return terra::fail(TERRA_ERR_ARG, "terra_erosion_shard_finish_dev: row_end must be non-decreasing and end at ysize");
	}
	TERRA_TRY
		terra::sparse_shard_t sh{};
		sh.phase = 2; sh.arena = (uint8_t *)d_arena_self; sh.world = world; sh.self = self; sh.stride = (long long)arena_stride;
		for (uint32_t r = 0; r < world; ++r) {sh.rows.end[r] = row_end[r];}
		ctx->eng.apply_erosion_dev(d, xs, ys, 0.0f, iters, flags, d_min_zval, &sh);
	TERRA_CATCH
}
int terra_apply_erosion(terra_ctx *ctx, float *h, int xs, int ys, float min_zval, uint32_t iters) {
	TERRA_CHECK_CTX if (!h) return terra::fail(TERRA_ERR_ARG, "null heightmap");
	TERRA_TRY
		if (iters == 0 || ctx->eng.erode_amount <= 0.0f) return TERRA_OK;
		if (xs <= 0 || ys <= 0) throw std::invalid_argument("apply_erosion: bad grid size");
		size_t const bytes = (size_t)xs*ys*sizeof(float);
		float *d = ctx->eng.host_grid_scratch(bytes); // grow-only (a 1 GiB hipMalloc + hipFree per call cost more than the erosion)
		ctx->eng.be.h2d(d, h, bytes); ctx->eng.apply_erosion_dev(d, xs, ys, min_zval, iters, 0); ctx->eng.be.d2h(h, d, bytes);
	TERRA_CATCH
}
int terra_set_erosion_tuning(terra_ctx *ctx, uint32_t window, uint32_t cap_log2, uint32_t maxb) {
	TERRA_CHECK_CTX
	if ((cap_log2 && (cap_log2 < 12 || cap_log2 > 20)) || (maxb && (maxb < 8 || maxb > 65536)) || (window > (1u << 20) && window != 0xFFFFFFFFu)) return terra::fail(TERRA_ERR_ARG, "terra_set_erosion_tuning: value out of range");
	if (window) ctx->eng.spec_cfg.window = (window == 0xFFFFFFFFu) ? 0 : window; // 0xFFFFFFFF: back to the automatic choice
	(void)cap_log2; // versions are stored as one 64-float page per footprint block since round 2: there is no hashed log to size any more (accepted for compatibility)
	if (maxb) ctx->eng.spec_cfg.maxb = maxb;
	return TERRA_OK;
}
int terra_set_erosion_slice_steps(terra_ctx *ctx, uint32_t steps) {
	TERRA_CHECK_CTX
	if (steps == 0) return terra::fail(TERRA_ERR_ARG, "terra_set_erosion_slice_steps: steps must be > 0");
	ctx->eng.spec_cfg.slice_steps = steps;
	return TERRA_OK;
}
int terra_get_erosion_report(terra_ctx *ctx, terra_erosion_report *out) {TERRA_CHECK_CTX if (!out) return terra::fail(TERRA_ERR_ARG, "null out"); *out = ctx->eng.report; return TERRA_OK;}

// ---- whole heightmap (heightmap_t::proc_gen, src/heightmap.cpp:130-151)
int terra_minmax_dev(terra_ctx *ctx, const float *d, size_t n, float *h_min, float *h_max) {
	TERRA_CHECK_CTX if (!d || n == 0) return terra::fail(TERRA_ERR_ARG, "empty input");
	TERRA_TRY float mn, mx; ctx->eng.minmax_dev(d, n, mn, mx); if (h_min) *h_min = mn; if (h_max) *h_max = mx; TERRA_CATCH
}
int terra_quantize16_dev(terra_ctx *ctx, const float *d, size_t n, float min_z, float dz, uint8_t *d_pix) {
	TERRA_CHECK_CTX if (!d || !d_pix) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (!(dz > 0.0f)) return terra::fail(TERRA_ERR_ARG, "quantize16: dz must be > 0"); // assert(dz > 0.0), src/mesh_gen.cpp:126
	TERRA_TRY ctx->eng.quantize16_dev(d, n, min_z, dz, d_pix); TERRA_CATCH
}
int terra_heightmap_proc_gen_dev(terra_ctx *ctx, uint32_t width, uint32_t height, uint32_t erosion_iters, float *d_vals, uint8_t *d_pix, float *h_range) {
	TERRA_CHECK_CTX if (!d_vals) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY
		auto &e = ctx->eng;
		size_t const n = (size_t)width*height;
		float mm[2];
		bool const erode = erosion_iters > 0 && e.erode_amount > 0.0f;
		e.gen_grid_dev((float)(-0.5*(double)width), (float)(-0.5*(double)height), e.DX_VAL, e.DY_VAL, width, height, TERRA_GEN_GLACIATE, 0, d_vals, erode ? mm : nullptr); // src/heightmap.cpp:135-143
		if (erode) {e.apply_erosion_dev(d_vals, (int)width, (int)height, mm[0], erosion_iters, TERRA_ERODE_MINZ_IS_MIN);} // run_erosion (src/heightmap.cpp:153-187): min_zval = min(vals)
		if (d_pix || h_range) {
			float mn, mx; e.minmax_dev(d_vals, n, mn, mx); // get_heightmap_z_range
			float const dz = terra::max_std(1.0E-12f, (mx - mn)); // max(TOLERANCE, ...), src/heightmap.cpp:148
			if (h_range) {h_range[0] = mn; h_range[1] = dz;}
			if (d_pix) {e.quantize16_dev(d_vals, n, mn, dz, d_pix);}
		}
	TERRA_CATCH
}

// host form: what heightmap_t::proc_gen leaves in the texture's pixel buffer (src/heightmap.cpp:130-151) -- 2 bytes per cell cross the host link, nothing else
int terra_heightmap_proc_gen(terra_ctx *ctx, uint32_t width, uint32_t height, uint32_t erosion_iters, uint8_t *h_pix, float *h_range) {
	TERRA_CHECK_CTX if (!h_pix) return terra::fail(TERRA_ERR_ARG, "null output");
	if (width == 0 || height == 0) return terra::fail(TERRA_ERR_ARG, "terra_heightmap_proc_gen: empty map");
	TERRA_TRY
		size_t const n = (size_t)width*height;
		uint8_t *d = (uint8_t *)ctx->eng.host_grid_scratch(n*6 + 512); // floats, then the 16-bit pixels (16-byte aligned)
		uint8_t *d_pix = d + ((n*4 + 255) & ~(size_t)255);
		int const rc = terra_heightmap_proc_gen_dev(ctx, width, height, erosion_iters, (float *)d, d_pix, h_range);
		if (rc != TERRA_OK) return rc;
		ctx->eng.be.d2h(h_pix, d_pix, n*2);
	TERRA_CATCH
}

// ---- the loaded-heightmap path (rest of row a12): heightmap_t::to_floats / from_floats / postprocess_height (src/heightmap.cpp:117-128,191-215)
int terra_set_mesh_file_scale(terra_ctx *ctx, float mesh_file_scale, float mesh_file_tz) {
	TERRA_CHECK_CTX
	if (!(mesh_file_scale != 0.0f) || mesh_file_scale != mesh_file_scale || mesh_file_tz != mesh_file_tz) return terra::fail(TERRA_ERR_ARG, "terra_set_mesh_file_scale: scale must be a non-zero number");
	ctx->eng.mesh_file_scale = mesh_file_scale; ctx->eng.mesh_file_tz = mesh_file_tz;
	return TERRA_OK;
}
int terra_get_mesh_file_scale(terra_ctx *ctx, float *mesh_file_scale, float *mesh_file_tz) {
	TERRA_CHECK_CTX
	if (mesh_file_scale) *mesh_file_scale = ctx->eng.mesh_file_scale;
	if (mesh_file_tz) *mesh_file_tz = ctx->eng.mesh_file_tz;
	return TERRA_OK;
}
static int terra_check_image(uint32_t width, uint32_t height, int ncolors) {
	if (width == 0 || height == 0 || (uint64_t)width*height >= (1ull << 31)) return terra::fail(TERRA_ERR_ARG, "heightmap image: bad size");
	if (ncolors != 1 && ncolors != 2) return terra::fail(TERRA_ERR_ARG, "heightmap image: one or two byte grayscale only");
	return TERRA_OK;
}
int terra_heightmap_to_floats_dev(terra_ctx *ctx, const uint8_t *d_pixels, uint32_t width, uint32_t height, int ncolors, float *d_vals) {
	TERRA_CHECK_CTX if (!d_pixels || !d_vals) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (int const rc = terra_check_image(width, height, ncolors)) return rc;
	TERRA_TRY ctx->eng.heightmap_to_floats_dev(d_pixels, (size_t)width*height, ncolors, d_vals); TERRA_CATCH
}
int terra_heightmap_from_floats_dev(terra_ctx *ctx, const float *d_vals, uint32_t width, uint32_t height, int ncolors, uint8_t *d_pixels, uint32_t *h_out_of_range) {
	TERRA_CHECK_CTX if (!d_pixels || !d_vals) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (int const rc = terra_check_image(width, height, ncolors)) return rc;
	TERRA_TRY
		uint32_t const bad = ctx->eng.heightmap_from_floats_dev(d_vals, (size_t)width*height, ncolors, d_pixels);
		if (h_out_of_range) {*h_out_of_range = bad;}
		else if (bad) throw std::logic_error("heightmap from_floats: values outside [0, 256) pixel units (the reference asserts, src/heightmap.cpp:210)");
	TERRA_CATCH
}
int terra_heightmap_postprocess_dev(terra_ctx *ctx, uint8_t *d_pixels, uint32_t width, uint32_t height, int ncolors, uint32_t erosion_iters_tt, float *d_vals, uint32_t *h_out_of_range) {
	TERRA_CHECK_CTX if (!d_pixels) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (int const rc = terra_check_image(width, height, ncolors)) return rc;
	TERRA_TRY
		auto &be = ctx->eng.be;
		float *vals = d_vals;
		if (!vals && erosion_iters_tt) {vals = (float *)be.alloc((size_t)width*height*sizeof(float));}
		uint32_t bad = 0;
		try {bad = ctx->eng.heightmap_postprocess_dev(d_pixels, width, height, ncolors, erosion_iters_tt, vals); if (!d_vals) {be.sync();}} catch (...) {if (!d_vals && vals) be.free(vals); throw;}
		if (!d_vals && vals) {be.free(vals);}
		if (h_out_of_range) {*h_out_of_range = bad;}
		else if (bad) throw std::logic_error("heightmap postprocess_height: eroded values outside [0, 256) pixel units (the reference asserts, src/heightmap.cpp:210)");
	TERRA_CATCH
}

// ---- heightmap files (host): 8- / 16-bit grayscale PNG with the reference's row order and byte order (terra_png.hpp)
int terra_heightmap_write_png(const char *path, const uint8_t *h_pixels, uint32_t width, uint32_t height, int ncolors) {
	if (!path || !h_pixels) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY terra::png_write_gray(path, h_pixels, width, height, ncolors); TERRA_CATCH
}
int terra_heightmap_read_png(const char *path, int allow_two_byte_grayscale, uint32_t *width, uint32_t *height, int *ncolors, uint8_t *h_pixels, size_t capacity) {
	if (!path || !width || !height || !ncolors) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY
		std::vector<uint8_t> const px = terra::png_read_gray(path, *width, *height, *ncolors, allow_two_byte_grayscale != 0);
		if (h_pixels) {
			if (capacity < px.size()) return terra::fail(TERRA_ERR_ARG, "terra_heightmap_read_png: buffer too small");
			memcpy(h_pixels, px.data(), px.size());
		}
	TERRA_CATCH
}

// ---- the ground mesh's text file (host): read_mesh / write_mesh, src/mesh_gen.cpp:895-965
int terra_read_mesh(terra_ctx *ctx, const char *filename, float zmm, float *h_mesh, uint32_t nx, uint32_t ny, float *h_zbottom_ztop) {
	TERRA_CHECK_CTX
	if (!filename || !h_mesh || nx == 0 || ny == 0) return terra::fail(TERRA_ERR_ARG, "terra_read_mesh: null or empty argument");
	FILE *fp = fopen(filename, "r");
	if (!fp) return terra::fail(TERRA_ERR_ARG, (std::string("terra_read_mesh: cannot open ") + filename).c_str());
	int xsize = 0, ysize = 0;
	if (fscanf(fp, "%i%i", &xsize, &ysize) != 2) {fclose(fp); return terra::fail(TERRA_ERR_ARG, "terra_read_mesh: error reading the size header");}
	if (xsize != (int)nx || ysize != (int)ny) {fclose(fp); return terra::fail(TERRA_ERR_ARG, "terra_read_mesh: the mesh size in the file is not the scene's");}
	float const fscale = ctx->eng.mesh_file_scale, ftz = ctx->eng.mesh_file_tz;
	for (size_t i = 0; i < (size_t)nx*ny; ++i) {
		float height;
		if (fscanf(fp, "%f", &height) != 1) {fclose(fp); return terra::fail(TERRA_ERR_ARG, "terra_read_mesh: error reading mesh heights");}
		h_mesh[i] = fscale*height + ftz;
	}
	fclose(fp);
	float mn = h_mesh[0], mx = h_mesh[0]; // matrix_min_max (src/mesh_gen.cpp:84-94): std::min / std::max, so a NaN never replaces a number
	for (size_t i = 0; i < (size_t)nx*ny; ++i) {float const v = h_mesh[i]; mn = (v < mn) ? v : mn; mx = (mx < v) ? v : mx;}
	if (h_zbottom_ztop) {h_zbottom_ztop[0] = mn; h_zbottom_ztop[1] = mx;}
	float const neg = -mn;
	ctx->eng.set_zmax_est((zmm != 0.0f) ? zmm : ((neg < mx) ? mx : neg)); // max(-zmin, zmax)
	ctx->eng.set_zvals();
	return TERRA_OK;
}
int terra_write_mesh(const char *filename, const float *h_mesh, uint32_t nx, uint32_t ny) {
	if (!filename || !h_mesh) return terra::fail(TERRA_ERR_ARG, "terra_write_mesh: null argument");
	FILE *fp = fopen(filename, "w");
	if (!fp) return terra::fail(TERRA_ERR_ARG, (std::string("terra_write_mesh: cannot open ") + filename).c_str());
	bool ok = fprintf(fp, "%i %i\n", (int)nx, (int)ny) > 0;
	for (uint32_t i = 0; i < ny && ok; ++i) {
		for (uint32_t j = 0; j < nx && ok; ++j) {ok = fprintf(fp, "%f ", (double)h_mesh[(size_t)i*nx + j]) > 0;}
		ok = ok && fprintf(fp, "\n") > 0;
	}
	ok = (fclose(fp) == 0) && ok;
	return ok ? TERRA_OK : terra::fail(TERRA_ERR_ARG, "terra_write_mesh: write error");
}

// ---- tiles
int terra_hmap_set_dev(terra_ctx *ctx, const uint8_t *d_pixels, int width, int height, int ncolors) {
	TERRA_CHECK_CTX
	if (d_pixels && (width <= 0 || height <= 0 || (ncolors != 1 && ncolors != 2) || (int64_t)width*height >= (1ll << 31))) return terra::fail(TERRA_ERR_ARG, "terra_hmap_set_dev: bad image shape");
	if (d_pixels && ncolors == 2 && ((uintptr_t)d_pixels & 1u)) return terra::fail(TERRA_ERR_ARG, "terra_hmap_set_dev: a 16-bit image must be 2-byte aligned"); // the edit kernels update a texel inside its aligned 32-bit word
	ctx->eng.hmap_pix = d_pixels; ctx->eng.hmap_w = d_pixels ? width : 0; ctx->eng.hmap_h = d_pixels ? height : 0; ctx->eng.hmap_nc = d_pixels ? ncolors : 0;
	return TERRA_OK;
}
int terra_set_mesh_height_scales_for_zval_range(terra_ctx *ctx, float min_z, float dz) {
	TERRA_CHECK_CTX
	TERRA_TRY ctx->eng.set_mesh_height_scales_for_zval_range(min_z, dz); TERRA_CATCH
}
int terra_set_tiled_mesh_ao(terra_ctx *ctx, int enable) {TERRA_CHECK_CTX ctx->eng.tiled_mesh_ao = (enable != 0); return TERRA_OK;}
// ---- height edits, the .mod file and the map exporter (rest of row f4)
static_assert(sizeof(terra_hmap_brush) == sizeof(terra::hmap_brush_pod_t) && sizeof(terra_hmap_mod) == sizeof(terra::hmap_mod_pod_t), "mod record layouts");
int terra_hmap_apply_brushes_dev(terra_ctx *ctx, const terra_hmap_brush *brushes, uint32_t n, int step_sz, uint32_t num_steps) {
	TERRA_CHECK_CTX if (n && !brushes) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.hmap_apply_brushes_dev((terra::hmap_brush_pod_t const *)brushes, n, step_sz, num_steps); ctx->eng.be.sync(); TERRA_CATCH
}
int terra_hmap_apply_mods_dev(terra_ctx *ctx, const terra_hmap_mod *mods, uint32_t n) {
	TERRA_CHECK_CTX if (n && !mods) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.hmap_apply_mods_dev((terra::hmap_mod_pod_t const *)mods, n); TERRA_CATCH
}
int terra_hmap_read_and_apply_mod_dev(terra_ctx *ctx, const char *path) {
	TERRA_CHECK_CTX if (!path) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.hmap_read_and_apply_mod_dev(path); ctx->eng.be.sync(); TERRA_CATCH
}
int terra_hmap_write_mod(const char *path, const terra_hmap_mod *mods, uint32_t n, const terra_hmap_brush *brushes, uint32_t n_brushes) {
	if (!path || (n && !mods) || (n_brushes && !brushes)) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY terra::write_mod_file(path, (terra::hmap_mod_pod_t const *)mods, n, (terra::hmap_brush_pod_t const *)brushes, n_brushes); TERRA_CATCH
}
int terra_hmap_read_mod(const char *path, terra_hmap_mod *mods, uint32_t mods_capacity, uint32_t *n_mods, terra_hmap_brush *brushes, uint32_t brushes_capacity, uint32_t *n_brushes) {
	if (!path || !n_mods || !n_brushes) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY
		std::vector<terra::hmap_mod_pod_t> m; std::vector<terra::hmap_brush_pod_t> b;
		terra::read_mod_file(path, m, b);
		*n_mods = (uint32_t)m.size(); *n_brushes = (uint32_t)b.size();
		if (mods) {if (mods_capacity < m.size()) return terra::fail(TERRA_ERR_ARG, "terra_hmap_read_mod: mods buffer too small"); if (!m.empty()) memcpy(mods, m.data(), m.size()*sizeof(terra_hmap_mod));}
		if (brushes) {if (brushes_capacity < b.size()) return terra::fail(TERRA_ERR_ARG, "terra_hmap_read_mod: brushes buffer too small"); if (!b.empty()) memcpy(brushes, b.data(), b.size()*sizeof(terra_hmap_brush));}
	TERRA_CATCH
}
int terra_export_heightmap_dev(terra_ctx *ctx, float xstart, float ystart, uint32_t width, uint32_t height, float *d_vals, uint8_t *d_pixels16, float *h_min_z_dz) {
	TERRA_CHECK_CTX if (!d_vals) return terra::fail(TERRA_ERR_ARG, "null output");
	TERRA_TRY ctx->eng.export_heightmap_dev(xstart, ystart, width, height, d_vals, d_pixels16, h_min_z_dz); ctx->eng.be.sync(); TERRA_CATCH
}
int terra_write_map_mode_heightmap_image(terra_ctx *ctx, const char *path, float xstart, float ystart, uint32_t width, uint32_t height) {
	TERRA_CHECK_CTX if (!path) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY
		auto &be = ctx->eng.be;
		size_t const n = (size_t)width*height;
		if (n == 0) return terra::fail(TERRA_ERR_ARG, "empty image");
		uint8_t *d = (uint8_t *)be.alloc(n*6);
		try {
			ctx->eng.export_heightmap_dev(xstart, ystart, width, height, (float *)d, d + n*4, nullptr);
			std::vector<uint8_t> px(n*2);
			be.d2h(px.data(), d + n*4, n*2);
			terra::png_write_gray(path, px.data(), width, height, 2);
		} catch (...) {be.free(d); throw;}
		be.free(d);
	TERRA_CATCH
}
int terra_set_landscape(terra_ctx *ctx, const terra_landscape *params) {
	TERRA_CHECK_CTX if (!params) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.set_landscape(*params); TERRA_CATCH
}
int terra_get_landscape(terra_ctx *ctx, terra_landscape *out) {
	TERRA_CHECK_CTX if (!out) return terra::fail(TERRA_ERR_ARG, "null argument");
	*out = ctx->eng.ls; return TERRA_OK;
}
int terra_tiles_terrain_params(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, float *h_params) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !h_params)) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.tiles_terrain_params(tile_xy, n, h_params); TERRA_CATCH
}
int terra_tiles_create_weights_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, uint8_t *d_weights, terra_grass_block *d_grass_blocks, uint8_t *d_has_any_grass) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !d_zvals || !d_weights)) return terra::fail(TERRA_ERR_ARG, "null argument");
	static_assert(sizeof(terra_grass_block) == sizeof(terra::grass_block_pod_t), "grass block layout");
	TERRA_TRY ctx->eng.tiles_create_weights_dev(tile_xy, n, d_zvals, d_weights, (terra::grass_block_pod_t *)d_grass_blocks, d_has_any_grass); TERRA_CATCH
}
int terra_tiles_create_weights(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, uint8_t *h_weights, terra_grass_block *h_grass_blocks, uint8_t *h_has_any_grass) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !h_zvals || !h_weights)) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (n == 0) return TERRA_OK;
	TERRA_TRY
		auto &be = ctx->eng.be;
		size_t const zb = (size_t)n*130*130*4, wb = (size_t)n*129*129*4, gb = (size_t)n*32*32*sizeof(terra_grass_block), hb = ((size_t)n + 3) & ~(size_t)3;
		uint8_t *d = (uint8_t *)be.alloc(zb + wb + gb + hb);
		try {
			be.h2d(d, h_zvals, zb);
			ctx->eng.tiles_create_weights_dev(tile_xy, n, (float const *)d, d + zb, h_grass_blocks ? (terra::grass_block_pod_t *)(d + zb + wb) : nullptr, h_has_any_grass ? d + zb + wb + gb : nullptr);
			be.d2h(h_weights, d + zb, wb);
			if (h_grass_blocks) {be.d2h(h_grass_blocks, d + zb + wb, gb);}
			if (h_has_any_grass) {be.d2h(h_has_any_grass, d + zb + wb + gb, n);}
		} catch (...) {be.free(d); throw;}
		be.free(d);
	TERRA_CATCH
}
int terra_tiles_ao_lighting_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, uint8_t *d_ao) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !d_zvals || !d_ao)) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.tiles_ao_lighting_dev(tile_xy, n, d_zvals, d_ao); TERRA_CATCH
}
int terra_tiles_ao_lighting(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, uint8_t *h_ao) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !h_zvals || !h_ao)) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (n == 0) return TERRA_OK;
	TERRA_TRY
		auto &be = ctx->eng.be;
		size_t const zb = (size_t)n*130*130*4, ab = (size_t)n*129*129;
		uint8_t *d = (uint8_t *)be.alloc(zb + ab);
		try {
			be.h2d(d, h_zvals, zb);
			ctx->eng.tiles_ao_lighting_dev(tile_xy, n, (float const *)d, d + zb);
			be.d2h(h_ao, d + zb, ab);
		} catch (...) {be.free(d); throw;}
		be.free(d);
	TERRA_CATCH
}
int terra_tiles_mesh_shadows_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, const float light_pos[3], uint8_t *d_smask) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !d_zvals || !d_smask || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.tiles_mesh_shadows_dev(tile_xy, n, d_zvals, light_pos, d_smask); TERRA_CATCH
}
int terra_tiles_mesh_shadows_halo_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, const float light_pos[3], uint8_t *d_smask,
	const float *h_edge_in, const uint8_t *h_edge_in_present, float *h_edge_out)
{
	TERRA_CHECK_CTX if (n && (!tile_xy || !d_zvals || !d_smask || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	if ((h_edge_in == nullptr) != (h_edge_in_present == nullptr)) return terra::fail(TERRA_ERR_ARG, "edge_in and edge_in_present go together");
	TERRA_TRY ctx->eng.tiles_mesh_shadows_dev(tile_xy, n, d_zvals, light_pos, d_smask, h_edge_in, h_edge_in_present, h_edge_out); TERRA_CATCH
}
int terra_tiles_mesh_shadows_edges_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, const float light_pos[3], uint8_t *d_smask,
	const float *d_edge_in, const uint8_t *h_edge_in_present, float *d_edge_out)
{
	TERRA_CHECK_CTX if (n && (!tile_xy || !d_zvals || !d_smask || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	if ((d_edge_in == nullptr) != (h_edge_in_present == nullptr)) return terra::fail(TERRA_ERR_ARG, "edge_in and edge_in_present go together");
	TERRA_TRY ctx->eng.tiles_mesh_shadows_dev(tile_xy, n, d_zvals, light_pos, d_smask, nullptr, h_edge_in_present, nullptr, d_edge_in, d_edge_out); TERRA_CATCH
}
int terra_tiles_mesh_shadows(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, const float light_pos[3], uint8_t *h_smask) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !h_zvals || !h_smask || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (n == 0) return TERRA_OK;
	TERRA_TRY
		auto &be = ctx->eng.be;
		size_t const zb = (size_t)n*130*130*4, sb = (size_t)n*130*130;
		uint8_t *d = (uint8_t *)be.alloc(zb + sb);
		try {
			be.h2d(d, h_zvals, zb);
			ctx->eng.tiles_mesh_shadows_dev(tile_xy, n, (float const *)d, light_pos, d + zb);
			be.d2h(h_smask, d + zb, sb);
		} catch (...) {be.free(d); throw;}
		be.free(d);
	TERRA_CATCH
}
int terra_tiles_create_zvals_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, uint32_t iters_tt, float *d_zvals, terra_tile_stats *d_stats, uint8_t *d_normals, float *d_min_nz) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !d_zvals)) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.tiles_create_zvals_dev(tile_xy, n, iters_tt, d_zvals, d_stats, d_normals, d_min_nz); TERRA_CATCH
}
int terra_tiles_post_dev(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *d_zvals, terra_tile_stats *d_stats, uint8_t *d_normals, float *d_min_nz) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !d_zvals)) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY ctx->eng.tiles_post_dev(tile_xy, n, d_zvals, d_stats, d_normals, d_min_nz); TERRA_CATCH
}
int terra_tiles_post(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, const float *h_zvals, terra_tile_stats *h_stats, uint8_t *h_normals, float *h_min_nz) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !h_zvals)) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (n == 0) return TERRA_OK;
	TERRA_TRY
		auto &be = ctx->eng.be;
		size_t const zb = (size_t)n*130*130*4, sb = (size_t)n*sizeof(terra_tile_stats), nb = (size_t)n*129*129*4, mb = (size_t)n*4;
		uint8_t *d = (uint8_t *)be.alloc(zb + sb + nb + mb + 1024);
		float *dz = (float *)d; terra_tile_stats *ds = (terra_tile_stats *)(d + zb); uint8_t *dn = d + zb + sb; float *dm = (float *)(d + zb + sb + nb);
		try {
			be.h2d(dz, h_zvals, zb);
			ctx->eng.tiles_post_dev(tile_xy, n, dz, h_stats ? ds : nullptr, h_normals ? dn : nullptr, (h_normals && h_min_nz) ? dm : nullptr);
			if (h_stats) be.d2h(h_stats, ds, sb);
			if (h_normals) be.d2h(h_normals, dn, nb);
			if (h_normals && h_min_nz) be.d2h(h_min_nz, dm, mb);
		} catch (...) {be.free(d); throw;}
		be.free(d);
	TERRA_CATCH
}
int terra_tiles_create_zvals(terra_ctx *ctx, const int32_t *tile_xy, uint32_t n, uint32_t iters_tt, float *h_zvals, terra_tile_stats *h_stats, uint8_t *h_normals, float *h_min_nz) {
	TERRA_CHECK_CTX if (n && (!tile_xy || !h_zvals)) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (n == 0) return TERRA_OK;
	TERRA_TRY
		auto &be = ctx->eng.be;
		size_t const zb = (size_t)n*130*130*4, sb = (size_t)n*sizeof(terra_tile_stats), nb = (size_t)n*129*129*4, mb = (size_t)n*4;
		uint8_t *d = (uint8_t *)be.alloc(zb + sb + nb + mb + 1024);
		float *dz = (float *)d; terra_tile_stats *ds = (terra_tile_stats *)(d + zb); uint8_t *dn = d + zb + sb; float *dm = (float *)(d + zb + sb + nb);
		try {
			ctx->eng.tiles_create_zvals_dev(tile_xy, n, iters_tt, dz, h_stats ? ds : nullptr, h_normals ? dn : nullptr, (h_normals && h_min_nz) ? dm : nullptr);
			be.d2h(h_zvals, dz, zb);
			if (h_stats) be.d2h(h_stats, ds, sb);
			if (h_normals) be.d2h(h_normals, dn, nb);
			if (h_normals && h_min_nz) be.d2h(h_min_nz, dm, mb);
		} catch (...) {be.free(d); throw;}
		be.free(d);
	TERRA_CATCH
}

int terra_selftest_hot_sqrt(terra_ctx *ctx, uint32_t stride, uint64_t *mismatches) {
	TERRA_CHECK_CTX
	TERRA_TRY
	if (!mismatches) throw std::invalid_argument("terra_selftest_hot_sqrt: null result pointer");
	std::lock_guard<std::recursive_mutex> lk(ctx->eng_mtx);
	*mismatches = ctx->eng.selftest_hot_sqrt(stride);
	return TERRA_OK;
	TERRA_CATCH
}

// ---- voxels
int terra_voxel_fill_dev(terra_ctx *ctx, float *d_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo[3], const float vsz[3], const float off[3],
	float mag, float freq, int rs1, int rs2, int gen_mode, float zscale, int normalize)
{
	TERRA_CHECK_CTX if (!d_out || !lo || !vsz || !off) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (gen_mode < 0 || gen_mode > TERRA_MGEN_DWARP_GPU) return terra::fail(TERRA_ERR_ARG, "bad gen_mode");
	if (!(mag > 0.0f) || !(freq > 0.0f)) return terra::fail(TERRA_ERR_ARG, "voxel_fill: mag and freq must be > 0"); // assert(mag > 0.0 && freq > 0.0), src/upsurface.cpp:19
	TERRA_TRY ctx->eng.voxel_fill_dev(d_out, nx, ny, nz, lo, vsz, off, mag, freq, rs1, rs2, gen_mode, zscale, normalize); TERRA_CATCH
}
int terra_voxel_fill_slab_dev(terra_ctx *ctx, float *d_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo[3], const float vsz[3], const float off[3],
	float mag, float freq, int rs1, int rs2, int gen_mode, float zscale, int normalize, uint32_t y0, uint32_t nys)
{
	TERRA_CHECK_CTX if (!d_out || !lo || !vsz || !off) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (gen_mode < 0 || gen_mode > TERRA_MGEN_DWARP_GPU) return terra::fail(TERRA_ERR_ARG, "bad gen_mode");
	if (!(mag > 0.0f) || !(freq > 0.0f)) return terra::fail(TERRA_ERR_ARG, "voxel_fill: mag and freq must be > 0");
	TERRA_TRY ctx->eng.voxel_fill_dev(d_out, nx, ny, nz, lo, vsz, off, mag, freq, rs1, rs2, gen_mode, zscale, normalize, y0, nys); TERRA_CATCH
}
int terra_voxel_fill(terra_ctx *ctx, float *h_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo[3], const float vsz[3], const float off[3],
	float mag, float freq, int rs1, int rs2, int gen_mode, float zscale, int normalize)
{
	TERRA_CHECK_CTX if (!h_out) return terra::fail(TERRA_ERR_ARG, "null output");
	size_t const bytes = (size_t)nx*ny*nz*sizeof(float);
	if (bytes == 0) return terra::fail(TERRA_ERR_ARG, "voxel_fill: empty grid");
	float *d = nullptr;
	try {d = (float *)ctx->eng.be.alloc(bytes);} catch (std::exception const &e) {return terra::fail(TERRA_ERR_LIMIT, e.what());}
	int const rc = terra_voxel_fill_dev(ctx, d, nx, ny, nz, lo, vsz, off, mag, freq, rs1, rs2, gen_mode, zscale, normalize);
	if (rc == TERRA_OK) {try {ctx->eng.be.d2h(h_out, d, bytes);} catch (std::exception const &e) {ctx->eng.be.free(d); return terra::fail(TERRA_ERR_HIP, e.what());}}
	ctx->eng.be.free(d);
	return rc;
}

// ---- plumbing
int terra_malloc(terra_ctx *ctx, void **p, size_t bytes) {TERRA_CHECK_CTX if (!p) return terra::fail(TERRA_ERR_ARG, "null out"); TERRA_TRY *p = ctx->eng.be.alloc(bytes ? bytes : 1); TERRA_CATCH}
int terra_free(terra_ctx *ctx, void *p) {TERRA_CHECK_CTX TERRA_TRY if (p) {ctx->eng.be.sync(); ctx->eng.be.free(p);} TERRA_CATCH}
int terra_memcpy_h2d(terra_ctx *ctx, void *d, const void *h, size_t bytes) {TERRA_CHECK_CTX TERRA_TRY ctx->eng.be.h2d(d, h, bytes); ctx->eng.be.sync(); TERRA_CATCH}
int terra_memcpy_d2h(terra_ctx *ctx, void *h, const void *d, size_t bytes) {TERRA_CHECK_CTX TERRA_TRY ctx->eng.be.d2h(h, d, bytes); TERRA_CATCH}
int terra_timer_start(terra_ctx *ctx) {TERRA_CHECK_CTX TERRA_TRY ctx->eng.be.timer_start(); TERRA_CATCH}
int terra_timer_stop(terra_ctx *ctx, float *ms) {TERRA_CHECK_CTX TERRA_TRY float const t = ctx->eng.be.timer_stop(); if (ms) *ms = t; TERRA_CATCH}

} // extern "C"
#include "terra_multi.hpp"
#include "terra_dgrid.hpp"
