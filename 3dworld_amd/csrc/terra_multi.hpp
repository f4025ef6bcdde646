// terra_multi.hpp -- several GPUs driven from ONE host process through the C ABI (include/terra.h: terra_multi_*).
//
// 3DWorld is one C++ process; it keeps eight generator objects in flight and collects them as they finish (height_gens[8], src/tiled_mesh.h:418,
// src/tiled_mesh.cpp:2317,2367-2416).  The equivalent here: N terra_ctx, one per device (a device may appear more than once), each driven by its own host
// thread for the duration of a call.  Work is dealt out in contiguous blocks of independent units -- tiles (tile_t::create_zvals needs nothing from a
// neighbour, src/tiled_mesh.cpp:515), rows of one heightmap (src/heightmap.cpp:139-143), y slabs of a voxel field -- so nothing in those paths is collective.
// The one exchange of the tile path is the mesh-shadow pass (tile_t::calc_shadows_for_light, src/tiled_mesh.cpp:664-692): tile columns are cut into strips,
// a strip walks its tile rows toward the light's far side in chunks of about as many rows as it has columns and, after every chunk, hands the outgoing edges
// of its border tiles to the next strip with device-to-device copies (hipMemcpyPeerAsync: one xGMI hop), which then runs the same chunk -- a software pipeline
// over (strips + chunks) steps.  The pass is a wavefront over the terrain (127 dependency levels for 64 x 64 tiles, at most 64 tiles per level): it spreads the
// DATA over the GPUs, it cannot run faster than on one.
// Included by terra_api_impl.hpp (so both libterra_hip.so and the test-only host emulation build it).
#pragma once
#include <thread>
#include <condition_variable>
#include <functional>

struct terra_multi {
	std::vector<terra_ctx *> ctxs;
};

namespace terra {

// f(i) on one host thread per context; the first exception becomes the call's error
template<class F> int multi_run(terra_multi *m, F f) {
	size_t const n = m->ctxs.size();
	std::vector<std::thread> th;
	std::mutex mtx; int code = TERRA_OK; std::string msg;
	auto body = [&](size_t i) {
		try {f((uint32_t)i);}
		catch (std::invalid_argument const &e) {std::lock_guard<std::mutex> l(mtx); if (code == TERRA_OK) {code = TERRA_ERR_ARG; msg = e.what();}}
		catch (std::logic_error const &e) {std::lock_guard<std::mutex> l(mtx); if (code == TERRA_OK) {code = TERRA_ERR_STATE; msg = e.what();}}
		catch (std::bad_alloc const &) {std::lock_guard<std::mutex> l(mtx); if (code == TERRA_OK) {code = TERRA_ERR_LIMIT; msg = "out of memory";}}
		catch (std::exception const &e) {std::lock_guard<std::mutex> l(mtx); if (code == TERRA_OK) {code = TERRA_ERR_HIP; msg = e.what();}}
		catch (...) {std::lock_guard<std::mutex> l(mtx); if (code == TERRA_OK) {code = TERRA_ERR_HIP; msg = "unknown exception in a context thread";}} // nothing may leave a thread (std::terminate inside a C ABI call)
	};
	th.reserve(n);
	size_t started = 1;
	try {for (; started < n; ++started) {th.emplace_back(body, started);}}
	catch (...) {std::lock_guard<std::mutex> l(mtx); if (code == TERRA_OK) {code = TERRA_ERR_LIMIT; msg = "could not start a host thread per context";}} // the threads already running are joined below
	if (started == n) {body(0);} // the caller's thread drives context 0 (skipped when the call already failed: the contexts that did start still finish their share)
	for (std::thread &t : th) {t.join();}
	if (code != TERRA_OK) return fail(code, msg.c_str());
	return TERRA_OK;
}
inline void multi_block(uint32_t n_units, uint32_t n_parts, uint32_t part, uint32_t &first, uint32_t &count) { // the same contiguous blocks as 3dworld_amd/dist.py
	uint64_t const parts = n_parts ? n_parts : 1, per = ((uint64_t)n_units + parts - 1)/parts; // 64-bit: n_units near 2^32 must not wrap
	uint64_t const lo = std::min<uint64_t>((uint64_t)part*per, n_units), hi = std::min<uint64_t>(((uint64_t)part + 1)*per, n_units);
	first = (uint32_t)lo; count = (uint32_t)(hi - lo);
}

// ---- mesh shadows of one terrain over several contexts: strips of tile columns, rows pipelined, border edges device to device
struct shadow_strip_t {
	std::vector<uint32_t> tiles;                 // indices into the caller's tile list, sorted by row (in processing order) then column
	float *d_z = nullptr, *d_ein = nullptr, *d_eout = nullptr; uint8_t *d_sm = nullptr;
	// chunk hand-over to the next strip
	std::mutex mtx; std::condition_variable cv; int rows_done = 0 /* chunks finished */; bool failed = false;
};

inline int multi_tiles_mesh_shadows(terra_multi *m, int32_t const *tile_xy, uint32_t n, float const *h_zvals, float const lpos[3], uint8_t *h_smask) {
	uint32_t const S = (uint32_t)m->ctxs.size(), zv = 130;
	if (n == 0) return TERRA_OK;
	int const sx = (lpos[0] < 0.0f) ? -1 : 1, sy = (lpos[1] < 0.0f) ? -1 : 1; // toward the light
	// strips of tile columns, strip 0 nearest the light in x (its tiles depend on no other strip)
	int32_t xmin = tile_xy[0], xmax = tile_xy[0];
	for (uint32_t i = 0; i < n; ++i) {xmin = std::min(xmin, tile_xy[2*i]); xmax = std::max(xmax, tile_xy[2*i]);}
	int64_t const width = (int64_t)xmax - xmin + 1, per = (width + S - 1)/S;
	auto strip_of = [&](int32_t tx) {int64_t const c = (sx > 0) ? ((int64_t)xmax - tx) : ((int64_t)tx - xmin); return (uint32_t)std::min<int64_t>(c/per, S - 1);};
	std::map<std::pair<int32_t, int32_t>, uint32_t> index;
	for (uint32_t i = 0; i < n; ++i) {
		if (!index.insert(std::make_pair(std::make_pair(tile_xy[2*i], tile_xy[2*i+1]), i)).second) return fail(TERRA_ERR_ARG, "terra_multi_tiles_mesh_shadows: a tile is named twice");
	}
	std::vector<shadow_strip_t> strips(S);
	std::vector<uint32_t> owner(n), pos(n); // strip and position inside the strip's tile order
	for (uint32_t i = 0; i < n; ++i) {owner[i] = strip_of(tile_xy[2*i]); strips[owner[i]].tiles.push_back(i);}
	for (shadow_strip_t &st : strips) {
		std::sort(st.tiles.begin(), st.tiles.end(), [&](uint32_t a, uint32_t b) { // rows toward the light first; inside a row any order (the engine sorts its batch into dependency levels)
			int64_t const ya = (int64_t)sy*tile_xy[2*a+1], yb = (int64_t)sy*tile_xy[2*b+1];
			if (ya != yb) return ya > yb;
			return (int64_t)sx*tile_xy[2*a] > (int64_t)sx*tile_xy[2*b];
		});
		for (uint32_t k = 0; k < st.tiles.size(); ++k) {pos[st.tiles[k]] = k;}
	}
	// every strip walks the rows of the terrain in the same order, in CHUNKS of `chunk` consecutive rows (about as many rows as a strip has columns: a call then covers a
	// square block of tiles, whose anti-diagonal dependency levels keep cols + chunk - 1 launches busy; one row per call would serialise a strip into rows x cols levels).
	// "chunk c of the previous strip is done" is one counter; chunks a strip has no tile in are empty steps
	std::vector<int32_t> all_rows;
	for (uint32_t i = 0; i < n; ++i) {all_rows.push_back(tile_xy[2*i+1]);}
	std::sort(all_rows.begin(), all_rows.end(), [&](int32_t a, int32_t b) {return (int64_t)sy*a > (int64_t)sy*b;});
	all_rows.erase(std::unique(all_rows.begin(), all_rows.end()), all_rows.end());
	size_t const chunk = (size_t)std::max<int64_t>(1, per), nchunks = (all_rows.size() + chunk - 1)/chunk;
	std::map<int32_t, size_t> chunk_of_row_w;
	for (size_t r = 0; r < all_rows.size(); ++r) {chunk_of_row_w[all_rows[r]] = r/chunk;}
	std::map<int32_t, size_t> const &chunk_of_row = chunk_of_row_w; // read-only from here on: the strip threads share it

	int const rc = multi_run(m, [&](uint32_t s) {
		shadow_strip_t &st = strips[s];
		auto &eng = m->ctxs[s]->eng; auto &be = eng.be;
		uint32_t const nt = (uint32_t)st.tiles.size();
		struct guard_t {shadow_strip_t &st; bool ok;
			~guard_t() {
				{std::lock_guard<std::mutex> l(st.mtx); if (!ok) {st.failed = true;} st.rows_done = 1 << 30;} st.cv.notify_all(); // never leave the next strip waiting
			}} guard{st, false};
		std::vector<float> hz((size_t)nt*zv*zv);
		std::vector<uint8_t> present((size_t)nt*2, 0);
		if (nt) {
			for (uint32_t k = 0; k < nt; ++k) {memcpy(hz.data() + (size_t)k*zv*zv, h_zvals + (size_t)st.tiles[k]*zv*zv, (size_t)zv*zv*4);}
			st.d_z = (float *)be.alloc(hz.size()*4); st.d_sm = (uint8_t *)be.alloc((size_t)nt*zv*zv);
			st.d_ein = (float *)be.alloc((size_t)nt*2*zv*4); st.d_eout = (float *)be.alloc((size_t)nt*2*zv*4);
			be.h2d(st.d_z, hz.data(), hz.size()*4);
		}
		uint32_t next = 0; // first tile of the strip (in its processing order) that has not been handed to the engine yet
		for (size_t c = 0; c < nchunks; ++c) {
			uint32_t const first = next;
			while (next < nt && chunk_of_row.at(tile_xy[2*st.tiles[next]+1]) == c) {++next;}
			uint32_t const cnt = next - first;
			if (cnt) {
				std::vector<int32_t> txy(2*(size_t)cnt);
				bool waited = false;
				for (uint32_t k = 0; k < cnt; ++k) {
					uint32_t const ti = st.tiles[first + k];
					txy[2*k] = tile_xy[2*ti]; txy[2*k+1] = tile_xy[2*ti+1];
					// sh_in_x: from the tile one row toward the light in the same column -- always in this strip; when it lies in an EARLIER chunk its outgoing edge is
					// copied in (inside the chunk the engine hands the edges on itself)
					auto up = index.find(std::make_pair(tile_xy[2*ti], tile_xy[2*ti+1] + sy));
					if (up != index.end() && pos[up->second] < first) {
						be.d2d(st.d_ein + ((size_t)(first + k)*2 + 0)*zv, st.d_eout + ((size_t)pos[up->second]*2 + 0)*zv, (size_t)zv*4);
						present[(size_t)(first + k)*2 + 0] = 1;
					}
					// sh_in_y: from the tile one column toward the light; when that tile belongs to the previous strip its edge comes from that strip's device
					auto side = index.find(std::make_pair(tile_xy[2*ti] + sx, tile_xy[2*ti+1]));
					if (side == index.end() || owner[side->second] == s) continue;
					shadow_strip_t &src = strips[owner[side->second]];
					if (!waited) { // the owner has finished this chunk (its kernels are drained before it counts the chunk)
						std::unique_lock<std::mutex> l(src.mtx);
						src.cv.wait(l, [&] {return src.rows_done > (int)c;});
						if (src.failed) throw std::runtime_error("terra_multi_tiles_mesh_shadows: a neighbouring strip failed");
						waited = true;
					}
					be.copy_from_peer(st.d_ein + ((size_t)(first + k)*2 + 1)*zv, m->ctxs[owner[side->second]]->eng.be, src.d_eout + ((size_t)pos[side->second]*2 + 1)*zv, (size_t)zv*4);
					present[(size_t)(first + k)*2 + 1] = 1;
				}
				eng.tiles_mesh_shadows_dev(txy.data(), cnt, st.d_z + (size_t)first*zv*zv, lpos, st.d_sm + (size_t)first*zv*zv, nullptr, present.data() + (size_t)first*2, nullptr,
					st.d_ein + (size_t)first*2*zv, st.d_eout + (size_t)first*2*zv);
				be.sync(); // the chunk's outgoing edges are in memory before the next strip is told
			}
			{std::lock_guard<std::mutex> l(st.mtx); st.rows_done = (int)c + 1;}
			st.cv.notify_all();
		}
		if (nt) {
			std::vector<uint8_t> sm((size_t)nt*zv*zv);
			be.d2h(sm.data(), st.d_sm, sm.size());
			for (uint32_t k = 0; k < nt; ++k) {memcpy(h_smask + (size_t)st.tiles[k]*zv*zv, sm.data() + (size_t)k*zv*zv, (size_t)zv*zv);}
		}
		guard.ok = true;
	});
	// the edge buffers are read by the neighbouring strip's peer copies: freed only when every thread is done
	for (uint32_t s = 0; s < S; ++s) {
		shadow_strip_t &st = strips[s];
		if (st.d_z) {auto &be = m->ctxs[s]->eng.be; for (void *p : {(void *)st.d_z, (void *)st.d_sm, (void *)st.d_ein, (void *)st.d_eout}) {if (p) be.free(p);}}
	}
	return rc;
}

} // namespace terra

extern "C" {

int terra_multi_create(terra_multi **out, const int *device_indices, uint32_t n) {
	if (!out) return terra::fail(TERRA_ERR_ARG, "terra_multi_create: null out pointer");
	*out = nullptr;
	if (!device_indices || n == 0 || n > 64) return terra::fail(TERRA_ERR_ARG, "terra_multi_create: 1 .. 64 device indices");
	terra_multi *m = new (std::nothrow) terra_multi();
	if (!m) return terra::fail(TERRA_ERR_LIMIT, "out of memory");
	for (uint32_t i = 0; i < n; ++i) {
		terra_ctx *c = nullptr;
		int const rc = terra_create(&c, device_indices[i]);
		if (rc != TERRA_OK) {for (terra_ctx *p : m->ctxs) terra_destroy(p); delete m; return rc;}
		m->ctxs.push_back(c);
	}
	for (uint32_t i = 0; i < n; ++i) {for (uint32_t j = 0; j < n; ++j) {if (i != j) m->ctxs[i]->eng.be.enable_peer(m->ctxs[j]->eng.be);}} // best effort: copies between devices work without it, through the host
	*out = m;
	return TERRA_OK;
}
void terra_multi_destroy(terra_multi *m) {if (!m) return; for (terra_ctx *c : m->ctxs) terra_destroy(c); delete m;}
uint32_t terra_multi_size(const terra_multi *m) {return m ? (uint32_t)m->ctxs.size() : 0u;}
terra_ctx *terra_multi_ctx(terra_multi *m, uint32_t i) {return (m && i < m->ctxs.size()) ? m->ctxs[i] : nullptr;}
void terra_multi_partition(uint32_t n_units, uint32_t n_parts, uint32_t part, uint32_t *first, uint32_t *count) {
	uint32_t f = 0, c = 0;
	if (n_parts && part < n_parts) {terra::multi_block(n_units, n_parts, part, f, c);}
	if (first) *first = f;
	if (count) *count = c;
}
#define TERRA_CHECK_MULTI if (!m || m->ctxs.empty()) return terra::fail(TERRA_ERR_ARG, "null terra_multi");
int terra_multi_foreach(terra_multi *m, int (*fn)(terra_ctx *ctx, uint32_t index, void *user), void *user) {
	TERRA_CHECK_MULTI if (!fn) return terra::fail(TERRA_ERR_ARG, "null callback");
	return terra::multi_run(m, [&](uint32_t i) {
		int const rc = fn(m->ctxs[i], i, user);
		if (rc < 0) {std::string const msg = terra_last_error(); throw std::runtime_error(msg.empty() ? std::string("terra_multi_foreach: the callback failed") : msg);}
	});
}
int terra_multi_synchronize(terra_multi *m) {TERRA_CHECK_MULTI return terra::multi_run(m, [&](uint32_t i) {m->ctxs[i]->eng.be.sync();});}
int terra_multi_init_scene(terra_multi *m, const terra_config *cfg) {
	TERRA_CHECK_MULTI if (!cfg) return terra::fail(TERRA_ERR_ARG, "null config");
	return terra::multi_run(m, [&](uint32_t i) {m->ctxs[i]->eng.init_scene(*cfg);});
}
// tile_t::create_zvals of n tiles, block i of the list on context i; per-context device outputs (d_zvals[i] etc. point into context i's device, sized for its block)
int terra_multi_tiles_create_zvals_dev(terra_multi *m, const int32_t *tile_xy, uint32_t n, uint32_t iters_tt, float *const *d_zvals, terra_tile_stats *const *d_stats, uint8_t *const *d_normals, float *const *d_min_nz) {
	TERRA_CHECK_MULTI if (n && (!tile_xy || !d_zvals)) return terra::fail(TERRA_ERR_ARG, "null argument");
	uint32_t const P = (uint32_t)m->ctxs.size();
	return terra::multi_run(m, [&](uint32_t i) {
		uint32_t first, cnt; terra::multi_block(n, P, i, first, cnt);
		if (cnt == 0) return;
		if (!d_zvals[i]) throw std::invalid_argument("terra_multi_tiles_create_zvals_dev: null block pointer");
		m->ctxs[i]->eng.tiles_create_zvals_dev(tile_xy + 2*(size_t)first, cnt, iters_tt, d_zvals[i], d_stats ? d_stats[i] : nullptr, d_normals ? d_normals[i] : nullptr, (d_normals && d_min_nz) ? d_min_nz[i] : nullptr);
	});
}
// the same with host outputs: every context copies its block into the caller's arrays
int terra_multi_tiles_create_zvals(terra_multi *m, const int32_t *tile_xy, uint32_t n, uint32_t iters_tt, float *h_zvals, terra_tile_stats *h_stats, uint8_t *h_normals, float *h_min_nz) {
	TERRA_CHECK_MULTI if (n && (!tile_xy || !h_zvals)) return terra::fail(TERRA_ERR_ARG, "null argument");
	uint32_t const P = (uint32_t)m->ctxs.size();
	return terra::multi_run(m, [&](uint32_t i) {
		uint32_t first, cnt; terra::multi_block(n, P, i, first, cnt);
		if (cnt == 0) return;
		int const rc = terra_tiles_create_zvals(m->ctxs[i], tile_xy + 2*(size_t)first, cnt, iters_tt, h_zvals + (size_t)first*130*130, h_stats ? h_stats + first : nullptr,
			h_normals ? h_normals + (size_t)first*129*129*4 : nullptr, (h_normals && h_min_nz) ? h_min_nz + first : nullptr);
		if (rc != TERRA_OK) throw std::runtime_error(terra_last_error());
	});
}
// ONE nx x ny heightmap as row strips (heightmap_t::proc_gen's row loop, src/heightmap.cpp:139-143): context i evaluates rows terra_multi_partition(ny, size, i) into d_out[i]
// (strip-local layout); h_min / h_max (optional): min / max of the whole map, folded on the host from the strips' (what run_erosion / from_floats need next)
int terra_multi_gen_grid_rows_dev(terra_multi *m, float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *const *d_out, float *h_min, float *h_max) {
	TERRA_CHECK_MULTI if (!d_out) return terra::fail(TERRA_ERR_ARG, "null output");
	uint32_t const P = (uint32_t)m->ctxs.size();
	std::vector<float> mn(P, INFINITY), mx(P, -INFINITY);
	bool const want = h_min || h_max;
	int const rc = terra::multi_run(m, [&](uint32_t i) {
		uint32_t first, cnt; terra::multi_block(ny, P, i, first, cnt);
		if (cnt == 0) return;
		if (!d_out[i]) throw std::invalid_argument("terra_multi_gen_grid_rows_dev: null strip pointer");
		float mm[2];
		m->ctxs[i]->eng.gen_grid_dev(x0, y0, dx, dy, nx, ny, flags, min_start_sin, d_out[i], want ? mm : nullptr, first, cnt);
		if (want) {mn[i] = mm[0]; mx[i] = mm[1];}
	});
	if (rc != TERRA_OK) return rc;
	if (want) {
		float a = mn[0], b = mx[0];
		for (uint32_t i = 1; i < P; ++i) {a = terra::min_std(a, mn[i]); b = terra::max_std(b, mx[i]);}
		if (h_min) *h_min = a;
		if (h_max) *h_max = b;
	}
	return TERRA_OK;
}
// ONE voxel field as y slabs (voxel_manager::create_procedural, src/voxels.cpp:278-346): context i fills slab terra_multi_partition(ny, size, i) into d_out[i]
int terra_multi_voxel_fill_dev(terra_multi *m, float *const *d_out, uint32_t nx, uint32_t ny, uint32_t nz, const float lo[3], const float vsz[3], const float off[3],
	float mag, float freq, int rs1, int rs2, int gen_mode, float zscale, int normalize)
{
	TERRA_CHECK_MULTI if (!d_out || !lo || !vsz || !off) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (gen_mode < 0 || gen_mode > TERRA_MGEN_DWARP_GPU) return terra::fail(TERRA_ERR_ARG, "bad gen_mode");
	if (!(mag > 0.0f) || !(freq > 0.0f)) return terra::fail(TERRA_ERR_ARG, "voxel_fill: mag and freq must be > 0");
	uint32_t const P = (uint32_t)m->ctxs.size();
	return terra::multi_run(m, [&](uint32_t i) {
		uint32_t first, cnt; terra::multi_block(ny, P, i, first, cnt);
		if (cnt == 0) return;
		if (!d_out[i]) throw std::invalid_argument("terra_multi_voxel_fill_dev: null slab pointer");
		m->ctxs[i]->eng.voxel_fill_dev(d_out[i], nx, ny, nz, lo, vsz, off, mag, freq, rs1, rs2, gen_mode, zscale, normalize, first, cnt);
	});
}
int terra_multi_tiles_mesh_shadows(terra_multi *m, const int32_t *tile_xy, uint32_t n, const float *h_zvals, const float light_pos[3], uint8_t *h_smask) {
	TERRA_CHECK_MULTI if (n && (!tile_xy || !h_zvals || !h_smask || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	try {return terra::multi_tiles_mesh_shadows(m, tile_xy, n, h_zvals, light_pos, h_smask);}
	catch (std::exception const &e) {return terra::fail(TERRA_ERR_HIP, e.what());}
}

} // extern "C"
