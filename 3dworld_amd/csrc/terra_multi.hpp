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
	struct strip_scratch_t {float *d_ein = nullptr, *d_eout = nullptr; size_t bytes = 0;}; // mesh-shadow edge buffers of context i (grow-only, kept between calls)
	std::vector<strip_scratch_t> shadow_scratch;
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

// ---- mesh shadows of one terrain over several contexts: strips of tile columns, rows pipelined, border edges device to device.
// Everything between two strips is ordered ON THE DEVICES: strip s records an event behind the kernels of its chunk c, strip s + 1 makes its stream wait for it and
// then gathers the border tiles' outgoing edges straight out of strip s's buffer (one launch over peer-mapped memory: a few KB over one xGMI hop; staged
// hipMemcpyPeerAsync copies when the devices cannot map each other).  The host threads only enqueue; the one host-side hand-shake left is "the event of chunk c has
// been RECORDED" (a stream may only wait for an event that was recorded before the wait is enqueued), which costs a condition variable per chunk and no GPU idle time.
struct shadow_layout_t {
	uint32_t S = 0; int sx = 1, sy = 1;
	std::vector<uint32_t> owner, pos;               // per tile of the caller's list: strip, position inside the strip's processing order
	std::vector<std::vector<uint32_t>> tiles;       // per strip: tile indices in processing order (rows toward the light first)
	std::vector<std::vector<uint32_t>> chunk_first; // per strip: first position of every chunk (+ the count at the end)
	size_t nchunks = 0;
	std::map<std::pair<int32_t, int32_t>, uint32_t> index;
};
inline bool shadow_layout(uint32_t S, int32_t const *tile_xy, uint32_t n, float const lpos[3], shadow_layout_t &L) {
	L.S = S; L.sx = (lpos[0] < 0.0f) ? -1 : 1; L.sy = (lpos[1] < 0.0f) ? -1 : 1; // toward the light
	L.owner.assign(n, 0); L.pos.assign(n, 0); L.tiles.assign(S, {}); L.chunk_first.assign(S, {}); L.index.clear(); L.nchunks = 0;
	if (n == 0) return true;
	// strips of tile columns, strip 0 nearest the light in x (its tiles depend on no other strip)
	int32_t xmin = tile_xy[0], xmax = tile_xy[0];
	for (uint32_t i = 0; i < n; ++i) {xmin = std::min(xmin, tile_xy[2*i]); xmax = std::max(xmax, tile_xy[2*i]);}
	int64_t const width = (int64_t)xmax - xmin + 1, per = (width + S - 1)/S;
	for (uint32_t i = 0; i < n; ++i) {
		if (!L.index.insert(std::make_pair(std::make_pair(tile_xy[2*i], tile_xy[2*i+1]), i)).second) return false; // a tile is named twice
		int64_t const c = (L.sx > 0) ? ((int64_t)xmax - tile_xy[2*i]) : ((int64_t)tile_xy[2*i] - xmin);
		L.owner[i] = (uint32_t)std::min<int64_t>(c/per, S - 1);
		L.tiles[L.owner[i]].push_back(i);
	}
	// every strip walks the rows of the terrain in the same order, in CHUNKS of `chunk` consecutive rows (about as many rows as a strip has columns: a call then covers a
	// square block of tiles, whose anti-diagonal dependency levels keep cols + chunk - 1 launches busy; one row per call would serialise a strip into rows x cols levels)
	std::vector<int32_t> all_rows;
	for (uint32_t i = 0; i < n; ++i) {all_rows.push_back(tile_xy[2*i+1]);}
	int const sy = L.sy, sx = L.sx;
	std::sort(all_rows.begin(), all_rows.end(), [&](int32_t a, int32_t b) {return (int64_t)sy*a > (int64_t)sy*b;});
	all_rows.erase(std::unique(all_rows.begin(), all_rows.end()), all_rows.end());
	size_t const chunk = (size_t)std::max<int64_t>(1, per);
	L.nchunks = (all_rows.size() + chunk - 1)/chunk;
	std::map<int32_t, size_t> chunk_of_row;
	for (size_t r = 0; r < all_rows.size(); ++r) {chunk_of_row[all_rows[r]] = r/chunk;}
	for (uint32_t s = 0; s < S; ++s) {
		std::vector<uint32_t> &t = L.tiles[s];
		std::sort(t.begin(), t.end(), [&](uint32_t a, uint32_t b) { // rows toward the light first; inside a row any order (the engine sorts its batch into dependency levels)
			int64_t const ya = (int64_t)sy*tile_xy[2*a+1], yb = (int64_t)sy*tile_xy[2*b+1];
			if (ya != yb) return ya > yb;
			return (int64_t)sx*tile_xy[2*a] > (int64_t)sx*tile_xy[2*b];
		});
		for (uint32_t k = 0; k < t.size(); ++k) {L.pos[t[k]] = k;}
		uint32_t next = 0;
		for (size_t c = 0; c < L.nchunks; ++c) {
			L.chunk_first[s].push_back(next);
			while (next < t.size() && chunk_of_row.at(tile_xy[2*t[next]+1]) == c) {++next;}
		}
		L.chunk_first[s].push_back(next);
	}
	return true;
}

struct shadow_strip_sync_t {std::mutex mtx; std::condition_variable cv; int recorded = 0 /* chunks whose event has been recorded */; bool failed = false; std::vector<void *> events;};
struct shadow_gather_t {unsigned long long src; uint32_t dst_row; uint32_t pad;}; // one incoming edge: source address (possibly in a peer's memory), destination row of d_ein

// device-resident form: d_z[s] / d_sm[s] hold strip s's tiles in the layout's processing order on context s's device
inline int multi_tiles_mesh_shadows_dev(terra_multi *m, int32_t const *tile_xy, uint32_t n, shadow_layout_t const &L, float *const *d_z, float const lpos[3], uint8_t *const *d_sm) {
	uint32_t const S = L.S, zv = 130;
	if (n == 0) return TERRA_OK;
	std::vector<shadow_strip_sync_t> sync(S);
	// per-strip edge buffers, kept by the handle between calls (grow-only): [tiles of the strip][2][130] floats in and out
	for (uint32_t s = 0; s < S; ++s) {
		size_t const need = (size_t)std::max<size_t>(L.tiles[s].size(), 1)*2*zv*4;
		terra_multi::strip_scratch_t &sc = m->shadow_scratch[s];
		if (sc.bytes < need) {
			auto &be = m->ctxs[s]->eng.be;
			if (sc.d_ein) {be.sync(); be.free(sc.d_ein); be.free(sc.d_eout); sc.d_ein = sc.d_eout = nullptr; sc.bytes = 0;}
			sc.d_ein = (float *)be.alloc(need); sc.d_eout = (float *)be.alloc(need); sc.bytes = need;
		}
	}
	int const rc = multi_run(m, [&](uint32_t s) {
		auto &eng = m->ctxs[s]->eng; auto &be = eng.be;
		std::vector<uint32_t> const &mine = L.tiles[s];
		uint32_t const nt = (uint32_t)mine.size();
		shadow_strip_sync_t &me = sync[s];
		struct guard_t {shadow_strip_sync_t &me; bool ok; ~guard_t() {{std::lock_guard<std::mutex> l(me.mtx); if (!ok) {me.failed = true;} me.recorded = 1 << 30;} me.cv.notify_all();}} guard{me, false}; // never leave the next strip waiting
		float *d_ein = m->shadow_scratch[s].d_ein, *d_eout = m->shadow_scratch[s].d_eout;
		std::vector<uint8_t> present((size_t)std::max<uint32_t>(nt, 1)*2, 0);
		std::vector<shadow_gather_t> gl;
		std::vector<int32_t> txy;
		bool const direct = (s == 0) || be.can_map(m->ctxs[s - 1]->eng.be); // the previous strip's buffer is addressable from this device's kernels
		for (size_t c = 0; c < L.nchunks; ++c) {
			uint32_t const first = L.chunk_first[s][c], cnt = L.chunk_first[s][c + 1] - first;
			if (cnt) {
				txy.resize(2*(size_t)cnt); gl.clear();
				bool waited = false;
				for (uint32_t k = 0; k < cnt; ++k) {
					uint32_t const ti = mine[first + k];
					txy[2*k] = tile_xy[2*ti]; txy[2*k+1] = tile_xy[2*ti+1];
					// sh_in_x: from the tile one row toward the light in the same column -- always in this strip; when it lies in an EARLIER chunk its outgoing edge is
					// gathered in (inside the chunk the engine hands the edges on itself)
					auto up = L.index.find(std::make_pair(tile_xy[2*ti], tile_xy[2*ti+1] + L.sy));
					if (up != L.index.end() && L.pos[up->second] < first) {
						gl.push_back(shadow_gather_t{(unsigned long long)(uintptr_t)(d_eout + ((size_t)L.pos[up->second]*2 + 0)*zv), (first + k)*2 + 0, 0});
						present[(size_t)(first + k)*2 + 0] = 1;
					}
					// sh_in_y: from the tile one column toward the light; when that tile belongs to the previous strip its edge lives on that strip's device
					auto side = L.index.find(std::make_pair(tile_xy[2*ti] + L.sx, tile_xy[2*ti+1]));
					if (side == L.index.end() || L.owner[side->second] == s) continue;
					uint32_t const os = L.owner[side->second];
					shadow_strip_sync_t &src = sync[os];
					if (!waited) { // the owner has RECORDED the event behind this chunk's kernels: this stream waits for it, the host does not
						void *ev = nullptr;
						{std::unique_lock<std::mutex> l(src.mtx); src.cv.wait(l, [&] {return src.recorded > (int)c;}); if (src.failed) throw std::runtime_error("terra_multi_tiles_mesh_shadows: a neighbouring strip failed"); ev = src.events[c];}
						if (ev) {be.event_wait(ev);}
						waited = true;
					}
					float const *peer = m->shadow_scratch[os].d_eout + ((size_t)L.pos[side->second]*2 + 1)*zv;
					if (direct) {gl.push_back(shadow_gather_t{(unsigned long long)(uintptr_t)peer, (first + k)*2 + 1, 0});}
					else {be.copy_from_peer(d_ein + ((size_t)(first + k)*2 + 1)*zv, m->ctxs[os]->eng.be, peer, (size_t)zv*4);}
					present[(size_t)(first + k)*2 + 1] = 1;
				}
				if (!gl.empty()) { // ONE launch gathers every incoming edge of the chunk
					shadow_gather_t *d_gl = eng.template scratch<shadow_gather_t>(eng.s_shadow_gather, gl.size());
					be.h2d_async(d_gl, gl.data(), gl.size()*sizeof(shadow_gather_t));
					float *ein = d_ein;
					be.launch(gl.size()*zv, [=] TERRA_LAMBDA (size_t j) {
						shadow_gather_t const g = d_gl[j / zv]; uint32_t const e = (uint32_t)(j % zv);
						ein[(size_t)g.dst_row*zv + e] = ((float const *)(uintptr_t)g.src)[e];
					});
				}
				eng.tiles_mesh_shadows_dev(txy.data(), cnt, d_z[s] + (size_t)first*zv*zv, lpos, d_sm[s] + (size_t)first*zv*zv, nullptr, present.data() + (size_t)first*2, nullptr,
					d_ein + (size_t)first*2*zv, d_eout + (size_t)first*2*zv);
			}
			void *ev = nullptr;
			if (cnt && s + 1 < S) {ev = be.event_create(); be.event_record(ev);} // behind the chunk's last kernel (the decode of its outgoing edges)
			{std::lock_guard<std::mutex> l(me.mtx); me.events.push_back(ev); me.recorded = (int)c + 1;}
			me.cv.notify_all();
		}
		be.sync(); // this strip's masks are complete (and nobody reads its edge buffer after the consumer's own sync)
		guard.ok = true;
	});
	for (uint32_t s = 0; s < S; ++s) {for (void *ev : sync[s].events) {if (ev) terra_backend_t::event_destroy(ev);}}
	return rc;
}

// host form: h_zvals in the caller's tile order in, h_smask out; strips are uploaded, processed where they lie, masks downloaded
inline int multi_tiles_mesh_shadows(terra_multi *m, int32_t const *tile_xy, uint32_t n, float const *h_zvals, float const lpos[3], uint8_t *h_smask) {
	uint32_t const S = (uint32_t)m->ctxs.size(), zv = 130;
	if (n == 0) return TERRA_OK;
	shadow_layout_t L;
	if (!shadow_layout(S, tile_xy, n, lpos, L)) return fail(TERRA_ERR_ARG, "terra_multi_tiles_mesh_shadows: a tile is named twice");
	std::vector<float *> d_z(S, nullptr); std::vector<uint8_t *> d_sm(S, nullptr);
	int rc = multi_run(m, [&](uint32_t s) {
		uint32_t const nt = (uint32_t)L.tiles[s].size();
		if (!nt) return;
		auto &be = m->ctxs[s]->eng.be;
		std::vector<float> hz((size_t)nt*zv*zv);
		for (uint32_t k = 0; k < nt; ++k) {memcpy(hz.data() + (size_t)k*zv*zv, h_zvals + (size_t)L.tiles[s][k]*zv*zv, (size_t)zv*zv*4);}
		d_z[s] = (float *)be.alloc(hz.size()*4); d_sm[s] = (uint8_t *)be.alloc((size_t)nt*zv*zv);
		be.h2d(d_z[s], hz.data(), hz.size()*4);
	});
	if (rc == TERRA_OK) {rc = multi_tiles_mesh_shadows_dev(m, tile_xy, n, L, d_z.data(), lpos, d_sm.data());}
	if (rc == TERRA_OK) {
		rc = multi_run(m, [&](uint32_t s) {
			uint32_t const nt = (uint32_t)L.tiles[s].size();
			if (!nt) return;
			std::vector<uint8_t> sm((size_t)nt*zv*zv);
			m->ctxs[s]->eng.be.d2h(sm.data(), d_sm[s], sm.size());
			for (uint32_t k = 0; k < nt; ++k) {memcpy(h_smask + (size_t)L.tiles[s][k]*zv*zv, sm.data() + (size_t)k*zv*zv, (size_t)zv*zv);}
		});
	}
	for (uint32_t s = 0; s < S; ++s) {auto &be = m->ctxs[s]->eng.be; if (d_z[s]) {be.sync(); be.free(d_z[s]);} if (d_sm[s]) be.free(d_sm[s]);}
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
	m->shadow_scratch.resize(n);
	for (uint32_t i = 0; i < n; ++i) {for (uint32_t j = 0; j < n; ++j) {if (i != j) m->ctxs[i]->eng.be.enable_peer(m->ctxs[j]->eng.be);}} // best effort: copies between devices work without it, through the host
	*out = m;
	return TERRA_OK;
}
void terra_multi_destroy(terra_multi *m) {
	if (!m) return;
	for (size_t i = 0; i < m->ctxs.size(); ++i) {
		terra_multi::strip_scratch_t &sc = m->shadow_scratch[i];
		if (sc.d_ein) {try {auto &be = m->ctxs[i]->eng.be; be.sync(); be.free(sc.d_ein); be.free(sc.d_eout);} catch (...) {}}
	}
	for (terra_ctx *c : m->ctxs) terra_destroy(c);
	delete m;
}
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
int terra_multi_shadow_layout(terra_multi *m, const int32_t *tile_xy, uint32_t n, const float light_pos[3], uint32_t *ctx_of_tile, uint32_t *pos_in_ctx, uint32_t *tiles_per_ctx) {
	TERRA_CHECK_MULTI if (n && (!tile_xy || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	try {
		terra::shadow_layout_t L;
		if (!terra::shadow_layout((uint32_t)m->ctxs.size(), tile_xy, n, light_pos, L)) return terra::fail(TERRA_ERR_ARG, "terra_multi_shadow_layout: a tile is named twice");
		for (uint32_t i = 0; i < n; ++i) {if (ctx_of_tile) ctx_of_tile[i] = L.owner[i]; if (pos_in_ctx) pos_in_ctx[i] = L.pos[i];}
		if (tiles_per_ctx) {for (uint32_t s = 0; s < L.S; ++s) tiles_per_ctx[s] = (uint32_t)L.tiles[s].size();}
		return TERRA_OK;
	} catch (std::exception const &e) {return terra::fail(TERRA_ERR_HIP, e.what());}
}
int terra_multi_tiles_mesh_shadows_dev(terra_multi *m, const int32_t *tile_xy, uint32_t n, float *const *d_zvals, const float light_pos[3], uint8_t *const *d_smask) {
	TERRA_CHECK_MULTI if (n && (!tile_xy || !d_zvals || !d_smask || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	try {
		terra::shadow_layout_t L;
		if (!terra::shadow_layout((uint32_t)m->ctxs.size(), tile_xy, n, light_pos, L)) return terra::fail(TERRA_ERR_ARG, "terra_multi_tiles_mesh_shadows_dev: a tile is named twice");
		for (uint32_t s = 0; s < L.S; ++s) {if (!L.tiles[s].empty() && (!d_zvals[s] || !d_smask[s])) return terra::fail(TERRA_ERR_ARG, "terra_multi_tiles_mesh_shadows_dev: null strip pointer");}
		return terra::multi_tiles_mesh_shadows_dev(m, tile_xy, n, L, d_zvals, light_pos, d_smask);
	} catch (std::exception const &e) {return terra::fail(TERRA_ERR_HIP, e.what());}
}
int terra_multi_tiles_mesh_shadows(terra_multi *m, const int32_t *tile_xy, uint32_t n, const float *h_zvals, const float light_pos[3], uint8_t *h_smask) {
	TERRA_CHECK_MULTI if (n && (!tile_xy || !h_zvals || !h_smask || !light_pos)) return terra::fail(TERRA_ERR_ARG, "null argument");
	try {return terra::multi_tiles_mesh_shadows(m, tile_xy, n, h_zvals, light_pos, h_smask);}
	catch (std::exception const &e) {return terra::fail(TERRA_ERR_HIP, e.what());}
}

} // extern "C"
