// terra_erosion.hpp -- droplet hydraulic erosion (apply_erosion, src/erosion.cpp:14-164), host+device core.
//
// The reference runs droplets `iter = 0,1,2,...` one after another over one shared padded grid; that serial order is the
// only deterministic semantics it has (its OpenMP loop is a data race) and droplet paths are chaotic, so the
// GPU implementation keeps SERIAL SEMANTICS EXACTLY while still running droplets in parallel:
//
//   optimistic multi-version fixed point (big grids)
//     round 1   every droplet of a window is traced in parallel against the untouched grid; its writes go to a private
//               log (cell -> final value) and the 8x8-cell blocks it touched go to a private block list;
//     round r   per-block linked lists (block -> droplets that touched it) are rebuilt; a droplet that shares a block with a
//               *lower-numbered* droplet whose log changed is re-traced, now reading, for every cell, the value written by
//               the highest-numbered lower droplet that wrote it (else the grid).  Logs are double-buffered so a round reads
//               only the previous round's logs.
//     stop      when no droplet needs a re-trace: every log then equals what the serial loop would have produced
//               (induction on droplet number; the lowest dirty droplet becomes final every round).
//     flush     for every logged cell the highest-numbered writer stores its value into the grid.
//   serial (tiles, and the overflow fall-back): one lane walks the droplets in order directly on the (LDS or HBM) grid.
//
// All arithmetic is the reference's, operation for operation, in fp32 without FMA contraction; std::min/max NaN behaviour
// and x86 float->int conversion are reproduced because the reference does produce NaNs (v = sqrtf(v*v + Kg*dh) with dh < 0).
#pragma once
#include "terra_common.hpp"
#include "terra_sincosf.hpp"

namespace terra {

// ------------------------------------------------------------------ grid addressing
// Padded coordinates X in [0,NX), Z in [0,NY), NX = xsize + 2*PAD.  The interior lives in the caller's buffer (in place);
// the PAD-wide ring is a small side buffer (2*PAD*NX + 2*PAD*ysize floats) -- no O(cells) pad / unpad copies.
struct grid_view_t {
	float *interior; // [ysize][xsize], x fastest
	float *border;   // ring storage, or nullptr when `interior` already IS a dense padded [NY][NX] array (tile / LDS mode)
	int xsize, ysize, NX, NY;

	TERRA_HD float *at(int X, int Z) const {
		if (border == nullptr) {return interior + (size_t)Z*NX + X;}
		int const x = X - EROSION_PAD, z = Z - EROSION_PAD;
		if ((unsigned)x < (unsigned)xsize && (unsigned)z < (unsigned)ysize) {return interior + (size_t)z*xsize + x;}
		if (Z < EROSION_PAD)           {return border + (size_t)Z*NX + X;}
		if (Z >= EROSION_PAD + ysize)  {return border + (size_t)(Z - ysize)*NX + X;}                 // rows PAD..2*PAD-1 of the band store
		size_t const side = (size_t)2*EROSION_PAD*NX + (size_t)z*(2*EROSION_PAD);
		return border + side + ((X < EROSION_PAD) ? X : (X - xsize));                                   // X - xsize in [PAD, 2*PAD)
	}
	TERRA_HD static size_t border_floats(int xsize, int ysize) {return (size_t)2*EROSION_PAD*(xsize + 2*EROSION_PAD) + (size_t)2*EROSION_PAD*ysize;}
};

struct erosion_consts_t {
	int   xsize, ysize, NX, NY;
	unsigned max_path_len;      // 4*NX*NY
	float erode_amount;
	float water_thresh;         // water_plane_z - HALF_DXY
	float relh_adj_tex, zmin, zrange, clip_hd1; // get_bare_ls_tid (src/Textures.cpp:1284-1287): zrange = zmax - zmin
	float two_pi;               // float(2.0*PI)
	float min_zval;
};

struct droplet_result_t {unsigned steps; int nan_seen;};

// ------------------------------------------------------------------ one droplet (src/erosion.cpp:67-155)
// MEM supplies: float read(int X, int Z) ; void write(int X, int Z, float v) ; bool begin_step(int xi, int zi) (false => abort trace)
// X/Z passed to read/write are already clamped into the padded grid (HMAP_INDEX).
template<class MEM> TERRA_HD droplet_result_t simulate_droplet(int iter, MEM &mem, erosion_consts_t const &ec) {
	float const Kq = 10, Kw = 0.001f, Kr = 0.9f, Kd = 0.02f, Ki = 0.1f, minSlope = 0.05f, g = 20, Kg = g*2;
	float const evap = 1 - Kw;
	int const NX = ec.NX, NY = ec.NY;
	droplet_result_t res = {0, 0};
	rand_gen_t rgen;
	rgen.set_state(iter + 11, 79*(int64_t)iter + 121);
	int xi = EROSION_PAD + (rgen.rand() % ec.xsize);
	int zi = EROSION_PAD + (rgen.rand() % ec.ysize);
	float xp = (float)xi, zp = (float)zi, xf = 0, zf = 0, s = 0, v = 0, w = 1, dx = 0, dz = 0;

#define TERRA_CX(x) imax(imin((x), NX-1), 0)
#define TERRA_CZ(z) imax(imin((z), NY-1), 0)
#define TERRA_HMAP(x, z) mem.read(TERRA_CX(x), TERRA_CZ(z))
#define TERRA_DEPOSIT_AT(X, Z, W) { \
	float const delta = ds*ec.erode_amount*(W); \
	if (!((X) < 0 || (Z) < 0 || (X) >= NX || (Z) >= NY)) {float const old = mem.read((X), (Z)); mem.write((X), (Z), old + delta);} \
}
#define TERRA_DEPOSIT(H) \
	TERRA_DEPOSIT_AT(xi  , zi  , (1-xf)*(1-zf)) \
	TERRA_DEPOSIT_AT(xi+1, zi  ,    xf *(1-zf)) \
	TERRA_DEPOSIT_AT(xi  , zi+1, (1-xf)*   zf ) \
	TERRA_DEPOSIT_AT(xi+1, zi+1,    xf *   zf ) \
	(H) += ds;

	if (!mem.begin_step(xi, zi)) {return res;}
	float h = TERRA_HMAP(xi, zi), h00 = h, h10 = TERRA_HMAP(xi+1, zi), h01 = TERRA_HMAP(xi, zi+1), h11 = TERRA_HMAP(xi+1, zi+1);
	unsigned numMoves = 0;

	for (; numMoves < ec.max_path_len; ++numMoves) {
		if (numMoves > 0 && !mem.begin_step(xi, zi)) {break;}
		float const gx = h00+h01-h10-h11, gz = h00+h10-h01-h11;
		dx = (dx-gx)*Ki+gx;
		dz = (dz-gz)*Ki+gz;
		float const dl = sqrtf(dx*dx+dz*dz);
		if (dl <= FLT_EPSILON) { // pick random dir: libm cosf/sinf in the reference, reproduced bit-for-bit (terra_sincosf.hpp)
			float const a = rgen.rand_float()*ec.two_pi;
			dx = glibc_cosf(a); dz = glibc_sinf(a);
		}
		else {dx /= dl; dz /= dl;}
		float const nxp = xp+dx, nzp = zp+dz;
		int const nxi = f2i_x86(floorf(nxp)), nzi = f2i_x86(floorf(nzp));
		float const nxf = nxp-(float)nxi, nzf = nzp-(float)nzi;
		float const nh00 = TERRA_HMAP(nxi, nzi), nh10 = TERRA_HMAP(nxi+1, nzi), nh01 = TERRA_HMAP(nxi, nzi+1), nh11 = TERRA_HMAP(nxi+1, nzi+1);
		float const nh = (nh00*(1-nxf)+nh10*nxf)*(1-nzf)+(nh01*(1-nxf)+nh11*nxf)*nzf;
		if (max_std(max_std(nh00, nh10), max_std(nh01, nh11)) < ec.water_thresh) break; // reached ocean water, sediment discarded

		bool const outside = (xi < 0 || zi < 0 || xi >= NX || zi >= NY);
		if (nh >= h || outside) {
			float ds = (nh-h)+0.001f;
			if (ds >= s || outside) {
				ds = s;
				TERRA_DEPOSIT(h)
				s = 0;
				break;
			}
			TERRA_DEPOSIT(h)
			s -= ds;
			v = 0;
		}
		float dh = h-nh;
		float const q = max_std(dh, minSlope)*v*w*Kq;
		float ds = s-q;
		if (ds >= 0) {
			ds *= Kd;
			TERRA_DEPOSIT(dh)
			s -= ds;
		}
		else {
			ds *= -Kr;
			ds = min_std(ds, dh*0.99f);
			float const relh = ec.relh_adj_tex + (nh - ec.zmin)/ec.zrange;
			ds = (float)((double)ds*((relh > ec.clip_hd1) ? 0.5 : 2.0)); // rock erodes slower than dirt
			for (int z = zi-1; z <= zi+2; ++z) {
				float const zo = (float)z-zp, zo2 = zo*zo;
				for (int x = xi-1; x <= xi+2; ++x) {
					float const xo = (float)x-xp;
					float wb = 1-(xo*xo+zo2)*0.25f;
					if (wb <= 0) continue;
					wb *= 0.1591549430918953f;
					float const delta = ds*ec.erode_amount*wb;
					int const cx = TERRA_CX(x), cz = TERRA_CZ(z);
					float const old = mem.read(cx, cz);
					mem.write(cx, cz, old - delta);
				}
			}
			dh -= ds;
			s  += ds;
		}
		v = sqrtf(v*v+Kg*dh);
		if (v != v) {res.nan_seen = 1;}
		w *= evap;
		xp = nxp; zp = nzp; xi = nxi; zi = nzi; xf = nxf; zf = nzf;
		h = nh; h00 = nh00; h10 = nh10; h01 = nh01; h11 = nh11;
	}
	res.steps = numMoves;
	return res;
#undef TERRA_CX
#undef TERRA_CZ
#undef TERRA_HMAP
#undef TERRA_DEPOSIT_AT
#undef TERRA_DEPOSIT
}

// ------------------------------------------------------------------ serial policy: reads/writes hit the grid directly
struct direct_mem_t {
	grid_view_t g;
	TERRA_HD bool  begin_step(int, int) {return true;}
	TERRA_HD float read(int X, int Z) const {return *g.at(X, Z);}
	TERRA_HD void  write(int X, int Z, float v) {*g.at(X, Z) = v;}
};

// ------------------------------------------------------------------ speculative policy
constexpr uint32_t SPEC_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t SPEC_NIL   = 0xFFFFFFFFu;
constexpr int      SPEC_BCACHE = 8;
enum {SPEC_F_LOG_OVERFLOW = 1, SPEC_F_BLK_OVERFLOW = 2, SPEC_F_NAN = 4};

struct spec_buffers_t {
	grid_view_t grid;
	erosion_consts_t ec;
	uint32_t first_iter;   // droplet number of window slot 0
	uint32_t W;            // slots in the window
	uint32_t cut;          // slots >= cut are excluded (overflowed droplet and everything after it)
	uint32_t cap_log2;     // log capacity = 1 << cap_log2
	uint32_t maxb;         // block-list capacity per droplet
	uint32_t bshift;       // block edge = 1 << bshift cells
	uint32_t nbx, nby;     // blocks per row / column of the padded grid
	uint32_t use_lists;    // 0 in round 1 (no cross-droplet reads yet)
	uint32_t *log_keys[2]; // [W][cap]
	float    *log_vals[2]; // [W][cap]
	uint32_t *blk_list[2]; // [W][maxb]
	uint32_t *blk_cnt[2];  // [W]
	uint64_t *chk[2];      // [W] order-dependent checksum of the write sequence
	uint32_t *cur;         // [W] which buffer holds the droplet's current trace
	uint32_t *need;        // [W] (re)trace in this round
	uint32_t *changed;     // [W] this round's trace differs from the previous one
	uint32_t *flags;       // [W]
	uint32_t *nsteps;      // [W]
	uint32_t *head;        // [nbx*nby] block -> first node
	uint32_t *next;        // [W*maxb]  node -> next node ; node id = slot*maxb + entry
	uint32_t *dirty_min;   // [nbx*nby] lowest changed droplet slot touching the block this round
	uint32_t *counters;    // [0] = any_need, [1] = min overflowed slot, [2] = traced this round, [3] = total steps (low), ...
};

#if defined(__HIP_DEVICE_COMPILE__)
#define TERRA_ATOMIC_MIN(p, v) atomicMin((p), (v))
#define TERRA_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define TERRA_ATOMIC_EXCH(p, v) atomicExch((p), (v))
#else
template<class T> inline T terra_host_atomic_min(T *p, T v) {T o = *p; if (v < o) *p = v; return o;}
template<class T> inline T terra_host_atomic_add(T *p, T v) {T o = *p; *p = o + v; return o;}
template<class T> inline T terra_host_atomic_exch(T *p, T v) {T o = *p; *p = v; return o;}
#define TERRA_ATOMIC_MIN(p, v) terra_host_atomic_min((p), (v))
#define TERRA_ATOMIC_ADD(p, v) terra_host_atomic_add((p), (v))
#define TERRA_ATOMIC_EXCH(p, v) terra_host_atomic_exch((p), (v))
#endif

TERRA_HD uint32_t spec_hash(uint32_t cell, uint32_t cap_log2) {return (cell*2654435761u) >> (32 - cap_log2);}

// probe a droplet's log for `cell`; returns slot index or SPEC_EMPTY-terminated miss (found=false)
TERRA_HD bool spec_log_find(uint32_t const *keys, float const *vals, uint32_t cap_log2, uint32_t cell, float &out) {
	uint32_t const mask = (1u << cap_log2) - 1;
	uint32_t h = spec_hash(cell, cap_log2);
	for (uint32_t n = 0; n <= mask; ++n, h = (h + 1) & mask) {
		uint32_t const k = keys[h];
		if (k == cell) {out = vals[h]; return true;}
		if (k == SPEC_EMPTY) return false;
	}
	return false;
}

struct spec_mem_t {
	spec_buffers_t const *sb;
	uint32_t slot;         // this droplet's window slot
	uint32_t *my_keys; float *my_vals; uint32_t *my_blks; // the "new" buffers (1 - cur)
	uint32_t nlog, nblk, flags;
	uint64_t chk;
	uint32_t bc_id[SPEC_BCACHE]; // recently touched blocks ...
	uint8_t  bc_shared[SPEC_BCACHE]; // ... and whether a lower-numbered droplet also touched them
	uint32_t bc_pos;

	TERRA_HD void init(spec_buffers_t const *sb_, uint32_t slot_) {
		sb = sb_; slot = slot_;
		uint32_t const nb = 1u - sb->cur[slot];
		size_t const cap = (size_t)1 << sb->cap_log2;
		my_keys = sb->log_keys[nb] + (size_t)slot*cap;
		my_vals = sb->log_vals[nb] + (size_t)slot*cap;
		my_blks = sb->blk_list[nb] + (size_t)slot*sb->maxb;
		nlog = 0; nblk = 0; flags = 0; chk = 1469598103934665603ull; bc_pos = 0;
		for (int i = 0; i < SPEC_BCACHE; ++i) {bc_id[i] = SPEC_NIL; bc_shared[i] = 0;}
	}
	// does any lower-numbered droplet of the window have this block in its (previous-round) footprint?
	TERRA_HD bool block_shared(uint32_t b) const {
		if (!sb->use_lists) return false;
		for (uint32_t node = sb->head[b]; node != SPEC_NIL; node = sb->next[node]) {
			if (node / sb->maxb < slot) return true;
		}
		return false;
	}
	TERRA_HD int touch_block(uint32_t b) { // returns cache index
		for (int i = 0; i < SPEC_BCACHE; ++i) {if (bc_id[i] == b) return i;}
		int const i = (int)(bc_pos++ % SPEC_BCACHE);
		bc_id[i] = b; bc_shared[i] = block_shared(b) ? 1 : 0;
		if (nblk >= sb->maxb) {flags |= SPEC_F_BLK_OVERFLOW; return i;}
		my_blks[nblk++] = b;
		return i;
	}
	TERRA_HD bool begin_step(int xi, int zi) { // footprint of one step = the 4x4 brush box, which also covers every read of that step
		int const x0 = imax(imin(xi-1, sb->ec.NX-1), 0) >> sb->bshift, x1 = imax(imin(xi+2, sb->ec.NX-1), 0) >> sb->bshift;
		int const z0 = imax(imin(zi-1, sb->ec.NY-1), 0) >> sb->bshift, z1 = imax(imin(zi+2, sb->ec.NY-1), 0) >> sb->bshift;
		touch_block((uint32_t)z0*sb->nbx + x0);
		if (x1 != x0) {touch_block((uint32_t)z0*sb->nbx + x1);}
		if (z1 != z0) {
			touch_block((uint32_t)z1*sb->nbx + x0);
			if (x1 != x0) {touch_block((uint32_t)z1*sb->nbx + x1);}
		}
		return (flags & (SPEC_F_LOG_OVERFLOW | SPEC_F_BLK_OVERFLOW)) == 0;
	}
	TERRA_HD float read(int X, int Z) {
		uint32_t const cell = (uint32_t)Z*sb->ec.NX + X;
		float v;
		if (nlog && spec_log_find(my_keys, my_vals, sb->cap_log2, cell, v)) return v; // own writes first
		uint32_t const b = (uint32_t)(Z >> sb->bshift)*sb->nbx + (uint32_t)(X >> sb->bshift);
		int const ci = touch_block(b);
		if (bc_shared[ci]) { // value written by the highest-numbered lower droplet, if any
			uint32_t best = SPEC_NIL;
			size_t const cap = (size_t)1 << sb->cap_log2;
			for (uint32_t node = sb->head[b]; node != SPEC_NIL; node = sb->next[node]) {
				uint32_t const j = node / sb->maxb;
				if (j >= slot || (best != SPEC_NIL && j <= best)) continue;
				uint32_t const cb = sb->cur[j];
				float vj;
				if (spec_log_find(sb->log_keys[cb] + (size_t)j*cap, sb->log_vals[cb] + (size_t)j*cap, sb->cap_log2, cell, vj)) {best = j; v = vj;}
			}
			if (best != SPEC_NIL) return v;
		}
		return *sb->grid.at(X, Z);
	}
	TERRA_HD void write(int X, int Z, float val) {
		uint32_t const cell = (uint32_t)Z*sb->ec.NX + X;
		uint32_t const mask = (1u << sb->cap_log2) - 1;
		uint32_t vb; memcpy(&vb, &val, 4);
		chk = (chk ^ (((uint64_t)cell << 32) | vb))*1099511628211ull;
		uint32_t h = spec_hash(cell, sb->cap_log2);
		for (uint32_t n = 0; n <= mask; ++n, h = (h + 1) & mask) {
			uint32_t const k = my_keys[h];
			if (k == cell) {my_vals[h] = val; return;}
			if (k == SPEC_EMPTY) {
				if (nlog >= mask - (mask >> 2)) {flags |= SPEC_F_LOG_OVERFLOW; return;} // keep load factor <= 0.75
				my_keys[h] = cell; my_vals[h] = val; ++nlog; return;
			}
		}
		flags |= SPEC_F_LOG_OVERFLOW;
	}
};

// ---- kernel bodies (one call per logical thread; the __global__ wrappers and the CPU emulator both call these)

// clear the "new" buffers of every droplet that will be traced this round: one thread per (slot, log entry)
TERRA_HD void spec_clear_body(spec_buffers_t const &sb, uint32_t slot, uint32_t entry) {
	if (slot >= sb.cut || !sb.need[slot]) return;
	uint32_t const nb = 1u - sb.cur[slot];
	sb.log_keys[nb][((size_t)slot << sb.cap_log2) + entry] = SPEC_EMPTY;
}

TERRA_HD void spec_trace_body(spec_buffers_t const &sb, uint32_t slot) {
	if (slot >= sb.cut || !sb.need[slot]) return;
	spec_mem_t mem;
	mem.init(&sb, slot);
	droplet_result_t const r = simulate_droplet((int)(sb.first_iter + slot), mem, sb.ec);
	uint32_t const nb = 1u - sb.cur[slot];
	sb.blk_cnt[nb][slot] = mem.nblk;
	sb.chk[nb][slot]     = mem.chk ^ ((uint64_t)r.steps << 40);
	sb.nsteps[slot]      = r.steps;
	sb.flags[slot]       = mem.flags | (r.nan_seen ? SPEC_F_NAN : 0);
	if (mem.flags & (SPEC_F_LOG_OVERFLOW | SPEC_F_BLK_OVERFLOW)) {TERRA_ATOMIC_MIN(&sb.counters[1], slot);}
	TERRA_ATOMIC_ADD(&sb.counters[2], 1u);
	TERRA_ATOMIC_ADD(&sb.counters[3], r.steps);
}

// after all traces of the round: publish dirty blocks of changed droplets, then flip their buffer
TERRA_HD void spec_post_body(spec_buffers_t const &sb, uint32_t slot, bool first_round) {
	if (slot >= sb.cut || !sb.need[slot]) {if (slot < sb.W) sb.changed[slot] = 0; return;}
	uint32_t const ob = sb.cur[slot], nb = 1u - ob;
	bool const changed = first_round || (sb.chk[ob][slot] != sb.chk[nb][slot]) || (sb.blk_cnt[ob][slot] != sb.blk_cnt[nb][slot]);
	sb.changed[slot] = changed ? 1u : 0u;
	if (changed) {
		if (!first_round) {
			uint32_t const *ol = sb.blk_list[ob] + (size_t)slot*sb.maxb;
			for (uint32_t e = 0; e < sb.blk_cnt[ob][slot]; ++e) {TERRA_ATOMIC_MIN(&sb.dirty_min[ol[e]], slot);}
		}
		uint32_t const *nl = sb.blk_list[nb] + (size_t)slot*sb.maxb;
		for (uint32_t e = 0; e < sb.blk_cnt[nb][slot]; ++e) {TERRA_ATOMIC_MIN(&sb.dirty_min[nl[e]], slot);}
	}
}
TERRA_HD void spec_flip_body(spec_buffers_t const &sb, uint32_t slot) {
	if (slot >= sb.cut || !sb.need[slot]) return;
	sb.cur[slot] = 1u - sb.cur[slot];
}
// rebuild block -> droplet lists from the current footprints (head[] was reset to SPEC_NIL before): one thread per (slot, entry)
TERRA_HD void spec_link_body(spec_buffers_t const &sb, uint32_t slot, uint32_t entry) {
	if (slot >= sb.cut) return;
	uint32_t const cb = sb.cur[slot];
	if (entry >= sb.blk_cnt[cb][slot]) return;
	uint32_t const b = sb.blk_list[cb][(size_t)slot*sb.maxb + entry];
	uint32_t const node = slot*sb.maxb + entry;
	sb.next[node] = TERRA_ATOMIC_EXCH(&sb.head[b], node);
}
// who must be re-traced next round: any droplet sharing a block with a lower-numbered droplet that changed this round
TERRA_HD void spec_mark_body(spec_buffers_t const &sb, uint32_t slot) {
	if (slot >= sb.W) return;
	uint32_t need = 0;
	if (slot < sb.cut) {
		uint32_t const cb = sb.cur[slot];
		uint32_t const *bl = sb.blk_list[cb] + (size_t)slot*sb.maxb;
		for (uint32_t e = 0; e < sb.blk_cnt[cb][slot]; ++e) {if (sb.dirty_min[bl[e]] < slot) {need = 1; break;}}
	}
	sb.need[slot] = need;
	if (need) {TERRA_ATOMIC_ADD(&sb.counters[0], 1u);}
}
// flush: the highest-numbered writer of a cell stores it; one thread per (slot, log entry). clamp_written applies max(min_zval, .)
TERRA_HD void spec_flush_body(spec_buffers_t const &sb, uint32_t slot, uint32_t entry, bool clamp_written) {
	if (slot >= sb.cut) return;
	uint32_t const cb = sb.cur[slot];
	size_t const cap = (size_t)1 << sb.cap_log2;
	uint32_t const cell = sb.log_keys[cb][(size_t)slot*cap + entry];
	if (cell == SPEC_EMPTY) return;
	uint32_t const X = cell % (uint32_t)sb.ec.NX, Z = cell / (uint32_t)sb.ec.NX;
	uint32_t const b = (Z >> sb.bshift)*sb.nbx + (X >> sb.bshift);
	for (uint32_t node = sb.head[b]; node != SPEC_NIL; node = sb.next[node]) {
		uint32_t const j = node / sb.maxb;
		if (j <= slot || j >= sb.cut) continue;
		uint32_t const jb = sb.cur[j];
		float vj;
		if (spec_log_find(sb.log_keys[jb] + (size_t)j*cap, sb.log_vals[jb] + (size_t)j*cap, sb.cap_log2, cell, vj)) return; // a later droplet owns the final value
	}
	float v = sb.log_vals[cb][(size_t)slot*cap + entry];
	if (clamp_written) {v = max_std(sb.ec.min_zval, v);}
	*sb.grid.at((int)X, (int)Z) = v;
}

// ring initialisation = the clamp-padded copy of src/erosion.cpp:31-37 restricted to the ring; one thread per ring float
TERRA_HD void border_init_body(grid_view_t const &g, size_t i) {
	int const PAD = EROSION_PAD;
	size_t const band = (size_t)2*PAD*g.NX;
	int X, Z;
	if (i < band) {int const r = (int)(i / g.NX); X = (int)(i % g.NX); Z = (r < PAD) ? r : (g.ysize + r);}
	else {size_t const k = i - band; int const z = (int)(k / (2*PAD)), c = (int)(k % (2*PAD)); Z = z + PAD; X = (c < PAD) ? c : (g.xsize + c);}
	int const sx = imax(imin(X - PAD, g.xsize-1), 0), sz = imax(imin(Z - PAD, g.ysize-1), 0);
	*g.at(X, Z) = g.interior[(size_t)sz*g.xsize + sx];
}

} // namespace terra
