// terra_erosion.hpp -- droplet hydraulic erosion (apply_erosion, src/erosion.cpp:14-164), host+device core.
//
// The reference runs droplets `iter = 0,1,2,...` one after another over one shared padded grid; that serial order is the
// only deterministic semantics it has (its OpenMP loop is a data race) and droplet paths are chaotic, so the
// GPU implementation keeps SERIAL SEMANTICS EXACTLY while still running droplets in parallel:
//
//   optimistic multi-version fixed point over a sliding ring of W in-flight droplets (big grids; droplet i lives in slot i % W)
//     trace     a droplet is one 64-lane wave that walks the reference's scalar state machine on a 32x32 LDS window which follows it; what it writes goes to a
//               private VERSION: the distinct 8x8-cell blocks of its footprint in first-touch order, one 64-float PAGE and one 64-bit written-cells mask per block
//               (a write-back is one plain store, nothing is hashed, nothing is cleared between traces); a cell that enters the window is read from the page of the
//               highest-numbered LOWER droplet that has published a write of it, else from the grid;
//     publish   finished versions are compared with the droplet's previously published one page by page; blocks whose content changed are dirty;
//     re-trace  every higher droplet whose footprint contains a dirty block starts over -- from its last checkpoint before the first dirty block of its footprint
//               (every 32 steps a trace saves its state, footprint length and masks and keeps an undo log of rewritten cells), not from its spawn;
//     commit    the finished, valid prefix of the ring is flushed to the grid (highest committed writer of a cell stores it), its slots go to the next droplets;
//     rounds    one round = trace waves + five bookkeeping passes + commit waves + end-of-round, captured once into a hipGraph; traces of droplets that wait for a
//               slot are sliced (suspended at a step boundary, resumed next round); the lowest uncommitted droplet always has final inputs, so the fixed point
//               -- every version equal to the serial loop's -- is reached (terra_driver.hpp: speculative_erosion).
//   serial (the overflow fall-back, TERRA_ERODE_SERIAL*): one lane / one wave walks the droplets in order directly on the grid.
//   tiles: the whole clamp-padded 138x138 tile in LDS, one wave, droplets in order (terra_kernels.hpp: k_tile_erosion).
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
	// the same address without a branch (selects only): for loops that issue many loads back to back -- at()'s four-way branch costs an exec-mask region per cell, and
	// every region re-reads the view's fields when the view sits behind a pointer
	TERRA_HD float *at_sel(int X, int Z) const {
		int const x = X - EROSION_PAD, z = Z - EROSION_PAD;
		bool const in = ((unsigned)x < (unsigned)xsize) & ((unsigned)z < (unsigned)ysize);
		long long const io = (long long)z*xsize + x;                                              // interior (meaningless when !in)
		long long const top = (long long)Z*NX + X, bot = (long long)(Z - ysize)*NX + X;           // the two bands of the ring store
		long long const side = (long long)2*EROSION_PAD*NX + (long long)z*(2*EROSION_PAD) + ((X < EROSION_PAD) ? X : (X - xsize));
		long long const ro = (Z < EROSION_PAD) ? top : ((Z >= EROSION_PAD + ysize) ? bot : side);
		float *const dense = interior + top;                                                        // border == nullptr: `interior` IS the dense padded array
		float *const ring = in ? (interior + io) : (border + ro);
		return (border == nullptr) ? dense : ring;
	}
	TERRA_HD static size_t border_floats(int xsize, int ysize) {return (size_t)2*EROSION_PAD*(xsize + 2*EROSION_PAD) + (size_t)2*EROSION_PAD*ysize;}
};

struct erosion_consts_t {
	int   xsize, ysize, NX, NY;
	unsigned max_path_len;      // 4*NX*NY
	float erode_amount;
	float water_thresh;         // water_plane_z - HALF_DXY
	float relh_adj_tex, zmin, zrange, clip_hd1; // get_bare_ls_tid (src/Textures.cpp:1284-1287): zrange = zmax - zmin
	// the rock test `relh_adj_tex + (nh - zmin)/zrange > clip_hd1` is monotone in t = nh - zmin when zrange > 0 (a correctly rounded division by a positive constant
	// and a correctly rounded addition of a constant are both non-decreasing), so it is `t >= rock_t` for one threshold that the host finds by bisection over the
	// floats with the very expression above (make_rock_threshold): a compare instead of an IEEE division on the step's dependent chain.  rock_div != 0: no such
	// threshold (zrange <= 0 or NaN) -- the step divides.
	float rock_t; int rock_div;
	float two_pi;               // float(2.0*PI)
	float min_zval;
	int   lead_mode;            // placement of a recentred LDS window: 0 centred on the droplet, 1 always ahead of it, 2 ahead only when the last window lasted (speed only)
};

struct droplet_result_t {unsigned steps; int nan_seen;};

TERRA_HD int clampi(int v, int hi) {return imax(imin(v, hi), 0);} // HMAP_INDEX clamp (src/erosion.cpp:39)
TERRA_HD int sati(int v, int n) {return imax(imin(v, n + 8), -8);}   // keeps xi-1 / xi+2 free of signed overflow when a NaN position became INT_MIN; clamping afterwards is unchanged

// Every lane of a droplet's wave holds the same droplet state; telling the compiler so (v_readfirstlane) lets the integer / address / control
// part of a step run on the scalar unit with plain scalar branches instead of vector compares and exec-mask bookkeeping.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int wave_uniform(int v) {return __builtin_amdgcn_readfirstlane(v);}
__device__ __forceinline__ unsigned wave_uniform(unsigned v) {return (unsigned)__builtin_amdgcn_readfirstlane((int)v);}
__device__ __forceinline__ float wave_uniform(float v) {return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));}
#else
inline int wave_uniform(int v) {return v;}
inline unsigned wave_uniform(unsigned v) {return v;}
inline float wave_uniform(float v) {return v;}
#endif

// ------------------------------------------------------------------ one droplet (src/erosion.cpp:67-155)
// The scalar state machine below is the reference's loop verbatim; everything that touches the grid goes through MEM:
//   bool begin_step(xi, zi)                 footprint / residency hook for the step's 4x4 brush box; false => abort the trace
//   void corners(x, z, out[4])              HMAP(x,z), HMAP(x+1,z), HMAP(x,z+1), HMAP(x+1,z+1) (indices clamped, src/erosion.cpp:39-40)
//   void deposit(xi, zi, xf, zf, dse)       DEPOSIT_AT x4 with dse = ds*erode_amount (src/erosion.cpp:42-54)
//   void erode(xi, zi, xp, zp, dse)         the 4x4 radial brush (src/erosion.cpp:134-147)
// MEM is either scalar (one lane does everything) or wave-cooperative (64 lanes run this scalar code redundantly and
// split the brush / corner accesses between them); both give identical results.
// Loop-carried state of one droplet at the top of a step: a trace can stop there and be resumed later, bit for bit
struct droplet_state_t {
	int xi, zi;
	float xp, zp, xf, zf, s, v, w, dx, dz, h, h00, h10, h01, h11;
	unsigned numMoves;
	int nan_seen;
	rand_gen_t rgen;
};
constexpr unsigned DROPLET_NO_BUDGET = 0xFFFFFFFFu;

// the rock test of a step (src/erosion.cpp:133, get_bare_ls_tid src/Textures.cpp:1284-1287) as the reference evaluates it
TERRA_HD bool rock_test_div(erosion_consts_t const &ec, float t) {float const relh = ec.relh_adj_tex + t/ec.zrange; return relh > ec.clip_hd1;}
TERRA_HD bool rock_test(erosion_consts_t const &ec, float nh) {
	float const t = nh - ec.zmin;
	if (TERRA_UNLIKELY(ec.rock_div != 0)) return rock_test_div(ec, t);
	return t >= ec.rock_t;
}
// smallest float t with rock_test_div(t) (host, once per erosion call): bisection over the order-preserving integer image of the floats, -inf .. +inf
inline void make_rock_threshold(erosion_consts_t &ec) {
	ec.rock_t = 0.0f; ec.rock_div = 1;
	if (!(ec.zrange > 0.0f) || ec.relh_adj_tex != ec.relh_adj_tex || ec.clip_hd1 != ec.clip_hd1) return;
	auto key2f = [](uint32_t k) {uint32_t const u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &u, 4); return f;}; // increasing in k
	uint32_t lo = 0x007FFFFFu /* -inf */, hi = 0xFF800000u /* +inf */;
	if (rock_test_div(ec, key2f(lo))) {ec.rock_t = key2f(lo); ec.rock_div = 0; return;}        // true everywhere (NaN heights still fail the compare, as they fail the original)
	if (!rock_test_div(ec, key2f(hi))) {ec.rock_t = NAN; ec.rock_div = 0; return;}             // never true: t >= NaN is false
	while (hi - lo > 1) {uint32_t const mid = lo + (hi - lo)/2; if (rock_test_div(ec, key2f(mid))) {hi = mid;} else {lo = mid;}}
	ec.rock_t = key2f(hi); ec.rock_div = 0;
}

// spawn (src/erosion.cpp:67-84); false => the memory policy aborted the trace before the first step
template<class MEM> TERRA_HD bool droplet_start(int iter, MEM &mem, erosion_consts_t const &ec, droplet_state_t &d) {
	d.rgen.set_state(iter + 11, 79*(int64_t)iter + 121);
	d.xi = EROSION_PAD + (d.rgen.rand() % ec.xsize);
	d.zi = EROSION_PAD + (d.rgen.rand() % ec.ysize);
	d.xp = (float)d.xi; d.zp = (float)d.zi; d.xf = 0; d.zf = 0; d.s = 0; d.v = 0; d.w = 1; d.dx = 0; d.dz = 0;
	d.numMoves = 0; d.nan_seen = 0;
	d.h = d.h00 = d.h10 = d.h01 = d.h11 = 0;
	if (!mem.begin_step(d.xi, d.zi)) return false;
	float c[4];
	mem.corners(d.xi, d.zi, c);
	d.h = c[0]; d.h00 = c[0]; d.h10 = c[1]; d.h01 = c[2]; d.h11 = c[3];
	return true;
}

// the step loop (src/erosion.cpp:86-154), at most `budget` steps of it; true => the droplet is finished (d.numMoves = its step count)
template<class MEM> TERRA_HD bool droplet_run(droplet_state_t &d, MEM &mem, erosion_consts_t const &ec, unsigned budget) {
	float const Kq = 10, Kw = 0.001f, Kr = 0.9f, Kd = 0.02f, Ki = 0.1f, minSlope = 0.05f, g = 20, Kg = g*2;
	float const evap = 1 - Kw;
	int const NX = ec.NX, NY = ec.NY;
	int xi = d.xi, zi = d.zi;
	float xp = d.xp, zp = d.zp, xf = d.xf, zf = d.zf, s = d.s, v = d.v, w = d.w, dx = d.dx, dz = d.dz;
	float h = d.h, h00 = d.h00, h10 = d.h10, h01 = d.h01, h11 = d.h11;
	rand_gen_t rgen = d.rgen;
	unsigned numMoves = d.numMoves, used = 0;
	int nan_seen = d.nan_seen;
	bool finished = true;
	float c[4];

	for (; numMoves < ec.max_path_len; ++numMoves, ++used) {
		if (TERRA_UNLIKELY(used == budget)) {finished = false; break;}
		if (numMoves > 0 && TERRA_UNLIKELY(!mem.begin_step(xi, zi))) {break;}
		float const gx = h00+h01-h10-h11, gz = h00+h10-h01-h11;
		dx = (dx-gx)*Ki+gx;
		dz = (dz-gz)*Ki+gz;
		float const dl = sqrtf(dx*dx+dz*dz);
		if (TERRA_UNLIKELY(dl <= FLT_EPSILON)) { // pick random dir: libm cosf/sinf in the reference, reproduced bit-for-bit (terra_sincosf.hpp)
			float const a = rgen.rand_float()*ec.two_pi;
			dx = glibc_cosf(a); dz = glibc_sinf(a);
		}
		else {dx /= dl; dz /= dl;}
		float const nxp = xp+dx, nzp = zp+dz;
		int const nxi = f2i_x86(floorf(nxp)), nzi = f2i_x86(floorf(nzp));
		float const nxf = nxp-(float)nxi, nzf = nzp-(float)nzi;
		mem.corners(nxi, nzi, c);
		float const nh00 = c[0], nh10 = c[1], nh01 = c[2], nh11 = c[3];
		float const nh = (nh00*(1-nxf)+nh10*nxf)*(1-nzf)+(nh01*(1-nxf)+nh11*nxf)*nzf;
		if (max_std(max_std(nh00, nh10), max_std(nh01, nh11)) < ec.water_thresh) break; // reached ocean water, sediment discarded

		bool const outside = (xi < 0 || zi < 0 || xi >= NX || zi >= NY);
		if (nh >= h || outside) {
			float ds = (nh-h)+0.001f;
			if (ds >= s || outside) {
				ds = s;
				mem.deposit(xi, zi, xf, zf, ds*ec.erode_amount); h += ds; // deposit all sediment
				s = 0;
				break;
			}
			mem.deposit(xi, zi, xf, zf, ds*ec.erode_amount); h += ds;
			s -= ds;
			v = 0;
		}
		float dh = h-nh;
		float const q = max_std(dh, minSlope)*v*w*Kq;
		float ds = s-q;
		if (ds >= 0) {
			ds *= Kd;
			mem.deposit(xi, zi, xf, zf, ds*ec.erode_amount); dh += ds;
			s -= ds;
		}
		else {
			ds *= -Kr;
			ds = min_std(ds, dh*0.99f);
			ds = (float)((double)ds*(rock_test(ec, nh) ? 0.5 : 2.0)); // rock erodes slower than dirt
			mem.erode(xi, zi, xp, zp, ds*ec.erode_amount);
			dh -= ds;
			s  += ds;
		}
		v = sqrtf(v*v+Kg*dh);
		if (v != v) {nan_seen = 1;}
		w *= evap;
		xp = nxp; zp = nzp; xi = nxi; zi = nzi; xf = nxf; zf = nzf;
		h = nh; h00 = nh00; h10 = nh10; h01 = nh01; h11 = nh11;
	}
	d.xi = xi; d.zi = zi; d.xp = xp; d.zp = zp; d.xf = xf; d.zf = zf; d.s = s; d.v = v; d.w = w; d.dx = dx; d.dz = dz;
	d.h = h; d.h00 = h00; d.h10 = h10; d.h01 = h01; d.h11 = h11;
	d.rgen = rgen; d.numMoves = numMoves; d.nan_seen = nan_seen;
	return finished;
}

// ---- the common case as its own loop.  A step is "hot" when its 4x4 brush box is interior to the grid and resident (MEM::hot_ready), the
// direction comes from the gradient and the new corners lie inside that box (a droplet moves one cell: they do, except after a NaN / a quotient a rounding
// above 1): then nothing is clamped and nothing has to be fetched, and the step is the same arithmetic as in droplet_run().  Everything else (window shifts with
// their multi-version look-ups, the libm random direction, brushes at the border, reads after a NaN position, the footprint overflowing) makes this loop stop
// BEFORE the step has had any effect, and droplet_run() executes that one step.  A step is a chain of dependent instructions on ONE wave: every instruction is
// ~4 cycles, every LDS round trip ~100, every taken branch a refetch -- so the loop is written for the length of that chain:
//   * MEM::hot_ready fetches the step's 4x4 box, one cell per lane, at the TOP of the step; the four new corners are lane reads of that box (v_readlane), not a
//     second LDS round trip; deposit / brush update the box in registers and store it (no read-modify-write round trip, no wait for the store: LDS executes a
//     wave's instructions in order);
//   * the position is carried as floor(xp) in float next to the integer cell: the new cell is the old one plus a small exact float difference, the box test is one
//     compare per axis, a NaN fails it (no float -> int conversion of unchecked values, no range test against the grid: the box is interior);
//   * the rock test is a compare against a threshold found on the host (rock_test above) instead of an IEEE division;
//   * one step counter instead of separate path-length / budget tests; one compact run of instructions (the general step is ~100 KB of rarely executed code).
// Correctly rounded fp32 square root of the hot step, WITHOUT the parts of the compiler's sqrtf expansion that only matter below 2^-96: that expansion is v_sqrt_f32 (1 ulp)
// + one residual test against each neighbour (these seven instructions), wrapped in a scale-by-2^32 / unscale / class fix-up for tiny and special arguments (ten more
// slots of the step's dependent chain, twice per step).  For x >= 2^-96 (and for +-0, +inf, negative and NaN arguments) the seven instructions alone give bit for bit
// what the full expansion gives -- and tests/test_gpu_parity.py::test_hot_sqrt_equals_sqrtf checks that on the device; the host build is libm's sqrtf.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float sqrt_rn_core(float x) {
	float s = __builtin_amdgcn_sqrtf(x);
	float const sm = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1), sp = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
	float const rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
	s = (rm <= 0.0f) ? sm : s;
	s = (rp > 0.0f) ? sp : s;
	return s;
}
__device__ __forceinline__ float sqrt_rn(float x) {return TERRA_LIKELY(x >= 0x1p-96f) ? sqrt_rn_core(x) : sqrtf(x);} // (x < 2^-96, negative, NaN: the full expansion)
// the argument is a sum of squares and the result is only used when it exceeds FLT_EPSILON: below 2^-96 whatever v_sqrt_f32 returns (<= 2^-47) fails that test like the exact root
__device__ __forceinline__ float sqrt_rn_direction(float x) {return sqrt_rn_core(x);}
#else
inline float sqrt_rn(float x) {return sqrtf(x);}
inline float sqrt_rn_direction(float x) {return sqrtf(x);}
#endif

enum {DROPLET_EV_DONE = 0, DROPLET_EV_BUDGET = 1, DROPLET_EV_GENERAL = 2};
struct hot_pos_t {int xi, zi; float xp, zp, xf, zf, fxi, fzi, h, h00, h10, h01, h11, dx, dz;}; // what a step reads of where the droplet is and replaces when it has been made (fxi = floorf(xp) == (float)xi)
struct hot_run_t {float s, v, w; unsigned left, done; int nan_seen, ev; bool ready;};          // what it updates in place; ready: the box of the step to come is interior, resident and on its way

// One hot step from position `a`.  true: made, the droplet is at `b` (a is dead); false: not made -- run.ev says why -- and `a` is still the droplet's position.
// The caller alternates two position records (a -> b, b -> a): a step has to keep the old position until its brush is written while it computes the new one, and
// with a single record every field would be copied at the end of every step (a fifth of the loop's instructions were such moves).
template<class MEM> TERRA_HD bool droplet_hot_step(hot_pos_t const &a, hot_pos_t &b, hot_run_t &run, MEM &mem, erosion_consts_t const &ec, unsigned numMoves) {
	float const Kq = 10, Kw = 0.001f, Kr = 0.9f, Kd = 0.02f, Ki = 0.1f, minSlope = 0.05f, g = 20, Kg = g*2;
	float const evap = 1 - Kw;
	int const xi = wave_uniform(a.xi), zi = wave_uniform(a.zi);
	run.left = wave_uniform(run.left);
	if (run.left == 0) {run.ev = (numMoves + run.done >= ec.max_path_len) ? DROPLET_EV_DONE : DROPLET_EV_BUDGET; return false;}
	if (!run.ready) {mem.set_travel(a.dx, a.dz); run.ev = DROPLET_EV_GENERAL; return false;}
	float const gx = a.h00+a.h01-a.h10-a.h11, gz = a.h00+a.h10-a.h01-a.h11;
	float tdx = (a.dx-gx)*Ki+gx, tdz = (a.dz-gz)*Ki+gz;
	float const dl = sqrt_rn_direction(tdx*tdx+tdz*tdz);
	if (TERRA_UNLIKELY(!(dl > FLT_EPSILON))) {run.ev = DROPLET_EV_GENERAL; return false;} // random direction (or a NaN): general step
	tdx /= dl; tdz /= dl;
	float const nxp = a.xp+tdx, nzp = a.zp+tdz;
	float const nfx = floorf(nxp), nfz = floorf(nzp), ofx = nfx - a.fxi, ofz = nfz - a.fzi; // cell offsets -1, 0, 1 (exact: small integers), anything else leaves the box
	if (TERRA_UNLIKELY(!(fabsf(ofx) <= 1.0f && fabsf(ofz) <= 1.0f))) {run.ev = DROPLET_EV_GENERAL; return false;} // (a NaN position fails too): general step
	int const ox = wave_uniform((int)ofx), oz = wave_uniform((int)ofz);
	float c[4];
	mem.corners_hot(ox, oz, c);
	int const nxi = xi + ox, nzi = zi + oz;
	float const nxf = nxp-nfx, nzf = nzp-nfz; // nfx == (float)nxi
	// ---- from here on the step is executed
	float const nh00 = c[0], nh10 = c[1], nh01 = c[2], nh11 = c[3];
	float const nh = (nh00*(1-nxf)+nh10*nxf)*(1-nzf)+(nh01*(1-nxf)+nh11*nxf)*nzf;
	if (TERRA_UNLIKELY(max_std(max_std(nh00, nh10), max_std(nh01, nh11)) < ec.water_thresh)) {run.ev = DROPLET_EV_DONE; return false;}
	float h = a.h, s = run.s, v = run.v;
	if (nh >= h) { // `outside` is false: the box is interior
		float ds = (nh-h)+0.001f;
		if (ds >= s) {
			mem.deposit_hot(xi, zi, a.xf, a.zf, s*ec.erode_amount);
			run.s = 0;
			run.ev = DROPLET_EV_DONE; return false;
		}
		mem.deposit_hot(xi, zi, a.xf, a.zf, ds*ec.erode_amount); h += ds;
		s -= ds;
		v = 0;
	}
	float dh = h-nh;
	float const q = max_std(dh, minSlope)*v*run.w*Kq;
	float ds = s-q;
	if (ds >= 0) {
		ds *= Kd;
		mem.deposit_hot(xi, zi, a.xf, a.zf, ds*ec.erode_amount); dh += ds;
		s -= ds;
	}
	else {
		ds *= -Kr;
		ds = min_std(ds, dh*0.99f);
		ds *= rock_test(ec, nh) ? 0.5f : 2.0f; // rock erodes slower than dirt.  (float)((double)ds*(rock ? 0.5 : 2.0)) in the reference: the double product is exact, so its rounding to float IS the float product
		mem.erode_hot(xi, zi, a.xp, a.zp, ds*ec.erode_amount);
		dh -= ds;
		s  += ds;
	}
	v = sqrt_rn(v*v+Kg*dh);
	if (v != v) {run.nan_seen = 1;}
	run.s = s; run.v = v; run.w *= evap;
	b.xi = nxi; b.zi = nzi; b.xp = nxp; b.zp = nzp; b.xf = nxf; b.zf = nzf; b.fxi = nfx; b.fzi = nfz;
	b.h = nh; b.h00 = nh00; b.h10 = nh10; b.h01 = nh01; b.h11 = nh11; b.dx = tdx; b.dz = tdz;
	++run.done; --run.left;
	run.ready = wave_uniform((int)mem.hot_ready(nxi, nzi)) != 0; // the next step's box: its load is under way while this step's speed and the next step's direction are worked out
	return true;
}

template<class MEM> TERRA_HD int droplet_hot_steps(droplet_state_t &d, MEM &mem, erosion_consts_t const &ec, unsigned budget, unsigned &used) {
	unsigned numMoves = wave_uniform(d.numMoves);
	used = wave_uniform(used);
	// steps this call may still make: the path-length limit (src/erosion.cpp:86) and the caller's budget folded into one counter
	unsigned const room = (numMoves < ec.max_path_len) ? ec.max_path_len - numMoves : 0u, allow = (budget == DROPLET_NO_BUDGET) ? room : ((used < budget) ? budget - used : 0u);
	hot_run_t run;
	run.left = (room < allow) ? room : allow; run.done = 0;
	if (run.left == 0) {return (numMoves >= ec.max_path_len) ? DROPLET_EV_DONE : DROPLET_EV_BUDGET;}
	run.s = d.s; run.v = d.v; run.w = d.w; run.nan_seen = d.nan_seen; run.ev = DROPLET_EV_GENERAL;
	hot_pos_t p0, p1;
	// (wave_uniform: the droplet's state IS the same in every lane, but the general step reads it through paths the compiler cannot prove uniform; with provably uniform
	// values every test of the loop is a scalar branch instead of an exec-mask region)
	run.s = wave_uniform(run.s); run.v = wave_uniform(run.v); run.w = wave_uniform(run.w);
	p0.xi = wave_uniform(d.xi); p0.zi = wave_uniform(d.zi); p0.xp = wave_uniform(d.xp); p0.zp = wave_uniform(d.zp); p0.xf = wave_uniform(d.xf); p0.zf = wave_uniform(d.zf);
	p0.dx = wave_uniform(d.dx); p0.dz = wave_uniform(d.dz);
	p0.h = wave_uniform(d.h); p0.h00 = wave_uniform(d.h00); p0.h10 = wave_uniform(d.h10); p0.h01 = wave_uniform(d.h01); p0.h11 = wave_uniform(d.h11);
	p0.fxi = floorf(p0.xp); p0.fzi = floorf(p0.zp); // == (float)xi, (float)zi for a position inside the grid (hot_ready refuses everything else)
	run.ready = wave_uniform((int)mem.hot_ready(p0.xi, p0.zi)) != 0;
	bool odd;
	for (;;) {
		if (!droplet_hot_step(p0, p1, run, mem, ec, numMoves)) {odd = false; break;}
		if (!droplet_hot_step(p1, p0, run, mem, ec, numMoves)) {odd = true; break;}
	}
	hot_pos_t const &f = odd ? p1 : p0;
	numMoves += run.done; used += run.done;
	d.xi = f.xi; d.zi = f.zi; d.xp = f.xp; d.zp = f.zp; d.xf = f.xf; d.zf = f.zf; d.s = run.s; d.v = run.v; d.w = run.w; d.dx = f.dx; d.dz = f.dz;
	d.h = f.h; d.h00 = f.h00; d.h10 = f.h10; d.h01 = f.h01; d.h11 = f.h11;
	d.numMoves = numMoves; d.nan_seen = run.nan_seen;
	return run.ev;
}

// at most `budget` steps: hot steps in their own loop, single general steps in between; true => the droplet is finished
template<class MEM> TERRA_HD bool droplet_run_fast(droplet_state_t &d, MEM &mem, erosion_consts_t const &ec, unsigned budget) {
	unsigned used = 0;
	for (;;) {
		int const ev = droplet_hot_steps(d, mem, ec, budget, used);
		if (ev == DROPLET_EV_DONE) return true;
		if (ev == DROPLET_EV_BUDGET) return false;
		if (droplet_run(d, mem, ec, 1u)) return true;
		++used;
	}
}

template<class MEM> TERRA_HD droplet_result_t simulate_droplet(int iter, MEM &mem, erosion_consts_t const &ec) {
	droplet_state_t d;
	droplet_result_t res = {0, 0};
	if (!droplet_start(iter, mem, ec, d)) {return res;}
	droplet_run_fast(d, mem, ec, DROPLET_NO_BUDGET);
	res.steps = d.numMoves; res.nan_seen = d.nan_seen;
	return res;
}

// brush weight of cell (x,z) for a droplet at (xp,zp): w = max(0, 1 - (xo^2+zo^2)/4) * 1/(2 pi) (src/erosion.cpp:135-141); <= 0 => untouched
TERRA_HD float brush_weight(int x, int z, float xp, float zp) {
	float const zo = (float)z-zp, zo2 = zo*zo, xo = (float)x-xp;
	float wb = 1-(xo*xo+zo2)*0.25f;
	if (wb <= 0) return 0.0f;
	return wb*0.1591549430918953f;
}
TERRA_HD float deposit_weight(int q, float xf, float zf) { // (1-xf)*(1-zf), xf*(1-zf), (1-xf)*zf, xf*zf
	return ((q & 1) ? xf : (1-xf))*((q & 2) ? zf : (1-zf));
}

// ------------------------------------------------------------------ scalar policy: one lane, reads/writes hit the grid directly.
// Used by the TERRA_ERODE_SERIAL walk and the one-thread-per-tile cross-check path: an implementation independent of the wave code.
struct direct_mem_t {
	grid_view_t g;
	TERRA_HD bool begin_step(int, int) {return true;}
	TERRA_HD void corners(int x, int z, float out[4]) const {
		int const x0 = clampi(x, g.NX-1), x1 = clampi(x+1, g.NX-1), z0 = clampi(z, g.NY-1), z1 = clampi(z+1, g.NY-1);
		out[0] = *g.at(x0, z0); out[1] = *g.at(x1, z0); out[2] = *g.at(x0, z1); out[3] = *g.at(x1, z1);
	}
	TERRA_HD void set_travel(float, float) {}
	int hx = 0, hz = 0; // cell of the hot step in progress
	TERRA_HD bool hot_ready(int xi, int zi) {hx = xi; hz = zi; return xi-1 >= 0 && zi-1 >= 0 && xi+2 <= g.NX-1 && zi+2 <= g.NY-1;}
	TERRA_HD void corners_hot(int ox, int oz, float out[4]) const { // corners at (hx + ox, hz + oz), ox, oz in -1 .. 1: inside the step's interior 4x4 box
		int const x = hx + ox, z = hz + oz;
		out[0] = *g.at(x, z); out[1] = *g.at(x+1, z); out[2] = *g.at(x, z+1); out[3] = *g.at(x+1, z+1);
	}
	TERRA_HD void deposit_hot(int xi, int zi, float xf, float zf, float dse) {deposit(xi, zi, xf, zf, dse);}
	TERRA_HD void erode_hot(int xi, int zi, float xp, float zp, float dse) {erode(xi, zi, xp, zp, dse);}
	TERRA_HD void deposit(int xi, int zi, float xf, float zf, float dse) {
		for (int q = 0; q < 4; ++q) {
			int const X = xi + (q & 1), Z = zi + (q >> 1);
			float const delta = dse*deposit_weight(q, xf, zf);
			if (!(X < 0 || Z < 0 || X >= g.NX || Z >= g.NY)) {*g.at(X, Z) += delta;}
		}
	}
	TERRA_HD void erode(int xi, int zi, float xp, float zp, float dse) {
		for (int z = zi-1; z <= zi+2; ++z) {
			for (int x = xi-1; x <= xi+2; ++x) {
				float const wb = brush_weight(x, z, xp, zp);
				if (wb <= 0) continue;
				*g.at(clampi(x, g.NX-1), clampi(z, g.NY-1)) -= dse*wb;
			}
		}
	}
};

// ================================================================== wave-cooperative execution
// One droplet = one 64-lane wave (workgroups of the wave kernels are exactly one wave, so __syncthreads() is the wave's
// LDS / global visibility point).  All lanes run simulate_droplet's scalar code redundantly (identical registers, no
// divergence); TERRA_LANES splits the memory-heavy parts across lanes.  On the host (test emulator) a "wave" is one
// call and TERRA_LANES is a plain loop in lane order, which visits brush cells in the reference's z-major order.
#if defined(__HIP_DEVICE_COMPILE__)
#define TERRA_CLOCK() ((unsigned long long)wall_clock64()) // 100 MHz constant-rate counter (diagnostics only)
#define TERRA_LANES(l, n) for (int l = (int)(threadIdx.x & 63); l < (int)(n); l += 64)
#define TERRA_EACH_LANE(l) for (int l = (int)(threadIdx.x & 63), l##_once = 1; l##_once; l##_once = 0)
#define TERRA_LANE0 ((threadIdx.x & 63) == 0)
#define TERRA_LANE_SLOTS 1          // per-lane values that live across a wave sync: registers on the device ...
#define TERRA_LANE_SLOT(l) 0
#define TERRA_WAVE_SYNC() __syncthreads()
// a value held by lane `l` of the wave (l wave-uniform), as a scalar
#define TERRA_READLANE(arr, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (arr)[0]), (l)))
#define TERRA_READLANE_U32(arr, l) ((uint32_t)__builtin_amdgcn_readlane((int)(arr)[0], (int)wave_uniform((uint32_t)(l))))
// LDS executes one wave's instructions in order: a store by one lane is seen by a later load of another lane without waiting for anything.  The fence keeps the
// COMPILER from moving memory operations across it and costs no instruction (wavefront scope)
#define TERRA_WAVE_FENCE() do {__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();} while (0)
#define TERRA_ORDER_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup") // this wave's earlier memory operations have completed (no cache maintenance: a wait, and the compiler keeps the order)
#define TERRA_ATOMIC_MIN(p, v) atomicMin((p), (v))
#define TERRA_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define TERRA_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define TERRA_ATOMIC_OR(p, v) atomicOr((p), (v))
#define TERRA_ATOMIC_EXCH(p, v) atomicExch((p), (v))
#define TERRA_ATOMIC_CAS(p, c, v) atomicCAS((p), (c), (v))
#else
#define TERRA_CLOCK() 0ull
#define TERRA_LANES(l, n) for (int l = 0; l < (int)(n); ++l)
#define TERRA_EACH_LANE(l) for (int l = 0; l < 64; ++l)
#define TERRA_LANE0 true
#define TERRA_LANE_SLOTS 64         // ... one array row per lane on the host, where the lanes of a TERRA_EACH_LANE loop run one after another
#define TERRA_LANE_SLOT(l) (l)
#define TERRA_WAVE_SYNC() do {} while (0)
#define TERRA_READLANE(arr, l) ((arr)[(l)])
#define TERRA_READLANE_U32(arr, l) ((arr)[(l)])
#define TERRA_WAVE_FENCE() do {} while (0)
template<class T> inline T terra_host_atomic_min(T *p, T v) {T o = *p; if (v < o) *p = v; return o;}
template<class T> inline T terra_host_atomic_max(T *p, T v) {T o = *p; if (v > o) *p = v; return o;}
template<class T> inline T terra_host_atomic_add(T *p, T v) {T o = *p; *p = o + v; return o;}
template<class T> inline T terra_host_atomic_or(T *p, T v) {T o = *p; *p = o | v; return o;}
template<class T> inline T terra_host_atomic_exch(T *p, T v) {T o = *p; *p = v; return o;}
template<class T> inline T terra_host_atomic_cas(T *p, T c, T v) {T o = *p; if (o == c) *p = v; return o;}
#define TERRA_ATOMIC_MIN(p, v) terra_host_atomic_min((p), (v))
#define TERRA_ATOMIC_MAX(p, v) terra_host_atomic_max((p), (v))
#define TERRA_ORDER_FENCE() do {} while (0)
#define TERRA_ATOMIC_ADD(p, v) terra_host_atomic_add((p), (v))
#define TERRA_ATOMIC_OR(p, v) terra_host_atomic_or((p), (v))
#define TERRA_ATOMIC_EXCH(p, v) terra_host_atomic_exch((p), (v))
#define TERRA_ATOMIC_CAS(p, c, v) terra_host_atomic_cas((p), (c), (v))
#endif

// n slots each for the lanes of a wave out of a counter that the whole chip shares, with ONE atomic for the wave (every lane of the wave must call it, n may be 0;
// on the host the lanes come one after another and simply take theirs).  Chip-wide counters in one cache line serialise at L2: 26 M single increments of the
// written-cells counter were 15 % of a dense erosion run.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t wave_reserve(uint32_t *ctr, uint32_t n) {
	unsigned const lane = threadIdx.x & 63u;
	uint32_t incl = n;
#pragma unroll
	for (unsigned off = 1; off < 64; off <<= 1) {uint32_t const t = __shfl_up(incl, off, 64); if (lane >= off) {incl += t;}}
	uint32_t const total = __shfl(incl, 63, 64);
	uint32_t base = 0;
	if (lane == 63u && total) {base = atomicAdd(ctr, total);}
	base = __shfl(base, 63, 64);
	return base + incl - n;
}
#else
inline uint32_t wave_reserve(uint32_t *ctr, uint32_t n) {uint32_t const o = *ctr; *ctr = o + n; return o;}
#endif

// L2 load: read through to L2.  A trace's OWN pages are written by the other lanes of its wave (plain stores) in the same kernel; the CU's vector L1 may
// still hold the line from an earlier read, so those reads must not hit L1.  Other droplets' versions were written by earlier kernels (the boundary makes
// them visible) and use plain cached loads.
#if defined(__HIP_DEVICE_COMPILE__)
#define TERRA_L2_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define TERRA_L2_LOAD(p) (*(p))
#endif

// lane-parallel deposit / brush on a cell store addressed through DERIVED::cell(X,Z) (an LDS pointer), + DERIVED::mark(X,Z)
template<class DERIVED> struct wave_cell_ops {
	TERRA_HD DERIVED &self() {return *static_cast<DERIVED *>(this);}
	TERRA_HD void wsync() {TERRA_WAVE_SYNC();} // the wave's visibility point after a cooperative write (a policy whose workgroup holds more than one wave overrides it)
	TERRA_HD void deposit_cells(int xi, int zi, float xf, float zf, float dse, int NX, int NY) {
		TERRA_LANES(q, 4) { // the four cells are distinct: no write conflicts between lanes
			int const X = xi + (q & 1), Z = zi + (q >> 1);
			float const delta = dse*deposit_weight(q, xf, zf);
			if (!(X < 0 || Z < 0 || X >= NX || Z >= NY)) {*self().cell(X, Z) += delta; self().mark(X, Z);}
		}
		self().wsync();
	}
	// ---- hot step: the brush box is interior and resident.  Lane l < 16 keeps cell (xi-1 + (l & 3), zi-1 + (l >> 2)) of the step's 4x4 box in a register from the
	// top of the step on (box_load: one load per lane, in flight while the step's arithmetic runs); every read and write of the step lies inside that box
	float boxv[TERRA_LANE_SLOTS];
	float *boxp[TERRA_LANE_SLOTS]; // where the lane's cell lives (its address is computed once per step)
	TERRA_HD void box_load(int xi, int zi) {
		TERRA_WAVE_FENCE(); // after the stores of the step before
		TERRA_LANES(l, 16) {float *p = self().cell(xi-1 + (l & 3), zi-1 + (l >> 2)); boxp[TERRA_LANE_SLOT(l)] = p; boxv[TERRA_LANE_SLOT(l)] = *p;}
	}
	// the four corners at the step's cell + (ox, oz), ox, oz in -1 .. 1, out of its box
	TERRA_HD void box_corners(int ox, int oz, float out[4]) const {
		int const l = (oz + 1)*4 + (ox + 1);
		out[0] = TERRA_READLANE(boxv, l); out[1] = TERRA_READLANE(boxv, l + 1); out[2] = TERRA_READLANE(boxv, l + 4); out[3] = TERRA_READLANE(boxv, l + 5);
	}
	TERRA_HD void deposit_cells_hot(int xi, int zi, float xf, float zf, float dse) {
		TERRA_LANES(l, 16) { // the four cells (xi + (q & 1), zi + (q >> 1)) are lanes 5, 6, 9, 10 of the box
			int const bx = l & 3, bz = l >> 2;
			if (bx == 0 || bx == 3 || bz == 0 || bz == 3) continue;
			int const q = (bx - 1) | ((bz - 1) << 1);
			float const nv = boxv[TERRA_LANE_SLOT(l)] + dse*deposit_weight(q, xf, zf);
			boxv[TERRA_LANE_SLOT(l)] = nv; *boxp[TERRA_LANE_SLOT(l)] = nv; self().mark(xi + (q & 1), zi + (q >> 1));
		}
		TERRA_WAVE_FENCE();
	}
	TERRA_HD void erode_cells_hot(int xi, int zi, float xp, float zp, float dse) {
		TERRA_LANES(l, 16) {
			int const x = xi-1 + (l & 3), z = zi-1 + (l >> 2);
			float const wb = brush_weight(x, z, xp, zp);
			if (wb > 0) {float const nv = boxv[TERRA_LANE_SLOT(l)] - dse*wb; boxv[TERRA_LANE_SLOT(l)] = nv; *boxp[TERRA_LANE_SLOT(l)] = nv; self().mark(x, z);}
		}
		TERRA_WAVE_FENCE();
	}
#if defined(__HIP_DEVICE_COMPILE__)
	// The same three operations WITHOUT a divergent branch: all 64 lanes take part, lane l works on box cell l & 15 (four lanes per cell: they load the same word and
	// store the same value), a cell the operation does not touch gets its old value stored back, DERIVED::mark_if takes the condition.  One wave issues one instruction
	// per ~4.7 cycles whatever its kind and pays ~40 cycles per taken branch (tools/lat_probe.hip), so what a step costs is its instruction count: with no lane-dependent
	// branch left in the hot loop the compiler does not structurize it (no exec save/restore, no Flow blocks, no copies of the droplet state at every merge).
	TERRA_HD void box_load_all(int xi, int zi) {
		TERRA_WAVE_FENCE();
		int const l = (int)(threadIdx.x & 15u);
		float *p = self().cell(xi-1 + (l & 3), zi-1 + (l >> 2)); boxp[0] = p; boxv[0] = *p;
	}
	TERRA_HD void deposit_cells_hot_all(int xi, int zi, float xf, float zf, float dse) {
		int const l = (int)(threadIdx.x & 15u), bx = l & 3, bz = l >> 2;
		bool const inner = ((unsigned)(bx - 1) < 2u) & ((unsigned)(bz - 1) < 2u);
		int const q = ((bx - 1) & 1) | (((bz - 1) & 1) << 1);
		float const nv = boxv[0] + dse*deposit_weight(q, xf, zf);
		boxv[0] = inner ? nv : boxv[0]; *boxp[0] = boxv[0]; self().mark_if(xi + (q & 1), zi + (q >> 1), inner);
		TERRA_WAVE_FENCE();
	}
	TERRA_HD void erode_cells_hot_all(int xi, int zi, float xp, float zp, float dse) {
		int const l = (int)(threadIdx.x & 15u), x = xi-1 + (l & 3), z = zi-1 + (l >> 2);
		float const wb = brush_weight(x, z, xp, zp), nv = boxv[0] - dse*wb;
		bool const hit = wb > 0;
		boxv[0] = hit ? nv : boxv[0]; *boxp[0] = boxv[0]; self().mark_if(x, z, hit);
		TERRA_WAVE_FENCE();
	}
#else
	TERRA_HD void box_load_all(int xi, int zi) {box_load(xi, zi);}
	TERRA_HD void deposit_cells_hot_all(int xi, int zi, float xf, float zf, float dse) {deposit_cells_hot(xi, zi, xf, zf, dse);}
	TERRA_HD void erode_cells_hot_all(int xi, int zi, float xp, float zp, float dse) {erode_cells_hot(xi, zi, xp, zp, dse);}
#endif
	TERRA_HD void erode_cells(int xi, int zi, float xp, float zp, float dse, int NX, int NY) {
		if (TERRA_LIKELY(xi-1 >= 0 && zi-1 >= 0 && xi+2 <= NX-1 && zi+2 <= NY-1)) { // interior: 16 distinct cells, one lane each
			TERRA_LANES(l, 16) {
				int const x = xi-1 + (l & 3), z = zi-1 + (l >> 2);
				float const wb = brush_weight(x, z, xp, zp);
				if (wb > 0) {*self().cell(x, z) -= dse*wb; self().mark(x, z);}
			}
		}
		else if (TERRA_LANE0) { // clamping may fold several brush cells onto one grid cell: keep the reference's loop order on one lane
			for (int z = zi-1; z <= zi+2; ++z) {
				for (int x = xi-1; x <= xi+2; ++x) {
					float const wb = brush_weight(x, z, xp, zp);
					if (wb <= 0) continue;
					int const cx = clampi(x, NX-1), cz = clampi(z, NY-1);
					*self().cell(cx, cz) -= dse*wb; self().mark(cx, cz);
				}
			}
		}
		self().wsync();
	}
};

// ---- tile mode: the whole clamp-padded grid is resident in LDS
struct wave_lds_mem_t : wave_cell_ops<wave_lds_mem_t> {
	float *pad; int NX, NY;
	TERRA_HD float *cell(int X, int Z) const {return pad + Z*NX + X;}
	TERRA_HD void mark(int, int) {}
	TERRA_HD void mark_if(int, int, bool) {}
	TERRA_HD bool begin_step(int, int) {return true;}
	TERRA_HD void corners(int x, int z, float out[4]) const {
		int const x0 = clampi(x, NX-1), x1 = clampi(x+1, NX-1), z0 = clampi(z, NY-1), z1 = clampi(z+1, NY-1);
		out[0] = *cell(x0, z0); out[1] = *cell(x1, z0); out[2] = *cell(x0, z1); out[3] = *cell(x1, z1);
	}
	TERRA_HD void deposit(int xi, int zi, float xf, float zf, float dse) {deposit_cells(xi, zi, xf, zf, dse, NX, NY);}
	TERRA_HD void erode(int xi, int zi, float xp, float zp, float dse) {erode_cells(xi, zi, xp, zp, dse, NX, NY);}
	TERRA_HD void set_travel(float, float) {}
	TERRA_HD bool hot_ready(int xi, int zi) {
		if (!(((unsigned)(xi-1) <= (unsigned)(NX-4)) & ((unsigned)(zi-1) <= (unsigned)(NY-4)))) return false; // xi-1 >= 0 && xi+2 <= NX-1, the same in z
		box_load_all(xi, zi);
		return true;
	}
	TERRA_HD void corners_hot(int ox, int oz, float out[4]) const {box_corners(ox, oz, out);}
	TERRA_HD void deposit_hot(int xi, int zi, float xf, float zf, float dse) {deposit_cells_hot_all(xi, zi, xf, zf, dse);}
	TERRA_HD void erode_hot(int xi, int zi, float xp, float zp, float dse) {erode_cells_hot_all(xi, zi, xp, zp, dse);}
};

// ---- big grids: a WS x WS window of the grid follows the droplet in LDS; BACK is where cells come from / go to.
constexpr int EW = 32; // window edge (cells): 4 KiB of LDS per droplet; a droplet moves one cell per step, so a centred window lasts >= 13 steps
constexpr unsigned SPEC_MAXB = 256;      // most blocks a droplet's footprint may hold (their write masks and the block -> entry map live in LDS)
constexpr unsigned SPEC_MAP_SLOTS = 512; // open-addressed block -> entry map of the running trace, load factor <= 1/2
constexpr unsigned SPEC_PAGE = 64;       // cells of an 8 x 8 block = floats of a version page
constexpr unsigned SPEC_WIN_BLOCKS = ((EW >> 3) + 1)*((EW >> 3) + 1); // blocks a window can overlap
constexpr unsigned SPEC_CAND = 16, SPEC_CAND_MANY = 255; // (power of two.  4 -> 16: 30 000 droplets on 1024^2 320 -> 212 ms, sparse runs unchanged; 32: 196 ms but 24 KB of LDS per wave cost the 16384^2 run 4 %)
constexpr unsigned SPEC_OWN_NONE = 0xFFFFu;
// checkpoints of a trace (measured: profiles/r02_erosion_checkpoint_sweep.txt): every SPEC_CK_STEPS steps the window's dirty cells are written back and the droplet state, the footprint length, the write masks and
// the position in the undo log are saved, so that a re-trace can resume from the last checkpoint whose inputs are still valid instead of from the spawn
constexpr unsigned SPEC_CK_STEPS = 32, SPEC_CK_MAX = 16, SPEC_UNDO_MAX = 4096; // (defaults: spec_buffers_t::ck_steps / ck_max are the values in force, ck_max <= SPEC_CK_MAX)
// where a cell entering the window is read from: the grid, or float index (bits 0..30) into version buffer (bit 31)
constexpr uint32_t SPEC_SRC_GRID = 0xFFFFFFFFu;
struct spec_cand_t {uint32_t page, it; unsigned long long mask;}; // page = slot*maxb + entry, bit 31: the version buffer
struct wave_shared_t { // per-wave LDS scratch
	uint32_t flags, pad_;
	uint32_t undo_n, pad1_;                  // entries of the trace's undo log
	uint32_t n_shift, pad2_;                 // window moves of the trace (terra_erosion_report)
	unsigned long long chk;
	uint8_t blk_shared[64];                  // per block under the window: number of published LOWER versions that wrote it (SPEC_CAND_MANY: more than fit below)
	spec_cand_t cand[SPEC_WIN_BLOCKS][SPEC_CAND]; // those versions, highest droplet first: a cell's value comes from the first whose mask has the cell
	uint32_t blk_nonempty, pad3_;            // bit i: block i under the window has a lower version or a page of this trace (only those are resolved / looked at)
	uint16_t blk_own[32];                    // per block under the window: its entry in THIS trace's block list (SPEC_OWN_NONE: not in the footprint)
	uint32_t src[SPEC_WIN_BLOCKS][SPEC_PAGE]; // per cell of those blocks, resolved once per window move: where a cell that enters the window is read from -- SPEC_SRC_GRID, or the float index
	                                          // (bit 31: buffer) of the highest lower version's value.  Cells this trace wrote back itself and crowded blocks are patched after the pass (blk_special)
	uint32_t blk_special, pad5_;             // bit i: block i holds cells of this trace's own pages, or has more lower versions than the candidate list holds
	uint32_t map_keys[SPEC_MAP_SLOTS];       // block id (SPEC_NIL: free)
	uint8_t  map_ent[SPEC_MAP_SLOTS];        // its entry in the trace's block list = its page
	unsigned long long masks[SPEC_MAXB];     // per entry: which cells of the block this trace has written back to its page
	uint32_t wrote[SPEC_MAXB/32];            // bit e: entry e was written back to in THIS slice (spec_back_t::store)
};

template<class BACK> struct window_mem_t : wave_cell_ops<window_mem_t<BACK>> {
	float *win; uint8_t *dirty; // LDS: EW*EW each.  A window move shifts the overlap in place (old cells to registers, wave sync, new positions)
	int wx0, wz0, NX, NY; bool have;
	int lead_x = 0, lead_z = 0; // where the droplet is heading (-1, 0, 1 per axis): a recentred window is placed ahead of it
	unsigned long long clk_shift = 0, clk_sh_flush = 0, clk_sh_prep = 0, clk_sh_load = 0; // time inside recenter() and its parts (diagnostics)
	int lead_mode = 2, steps_in_window = 0; // a window that lasted only a few steps means the droplet turned back (it circles in a pit): centre the next one instead
	BACK back;
	TERRA_HD void init(float *w, uint8_t *d, int nx, int ny) {win = w; dirty = d; NX = nx; NY = ny; wx0 = wz0 = 0; have = false;}
	TERRA_HD bool in_window(int X, int Z) const {return have && (unsigned)(X - wx0) < (unsigned)EW && (unsigned)(Z - wz0) < (unsigned)EW;}
	TERRA_HD float *cell(int X, int Z) const {return win + (Z - wz0)*EW + (X - wx0);}
	TERRA_HD void mark(int X, int Z) {dirty[(Z - wz0)*EW + (X - wx0)] = 1;}
	TERRA_HD void flush() { // final write-back of every dirty cell
		if (have) {
			TERRA_LANES(i, EW*EW) {if (dirty[i]) {back.store(wx0 + (i % EW), wz0 + (i / EW), win[i]); dirty[i] = 0;}}
			back.note_written_rect(wx0, wz0);
		}
		TERRA_WAVE_SYNC();
	}
	// Move the window so that (cx,cz) is near its centre.  Cells that stay inside are copied LDS -> LDS together with their dirty bit
	// (no global traffic, no log look-up); dirty cells that leave are written back; cells that enter are fetched with all plain grid
	// loads of a lane issued back to back (one HBM latency per shift), then patched where a multi-version look-up is needed.
	TERRA_HD void set_travel(float dx, float dz) {lead_x = (dx > 0.35f) ? 1 : ((dx < -0.35f) ? -1 : 0); lead_z = (dz > 0.35f) ? 1 : ((dz < -0.35f) ? -1 : 0);}
	// the sixteen cells (k*64 + lane) of a lane in the new window at (nx0, nz0): values into g[], dirty bits returned
	template<bool FAST, class SRC> TERRA_HD uint32_t load_pass(SRC const &S, int lane, int nx0, int nz0, float (&g)[EW*EW/64]) {
		constexpr int PER_LANE = EW*EW/64;
		float gl[PER_LANE];
		uint32_t db = 0, old_mask = 0, in_mask = 0;
		int const lx = lane % EW, lz = lane / EW, X = nx0 + lx; // i = k*64 + lane: column lx, row 2 k + lz
		bool const col_in = X < NX, col_old = have && (unsigned)(X - wx0) < (unsigned)EW;
		float const *const p0 = S.interior_ptr(col_in ? X : nx0, nz0 + lz); // FAST: the cell of row 2 k + lz is p0 + k*S.row_step()
#pragma unroll
		for (int k = 0; k < PER_LANE; ++k) {
			int const Z = nz0 + 2*k + lz;
			bool const in_old = col_old && (unsigned)(Z - wz0) < (unsigned)EW, inside = col_in && Z < NY;
			int const o = in_old ? (Z - wz0)*EW + (X - wx0) : 0;
			g[k] = win[o]; db |= (uint32_t)((dirty[o] != 0) & in_old) << k;
			int const Xs = inside ? X : nx0, Zs = inside ? Z : nz0; // (nx0, nz0) is a cell of the grid and of the prepared window
			uint32_t const cd = S.code(Xs, Zs);
			float const *const gp = FAST ? ((inside ? p0 + (size_t)k*S.row_step() : S.interior_ptr(nx0, nz0))) : S.g.at_sel(Xs, Zs);
			gl[k] = *S.ptr(cd, gp);
			old_mask |= (uint32_t)in_old << k; in_mask |= (uint32_t)inside << k;
		}
#pragma unroll
		for (int k = 0; k < PER_LANE; ++k) {g[k] = ((old_mask >> k) & 1u) ? g[k] : (((in_mask >> k) & 1u) ? gl[k] : 0.0f);}
		return db;
	}
	TERRA_HD void recenter(int cx, int cz) {
		// the droplet sits a quarter of the window behind the centre, in the direction it came from: ~21 instead of ~13 steps until its brush box leaves again.
		// Where the window lies never changes a result (it is a cache of the backing store), only how often it moves.
		unsigned long long const clk0 = TERRA_CLOCK();
		if (lead_mode == 1 || (lead_mode == 2 && (!have || steps_in_window >= 10))) {cx += lead_x*(EW/4); cz += lead_z*(EW/4);}
		steps_in_window = 0;
		int const nx0 = clampi(cx - EW/2, imax(NX - EW, 0)), nz0 = clampi(cz - EW/2, imax(NY - EW, 0));
		if (have) {
			TERRA_LANES(i, EW*EW) {
				if (dirty[i]) {
					int const X = wx0 + (i % EW), Z = wz0 + (i / EW);
					if (!((unsigned)(X - nx0) < (unsigned)EW && (unsigned)(Z - nz0) < (unsigned)EW)) {back.store(X, Z, win[i]);}
				}
			}
			back.note_written_rect(wx0, wz0);
			TERRA_WAVE_SYNC();
		}
		unsigned long long const clk1 = TERRA_CLOCK();
		back.prepare_window(nx0, nz0);
		unsigned long long const clk2 = TERRA_CLOCK();
		unsigned long long clk3 = clk2; (void)clk3;
		constexpr int PER_LANE = EW*EW/64;
		typename BACK::src_t const S = back.src_snapshot(); // (after prepare_window: the sources of the new window's cells are in LDS)
		float gv[TERRA_LANE_SLOTS][PER_LANE]; uint32_t dbits[TERRA_LANE_SLOTS]; // a lane's 16 cells of the new window and their dirty bits
		// Every cell of the new window: from the old window (LDS) if it stays, else ONE load from where its current value lives -- the grid or a lower version's page (S.code: one
		// LDS word per cell of a block that has versions, resolved per block in prepare_window).  No branch and no use of a loaded value inside the pass -- a lane whose cell stays (or
		// lies outside the grid) loads some valid word it will not use -- so the sixteen loads of a lane are in flight together: one memory latency per move.  A window that lies
		// inside the caller's array (all but the ones at the map's rim) addresses the grid with one add per cell.  (As first written -- grid_view_t::at_sel and the candidate /
		// own-page resolution per cell -- the pass was 2400 instructions, 5-10 us per move: profiles/r04_erosion_recenter_isa.txt.)
		bool const fast = S.interior_window(nx0, nz0);
		int const owx0 = wx0, owz0 = wz0; bool const had = have;
		TERRA_EACH_LANE(lane) {
			float (&g)[PER_LANE] = gv[TERRA_LANE_SLOT(lane)];
			if (fast) {dbits[TERRA_LANE_SLOT(lane)] = load_pass<true>(S, lane, nx0, nz0, g);}
			else {dbits[TERRA_LANE_SLOT(lane)] = load_pass<false>(S, lane, nx0, nz0, g);}
		}
		clk3 = TERRA_CLOCK();
		TERRA_WAVE_SYNC(); // every lane has taken what it needs from the old window: the new one goes into the same LDS
		TERRA_EACH_LANE(lane) {
			float const (&g)[PER_LANE] = gv[TERRA_LANE_SLOT(lane)];
			uint32_t const db = dbits[TERRA_LANE_SLOT(lane)];
#pragma unroll
			for (int k = 0; k < PER_LANE; ++k) {int const i = k*64 + lane; win[i] = g[k]; dirty[i] = (uint8_t)((db >> k) & 1u);}
		}
		TERRA_WAVE_SYNC();
		// the rare sources, block by block with one lane per cell: cells this trace wrote back to its own pages earlier (written in THIS kernel: read through to L2) and blocks
		// with more lower versions than the candidate list holds (their writer lists are walked); only cells that ENTER the window -- the others hold what the droplet left there
		for (uint32_t m = back.special_blocks(); m; m &= m - 1) {
			uint32_t const bi = (uint32_t)__builtin_ctz(m);
			TERRA_LANES(c, SPEC_PAGE) {
				int X, Z;
				back.special_cell(bi, (uint32_t)c, X, Z);
				bool const in_new = (unsigned)(X - nx0) < (unsigned)EW && (unsigned)(Z - nz0) < (unsigned)EW && X < NX && Z < NY;
				bool const in_old = had && (unsigned)(X - owx0) < (unsigned)EW && (unsigned)(Z - owz0) < (unsigned)EW;
				if (in_new && !in_old) {int const o = (Z - nz0)*EW + (X - nx0); win[o] = back.special_value(bi, (uint32_t)c, X, Z, win[o]);}
			}
		}
		if (back.special_blocks()) {TERRA_WAVE_SYNC();}
		clk_sh_load += TERRA_CLOCK() - clk3; // (diagnostics: the fill and the rare sources)
		wx0 = nx0; wz0 = nz0; have = true;
		TERRA_WAVE_SYNC();
		unsigned long long const clk4 = TERRA_CLOCK();
		clk_shift += clk4 - clk0; clk_sh_flush += clk1 - clk0; clk_sh_prep += clk2 - clk1;
	}
	TERRA_HD bool begin_step(int xi, int zi) {
		xi = sati(xi, NX); zi = sati(zi, NY);
		if (!back.begin_step(xi, zi)) return false;
		int const bx0 = clampi(xi-1, NX-1), bx1 = clampi(xi+2, NX-1), bz0 = clampi(zi-1, NY-1), bz1 = clampi(zi+2, NY-1);
		if (TERRA_UNLIKELY(!(have && bx0 >= wx0 && bx1 < wx0 + EW && bz0 >= wz0 && bz1 < wz0 + EW))) {recenter((bx0 + bx1)/2, (bz0 + bz1)/2);}
		return !back.failed();
	}
	TERRA_HD float read_any(int X, int Z) { // outside the window only after a NaN position (index INT_MIN clamps to 0): slow path, still part of the footprint
		if (TERRA_LIKELY(in_window(X, Z))) return *cell(X, Z);
		back.note_far_read(X, Z);
		float const b = back.base(X, Z);
		return back.lookup(X, Z, b);
	}
	TERRA_HD void corners(int x, int z, float out[4]) {
		int const x0 = clampi(x, NX-1), x1 = clampi(x+1, NX-1), z0 = clampi(z, NY-1), z1 = clampi(z+1, NY-1);
		out[0] = read_any(x0, z0); out[1] = read_any(x1, z0); out[2] = read_any(x0, z1); out[3] = read_any(x1, z1);
	}
	TERRA_HD void deposit(int xi, int zi, float xf, float zf, float dse) {back.note_write(); this->deposit_cells(xi, zi, xf, zf, dse, NX, NY);}
	TERRA_HD void erode(int xi, int zi, float xp, float zp, float dse) {back.note_write(); this->erode_cells(xi, zi, xp, zp, dse, NX, NY);}
	// hot step: interior brush box, footprint recorded, box inside the resident window
	TERRA_HD bool hot_ready(int xi, int zi) {
		if (!(xi-1 >= 0 && zi-1 >= 0 && xi+2 <= NX-1 && zi+2 <= NY-1)) return false;
		if (!back.begin_step(xi, zi)) return false; // idempotent: the general step may record the same blocks again
		wx0 = wave_uniform(wx0); wz0 = wave_uniform(wz0); steps_in_window = wave_uniform(steps_in_window) + 1;
		if (!(have & ((unsigned)(xi-1 - wx0) <= (unsigned)(EW-4)) & ((unsigned)(zi-1 - wz0) <= (unsigned)(EW-4)))) return false; // xi-1 >= wx0 && xi+2 < wx0 + EW, the same in z
		this->box_load_all(xi, zi);
#if defined(__HIP_DEVICE_COMPILE__)
		boxdp = dirty + (this->boxp[0] - win); boxd = *boxdp; // the lane's dirty flag travels with its cell: mark_if is an OR in a register and an unconditional byte store
#endif
		return true;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	uint8_t *boxdp; uint8_t boxd;
	TERRA_HD void mark_if(int, int, bool c) {boxd = (uint8_t)(boxd | (c ? 1 : 0)); *boxdp = boxd;}
#endif
	TERRA_HD void corners_hot(int ox, int oz, float out[4]) const {this->box_corners(ox, oz, out);} // (inside the step's box, which lies inside the window)
	TERRA_HD void deposit_hot(int xi, int zi, float xf, float zf, float dse) {back.note_write(); this->deposit_cells_hot_all(xi, zi, xf, zf, dse);}
	TERRA_HD void erode_hot(int xi, int zi, float xp, float zp, float dse) {back.note_write(); this->erode_cells_hot_all(xi, zi, xp, zp, dse);}
	TERRA_HD void finish() {flush();}
};

// does the EW x EW window at (nx0, nz0) lie inside the caller's array (no cell of the pad ring)?  Then cell (X, Z) is interior[(Z - PAD)*xsize + (X - PAD)]
TERRA_HD bool grid_interior_window(grid_view_t const &g, int nx0, int nz0) {
	return g.border != nullptr && nx0 >= EROSION_PAD && nz0 >= EROSION_PAD && nx0 + EW <= EROSION_PAD + g.xsize && nz0 + EW <= EROSION_PAD + g.ysize;
}
// backing store = the grid itself (serial fall-back droplet: it is the lowest uncommitted droplet, nothing to speculate about)
struct grid_back_t {
	grid_view_t g;
	uint32_t *touched; uint32_t *touched_count; uint32_t touched_cap; // optional record of written cells
	TERRA_HD bool begin_step(int, int) {return true;}
	TERRA_HD bool failed() const {return false;}
	TERRA_HD void prepare_window(int, int) {}
	TERRA_HD void note_far_read(int, int) {}
	TERRA_HD void note_write() {}
	TERRA_HD void note_written_rect(int, int) {}
	TERRA_HD float base(int X, int Z) const {return *g.at(X, Z);}
	TERRA_HD float lookup(int, int, float b) const {return b;}
	TERRA_HD uint32_t source(int, int, bool &slow, bool &own) const {slow = false; own = false; return SPEC_SRC_GRID;}
	TERRA_HD float const *source_ptr(uint32_t, int X, int Z) const {return g.at(X, Z);}
	// what a window move needs to find the sources of its entering cells, as plain values (see spec_back_t::src_t)
	struct src_t {
		grid_view_t g;
		TERRA_HD bool interior_window(int nx0, int nz0) const {return grid_interior_window(g, nx0, nz0);}
		TERRA_HD float const *interior_ptr(int X, int Z) const {return g.interior + (size_t)(Z - EROSION_PAD)*g.xsize + (X - EROSION_PAD);}
		TERRA_HD size_t row_step() const {return (size_t)2*g.xsize;}
		TERRA_HD uint32_t code(int, int) const {return SPEC_SRC_GRID;}
		TERRA_HD float const *ptr(uint32_t, float const *gp) const {return gp;}
	};
	TERRA_HD src_t src_snapshot() const {return src_t{g};}
	TERRA_HD uint32_t special_blocks() const {return 0u;}
	TERRA_HD void special_cell(uint32_t, uint32_t, int &X, int &Z) const {X = Z = 0;}
	TERRA_HD float special_value(uint32_t, uint32_t, int, int, float v) const {return v;}
	TERRA_HD void store(int X, int Z, float v) {
		*g.at(X, Z) = v;
		if (touched) {uint32_t const k = TERRA_ATOMIC_ADD(touched_count, 1u); if (k < touched_cap) {touched[k] = (uint32_t)Z*(uint32_t)g.NX + (uint32_t)X;}}
	}
};

// ------------------------------------------------------------------ speculative (multi-version) backing store
constexpr uint32_t SPEC_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t SPEC_NIL   = 0xFFFFFFFFu;
enum {SPEC_F_LOG_OVERFLOW = 1, SPEC_F_BLK_OVERFLOW = 2, SPEC_F_NAN = 4, SPEC_F_UNDO_OVERFLOW = 8};
constexpr uint32_t SPEC_BLK_WRITTEN = 0x80000000u; // block-list entry flag: the trace has WRITTEN cells of the block (else it only read them)
constexpr uint32_t SPEC_BLK_CHANGED = 0x40000000u; // ... and what it wrote there differs from what the droplet's previously published version wrote there (or there is no such version)
constexpr uint32_t SPEC_BLK_NOW     = 0x20000000u; // ... the trace wrote cells of the block back during its LATEST slice (set for every entry when the trace stops; a version that is visible while it grows changes exactly there)
constexpr uint32_t SPEC_BLK_ID      = 0x1FFFFFFFu; // the block number
constexpr uint32_t SPEC_VIS_NONE    = 2u;          // spec_buffers_t::vbuf: the slot has no version a higher droplet may read
// life of a ring slot: FRESH (trace from the spawn) -> RUNNING (trace suspended at a step boundary, state saved) -> DONE_NEW (finished in
// this round, not published yet) -> IDLE (its finished version is published and believed valid); FAILED = the trace overflowed its log or
// block list and waits to become the lowest uncommitted droplet, which then runs alone directly on the grid.
enum {SPEC_IDLE = 0, SPEC_FRESH = 1, SPEC_RUNNING = 2, SPEC_DONE_NEW = 3, SPEC_FAILED = 4};

struct alignas(16) spec_u32x4 {uint32_t x, y, z, w;};

struct spec_ctl_t { // device-resident control block: a round needs no host decision
	uint32_t base;         // lowest uncommitted droplet
	uint32_t new_base;     // commit scan: lowest droplet that is not committable
	uint32_t stop_at;      // lowest FAILED droplet (SPEC_NIL: none); droplets above it are paused
	uint32_t new_stop;
	uint32_t unfinished;   // slots that are not IDLE after the round
	uint32_t traces;       // traces started (first traces + restarts)
	uint32_t nan_droplets; // committed droplets that went NaN
	uint32_t touched;      // cells recorded for the sparse clamp (may exceed the capacity)
	uint32_t fb_steps, fb_nan; // fall-back droplet
	uint32_t ndirty;       // entries of dirty_list
	uint32_t nd2[2], par;  // entries of dirty_list2[]; this round's resume pass appends to dirty_list2[par], its scan pass resets the marks of dirty_list2[par ^ 1] (made a round ago)
	uint32_t round_max_steps, round_max_shifts; // most steps / window moves of one trace in this round (diagnostics)
	uint32_t ck_resumes, pad4_;                 // re-traces that resumed from a checkpoint
	unsigned long long ck_steps_saved;          // steps those did not have to repeat
	uint32_t rounds, retraces_same;             // rounds that had work to do (the host launches them in batches and may overshoot the end); re-traces that reproduced the published version
	unsigned long long traced_steps, steps; // steps simulated (restarts included) / steps of committed droplets
	unsigned long long n_shift;                        // window moves summed over all traces (diagnostics)
	unsigned long long crit_steps, crit_shifts;         // round_max_* summed over the rounds
	unsigned long long clk_wave, clk_init, clk_shift, clk_tail; // 10 ns ticks summed over all traces: whole wave body / before the first step / window moves / after the last step
	unsigned long long clk_sh_flush, clk_sh_prep, clk_sh_load; // parts of clk_shift: write-back of leaving cells / block flags / plain grid loads (the rest: look-ups + LDS fill)
	unsigned long long clk_crit, round_max_clk;          // longest wave body of a round, summed over the rounds
	unsigned long long round_max_pack;                   // the round's longest wave body: its ticks << 44 | ticks in window moves << 24 | ticks before + after the steps << 10 | steps/4
	unsigned long long crit_own_shift, crit_own_edge, crit_own_steps; // the packed fields summed over the rounds
	unsigned long long round_max_pack2, crit_own_flush, crit_own_load, crit_own_prep; // same key: ticks << 44 | write-back << 28 | plain loads << 14 | block flags
	uint32_t waves_done, pad5_;                         // waves of the round's last launch that have finished: the last one closes the round (spec_close_wave)
};
struct spec_resume_t {uint32_t nblk, flags, undo_n, nck;}; // spec_back_t state of a suspended trace (its masks and pages are in the version buffer)

struct spec_buffers_t {
	grid_view_t grid;
	erosion_consts_t ec;
	uint32_t num_iters;    // droplets of the whole run
	uint32_t W;            // ring slots; droplet `it` lives in slot it % W, in-flight droplets are [base, base + W)
	uint32_t diag;         // collect the device-clock breakdown (terra_erosion_report clk_* / crit_clk_*)
	uint32_t ck_steps, ck_max; // steps between checkpoints of a trace, most checkpoints per trace (0: no checkpoints)
	uint32_t live_partial; // a first trace is visible to higher droplets while it grows, slice by slice (0: versions appear when their trace has finished; experiment knob, results never depend on it)
	uint32_t near_count;   // the first near_count in-flight droplets (the next to commit) trace without a step budget; the others are sliced (0: all sliced)
	uint32_t maxb;         // block-list capacity per droplet (<= SPEC_MAXB)
	uint32_t bshift;       // block edge = 1 << bshift cells (3: a page holds the 8 x 8 cells of a block)
	uint32_t nbx, nby;     // blocks per row / column of the padded grid
	// a version = what one trace of a droplet wrote, stored by block: entry e of the block list names the block, page e holds its 64 cells, mask e says which
	// of them were written.  No hashing and no read-modify-write on the way in (a write-back is one plain store), and a reader that found the writer of a
	// block through the block lists reads the cell directly.
	float    *page_vals[2];// [W][maxb][64]
	unsigned long long *page_mask[2]; // [W][maxb], valid for entries < blk_cnt (finished version) / run_nblk (suspended trace)
	uint32_t *blk_list[2]; // [W][maxb] distinct blocks, in the order the trace met them
	uint32_t *blk_cnt[2];  // [W]
	// checkpoints of the trace in each buffer (see SPEC_CK_STEPS) and its undo log: (float index into the slot's pages, value it held) of every write-back that
	// changed a cell written back before, in order -- rolling the pages back to a checkpoint = its masks + the log entries after it, newest first
	droplet_state_t *ck_state[2];       // [W][SPEC_CK_MAX]
	uint32_t *ck_nblk[2], *ck_undo[2];  // [W][SPEC_CK_MAX] footprint length / undo-log length at the checkpoint
	unsigned long long *ck_masks[2];    // [W][SPEC_CK_MAX][maxb]
	uint32_t *ck_cnt[2];                // [W] checkpoints of the buffer's trace (0: none usable)
	uint32_t *undo_idx[2]; float *undo_val[2]; // [W][SPEC_UNDO_MAX]
	uint32_t *undo_n[2];                // [W] log length when the trace stopped
	uint32_t *linked;      // [W] entries of the slot that may be linked into the block -> writers lists (set by the link pass: bounds the next unlink pass)
	uint32_t *rentry;      // [W] lowest footprint entry of the slot's trace that a changed lower version touched this round (mark pass; SPEC_NIL: none)
	uint32_t *rsrc;        // [W] pending re-trace: 0 = from the spawn, 1 = may resume from a checkpoint of the published version, 2 = of the suspended trace
	uint32_t *rat;         // [W] ... whose footprint is valid below this entry
	uint32_t *it;          // [W] droplet number held by the slot (SPEC_NIL: none)
	uint32_t *phase;       // [W]
	uint32_t *has_ver;     // [W] buffer cur[] holds a published finished version (visible to higher droplets)
	uint32_t *vbuf;        // [W] the buffer higher droplets read this slot's version from: cur[] when has_ver, else 1 - cur[] while a first trace is suspended (its pages so far: the version
	                       //     is visible while it grows), else SPEC_VIS_NONE.  Changes only between trace passes (spec_flip_body, spec_resume_wave, re-assignment)
	uint32_t *cur;         // [W] which buffer holds the published version; a (re)trace builds the other one
	uint32_t *changed;     // [W] the version finished this round differs from the published one
	uint32_t *restart;     // [W] set by the mark pass
	uint32_t *run_nblk;    // [W] blocks recorded so far by the suspended / failed trace
	uint32_t *flags;       // [W]
	uint32_t *nsteps;      // [W]
	droplet_state_t *state;// [W] suspended traces
	spec_resume_t *resume; // [W]
	uint32_t *head;        // [nbx*nby] block -> first node
	spec_u32x4 *node_rec;  // [W*maxb]  node id = slot*maxb + entry -> {next node, the droplet, its write mask of the block (2 words)} as of the link pass: a list walk is ONE
	                       //           16-byte load per node (+ vbuf[] of its slot, independent of it) instead of five loads that wait for one another (profiles/r04_erosion_recenter_isa.txt)
	uint32_t *dirty_min;   // [nbx*nby] lowest droplet whose published version changed in a way that touches the block, this round
	uint32_t *node_blk;    // [W*maxb]  block a node is currently linked under (SPEC_NIL: not linked): head[] is reset through it, not by an O(grid) fill
	uint32_t *dirty_list;  // [2*W*maxb] blocks whose dirty_min was lowered this round (duplicates allowed), ctl->ndirty entries
	uint32_t *dirty_list2[2]; // [W*maxb] each: blocks dirtied AFTER a round's mark pass (a growing version rolled back or dropped, spec_resume_wave) for the NEXT round's; ctl->nd2[], ctl->par
	uint32_t *touched;     // [touched_cap] padded-cell ids written to the grid (for the sparse final clamp)
	uint32_t touched_cap;
	uint32_t *done_cnt;    // [(W + 63)/64] waves of the round's last launch that have finished, per 64 slots (spec_close_wave); all zero between launches
	spec_ctl_t *ctl;
};

struct spec_back_t {
	spec_buffers_t const *sb;
	wave_shared_t *sh;     // LDS
	uint32_t slot, iter;
	float *my_pages; unsigned long long *my_masks; uint32_t *my_blks; // the version being built (buffer 1 - cur)
	uint32_t *my_undo_idx; float *my_undo_val;
	uint32_t nblk;
	bool log_undo = false; // the trace has a checkpoint: write-backs that change a cell written back before are logged
	bool blk_overflow;
	int wbx0, wbz0, wnb;   // window origin in blocks, blocks per window edge
	uint32_t nonempty = 0, special = 0; // wave_shared_t::blk_nonempty / blk_special of the prepared window, in registers
	int lx0 = INT_MIN, lx1 = INT_MIN, lz0 = INT_MIN, lz1 = INT_MIN; // block range of the previous step's brush box

	TERRA_HD static uint32_t map_hash(uint32_t b) {return (b*2654435761u) >> (32 - 9);}
	TERRA_HD static uint32_t page_cell(int X, int Z) {return (uint32_t)((Z & 7) << 3) | (uint32_t)(X & 7);}
	// entry of block b in this trace's list, or SPEC_NIL.  Inserts are done by lane 0 (map_add) and followed by a wave sync.
	TERRA_HD uint32_t map_find(uint32_t b) const {
		for (uint32_t h = map_hash(b), n = 0; n < SPEC_MAP_SLOTS; ++n, h = (h + 1) & (SPEC_MAP_SLOTS - 1)) {
			uint32_t const k = sh->map_keys[h];
			if (k == b) return sh->map_ent[h];
			if (k == SPEC_NIL) return SPEC_NIL;
		}
		return SPEC_NIL;
	}
	TERRA_HD void map_add(uint32_t b, uint32_t e) { // one lane; b is not in the map; at most SPEC_MAXB = SPEC_MAP_SLOTS/2 blocks: a free slot exists
		uint32_t h = map_hash(b);
		while (sh->map_keys[h] != SPEC_NIL) {h = (h + 1) & (SPEC_MAP_SLOTS - 1);}
		sh->map_keys[h] = b; sh->map_ent[h] = (uint8_t)e;
	}
	TERRA_HD void init(spec_buffers_t const *sb_, uint32_t slot_, uint32_t iter_, wave_shared_t *sh_, spec_resume_t const *rs) {
		sb = sb_; slot = slot_; iter = iter_; sh = sh_;
		uint32_t const nb = 1u - sb->cur[slot];
		my_pages = sb->page_vals[nb] + (size_t)slot*sb->maxb*SPEC_PAGE;
		my_masks = sb->page_mask[nb] + (size_t)slot*sb->maxb;
		my_blks  = sb->blk_list[nb] + (size_t)slot*sb->maxb;
		my_undo_idx = sb->undo_idx[nb] + (size_t)slot*SPEC_UNDO_MAX; my_undo_val = sb->undo_val[nb] + (size_t)slot*SPEC_UNDO_MAX;
		log_undo = rs ? (rs->nck != 0) : false;
		nblk = rs ? rs->nblk : 0;
		blk_overflow = false; wbx0 = wbz0 = 0; wnb = (EW >> sb->bshift) + 1;
		lx0 = lx1 = lz0 = lz1 = INT_MIN;
		if (TERRA_LANE0) {sh->flags = rs ? rs->flags : 0; sh->undo_n = rs ? rs->undo_n : 0; sh->chk = 0; sh->n_shift = 0;}
		TERRA_LANES(e, SPEC_MAXB) {sh->masks[e] = ((uint32_t)e < nblk) ? my_masks[e] : 0ull;} // a resumed trace: what it has written back so far (saved when it was suspended)
		TERRA_LANES(q, SPEC_MAXB/32) {sh->wrote[q] = 0u;}
		TERRA_WAVE_SYNC();
		rebuild_map(); // (empty for a new trace)
	}
	TERRA_HD void save(spec_resume_t &rs, uint32_t nck) const {rs.nblk = nblk; rs.flags = sh->flags; rs.undo_n = (sh->undo_n < SPEC_UNDO_MAX) ? sh->undo_n : SPEC_UNDO_MAX; rs.nck = nck;}
	TERRA_HD void rebuild_map() { // from my_blks[0 .. nblk): distinct blocks, every lane claims a free slot for each of its entries
		TERRA_LANES(h, SPEC_MAP_SLOTS) {sh->map_keys[h] = SPEC_NIL;}
		TERRA_WAVE_SYNC();
		TERRA_LANES(e, nblk) {
			uint32_t const b = my_blks[e] & SPEC_BLK_ID;
			for (uint32_t h = map_hash(b);; h = (h + 1) & (SPEC_MAP_SLOTS - 1)) {
				if (TERRA_ATOMIC_CAS(&sh->map_keys[h], SPEC_NIL, b) == SPEC_NIL) {sh->map_ent[h] = (uint8_t)e; break;}
			}
		}
		TERRA_WAVE_SYNC();
	}
	// checkpoint k of this trace (the window's dirty cells have just been written back): footprint length, masks, undo-log length, droplet state
	TERRA_HD void ck_save(uint32_t k, droplet_state_t const &d) {
		uint32_t const nb = 1u - sb->cur[slot];
		size_t const cbase = (size_t)slot*SPEC_CK_MAX + k;
		unsigned long long *ckm = sb->ck_masks[nb] + cbase*sb->maxb;
		TERRA_LANES(e, nblk) {ckm[e] = sh->masks[e];}
		if (TERRA_LANE0) {sb->ck_state[nb][cbase] = d; sb->ck_nblk[nb][cbase] = nblk; sb->ck_undo[nb][cbase] = (sh->undo_n < SPEC_UNDO_MAX) ? sh->undo_n : SPEC_UNDO_MAX;}
		TERRA_WAVE_SYNC();
	}
	// Start this trace from checkpoint k of the slot's trace in buffer `sbuf`: either the suspended trace in this trace's own buffer (rolled back in place) or the
	// published version in the other buffer (copied: readers keep using it until the new version is published).  Afterwards footprint, masks, pages, undo log and
	// the checkpoints 0 .. k of this trace are what they were when that checkpoint was taken; the LDS window is empty and is fetched afresh.
	TERRA_HD void ck_restore(uint32_t sbuf, uint32_t k, droplet_state_t &d, uint32_t had_nblk = 0) { // (no LDS: runs in the commit kernel's waves)
		uint32_t const nb = 1u - sb->cur[slot];
		bool const copy = (sbuf != nb);
		size_t const pbase = (size_t)slot*sb->maxb, cb0 = (size_t)slot*SPEC_CK_MAX, ub = (size_t)slot*SPEC_UNDO_MAX;
		float const *s_pages = sb->page_vals[sbuf] + pbase*SPEC_PAGE;
		uint32_t const *s_blks = sb->blk_list[sbuf] + pbase, *s_uidx = sb->undo_idx[sbuf] + ub;
		float const *s_uval = sb->undo_val[sbuf] + ub;
		uint32_t const nk = sb->ck_nblk[sbuf][cb0 + k], uk = sb->ck_undo[sbuf][cb0 + k], un = sb->undo_n[sbuf][slot];
		d = sb->ck_state[sbuf][cb0 + k];
		unsigned long long const *s_ckm = sb->ck_masks[sbuf] + (cb0 + k)*sb->maxb;
		TERRA_LANES(e, nk) {my_masks[e] = s_ckm[e];} // what a resumed trace starts from (spec_back_t::init)
		if (!copy) {TERRA_LANES(e, had_nblk) {if ((uint32_t)e >= nk) {my_masks[e] = 0ull;}}} // the entries after the checkpoint are gone: a reader that still finds their nodes in the writer lists sees no cell
		if (copy) {
			TERRA_LANES(e, nk) {my_blks[e] = s_blks[e] & SPEC_BLK_ID;}
			// every cell the trace had written by then: independent copies, eight pages of a lane's column at a time -- the eight reads are in flight together (one by one every store
			// waited for its own load: one memory latency per page of the footprint, the longest wave of the commit pass)
			TERRA_EACH_LANE(c) {
				for (uint32_t e0 = 0; e0 < nk; e0 += 8) {
					float v[8]; bool w[8];
#pragma unroll
					for (uint32_t u = 0; u < 8; ++u) {
						uint32_t const e = (e0 + u < nk) ? e0 + u : e0;
						w[u] = (e0 + u < nk) && ((s_ckm[e] >> (uint32_t)c) & 1ull);
						v[u] = s_pages[(size_t)e*SPEC_PAGE + (uint32_t)c];
					}
#pragma unroll
					for (uint32_t u = 0; u < 8; ++u) {if (w[u]) {my_pages[(size_t)(e0 + u)*SPEC_PAGE + (uint32_t)c] = v[u];}}
				}
			}
			{ // the masks of the checkpoints 0 .. k, four words of a lane at a time
				unsigned long long const *const sm = sb->ck_masks[sbuf] + cb0*sb->maxb; unsigned long long *const dm = sb->ck_masks[nb] + cb0*sb->maxb;
				uint32_t const total = (k + 1)*sb->maxb;
				TERRA_EACH_LANE(l) {
					for (uint32_t i0 = 0; i0 < total; i0 += 256) {
						unsigned long long v[4];
#pragma unroll
						for (uint32_t u = 0; u < 4; ++u) {uint32_t const i = i0 + u*64 + (uint32_t)l; v[u] = sm[(i < total) ? i : 0u];}
#pragma unroll
						for (uint32_t u = 0; u < 4; ++u) {uint32_t const i = i0 + u*64 + (uint32_t)l; if (i < total) {dm[i] = v[u];}}
					}
				}
			}
			TERRA_LANES(i, k + 1) {sb->ck_state[nb][cb0 + i] = sb->ck_state[sbuf][cb0 + i]; sb->ck_nblk[nb][cb0 + i] = sb->ck_nblk[sbuf][cb0 + i]; sb->ck_undo[nb][cb0 + i] = sb->ck_undo[sbuf][cb0 + i];}
			TERRA_LANES(q, uk) {my_undo_idx[q] = s_uidx[q]; my_undo_val[q] = s_uval[q];}
		}
		TERRA_WAVE_SYNC(); // the page copies are stored before the undo entries overwrite some of them
		// the log entries after the checkpoint, newest first: a cell rewritten since gets back what it held at the checkpoint (the oldest entry after it, applied last).
		// One lane per page walks the whole range and takes the entries of its page: their order is kept, the pages are independent.
		TERRA_LANES(e, nk) {
			unsigned long long const m = s_ckm[e];
			if (m) {
				for (uint32_t q = un; q-- > uk;) {
					uint32_t const idx = s_uidx[q];
					if (idx / SPEC_PAGE == (uint32_t)e && ((m >> (idx % SPEC_PAGE)) & 1ull)) {my_pages[idx] = s_uval[q];}
				}
			}
		}
		TERRA_WAVE_SYNC();
		nblk = nk;
	}
	// the trace stops (finished or suspended): masks to global memory, written flags into the block list
	TERRA_HD void publish_masks() const {
		TERRA_LANES(e, nblk) {
			unsigned long long const m = sh->masks[e];
			my_masks[e] = m;
			uint32_t const b = my_blks[e] & SPEC_BLK_ID;
			bool const now = ((sh->wrote[(uint32_t)e >> 5] >> ((uint32_t)e & 31u)) & 1u) != 0;
			my_blks[e] = (m ? (b | SPEC_BLK_WRITTEN) : b) | (now ? SPEC_BLK_NOW : 0u); // only written blocks can invalidate a reader
		}
		TERRA_WAVE_SYNC();
	}
	// sh->flags only changes inside window write-backs, which end with a wave sync; blk_overflow is a wave-uniform register
	TERRA_HD bool failed() const {return blk_overflow || (sh->flags & SPEC_F_LOG_OVERFLOW) != 0;}
	// Footprint bookkeeping (wave-uniform; lane 0 owns the writes): every block the trace reads or writes gets one list entry, and with it a page
	TERRA_HD void touch_block(uint32_t b) {
		if (map_find(b) != SPEC_NIL) return;
		if (TERRA_UNLIKELY(nblk >= sb->maxb)) {blk_overflow = true; return;}
		my_blks[nblk] = b; map_add(b, nblk); // (every lane does the same insert in lock step: same slots, same values -- no lane-dependent branch inside the hot step loop)
		++nblk;
		TERRA_WAVE_SYNC(); // the new map entry is visible to every lane
	}
	TERRA_HD void note_write() {}
	// reads outside the window (only after a NaN position): part of the footprint like any other block
	TERRA_HD void note_far_read(int X, int Z) {
		nblk = wave_uniform(nblk);
		touch_block((uint32_t)(Z >> sb->bshift)*sb->nbx + (uint32_t)(X >> sb->bshift));
	}
	TERRA_HD bool begin_step(int xi, int zi) { // footprint of one step = the 4x4 brush box, which also covers every read of that step
		int const x0 = clampi(xi-1, sb->ec.NX-1) >> sb->bshift, x1 = clampi(xi+2, sb->ec.NX-1) >> sb->bshift;
		int const z0 = clampi(zi-1, sb->ec.NY-1) >> sb->bshift, z1 = clampi(zi+2, sb->ec.NY-1) >> sb->bshift;
		// a droplet moves at most one cell per step: most steps have the brush box in the same (at most four) blocks as the step before -- nothing to record
		lx0 = wave_uniform(lx0); lx1 = wave_uniform(lx1); lz0 = wave_uniform(lz0); lz1 = wave_uniform(lz1);
		if (x0 == lx0 && x1 == lx1 && z0 == lz0 && z1 == lz1) {return !blk_overflow;} // (a write-back failure is noticed at the next window move / at the end)
		lx0 = x0; lx1 = x1; lz0 = z0; lz1 = z1;
		nblk = wave_uniform(nblk);
		touch_block((uint32_t)z0*sb->nbx + x0);
		if (x1 != x0) {touch_block((uint32_t)z0*sb->nbx + x1);}
		if (z1 != z0) {
			touch_block((uint32_t)z1*sb->nbx + x0);
			if (x1 != x0) {touch_block((uint32_t)z1*sb->nbx + x1);}
		}
		return !failed();
	}
	// a list node counts when its slot holds a published version of a LOWER droplet (nodes of slots that were re-assigned since the lists
	// were built have no published version yet)
	TERRA_HD bool lower_version(uint32_t j) const {return sb->vbuf[j] != SPEC_VIS_NONE && sb->it[j] < iter;}
	// which published LOWER versions wrote the blocks under the new window (only those blocks need the multi-version look-up): one lane per block walks the
	// block's writer list once and leaves the versions in LDS, highest droplet first, so that a cell's look-up is a mask test there plus one load
	TERRA_HD void prepare_window(int wx0, int wz0) {
		if (TERRA_LANE0) {sh->n_shift += 1; sh->blk_nonempty = 0; sh->blk_special = 0;} // (LDS executes a wave's instructions in order: the zeros are in place before the ORs below)
		wbx0 = wx0 >> sb->bshift; wbz0 = wz0 >> sb->bshift;
		TERRA_LANES(i, wnb*wnb) {
			uint32_t cnt = 0;
			uint32_t const bx = (uint32_t)(wbx0 + i % wnb), bz = (uint32_t)(wbz0 + i / wnb);
			if (bx < sb->nbx && bz < sb->nby) {
				spec_u32x4 const *const nrec = sb->node_rec; uint32_t const *const vbufs = sb->vbuf; uint32_t const maxb = sb->maxb;
				for (uint32_t node = sb->head[bz*sb->nbx + bx]; node != SPEC_NIL;) {
					uint32_t const this_node = node;
					spec_u32x4 const r = nrec[node]; uint32_t const cb = vbufs[node / maxb]; // (two independent loads: one memory latency per node)
					node = r.x;
					uint32_t const ij = r.y;
					if (!(cb != SPEC_VIS_NONE && ij < iter)) continue; // lower_version(): a slot that was re-assigned since the lists were built shows nothing
					if (cnt == SPEC_CAND) {cnt = SPEC_CAND_MANY; break;}
					spec_cand_t nc; nc.page = this_node | (cb << 31); nc.it = ij; nc.mask = (unsigned long long)r.z | ((unsigned long long)r.w << 32);
					uint32_t k = cnt; // insertion by descending droplet number
					for (; k > 0 && sh->cand[i][k-1].it < ij; --k) {sh->cand[i][k] = sh->cand[i][k-1];}
					sh->cand[i][k] = nc;
					++cnt;
				}
			}
			sh->blk_shared[i] = (uint8_t)cnt;
			uint32_t const oe = (bx < sb->nbx && bz < sb->nby) ? map_find(bz*sb->nbx + bx) : SPEC_NIL;
			sh->blk_own[i] = (uint16_t)((oe == SPEC_NIL) ? SPEC_OWN_NONE : oe);
			if (cnt != 0 || oe != SPEC_NIL) {TERRA_ATOMIC_OR(&sh->blk_nonempty, 1u << i);}
			if (cnt == SPEC_CAND_MANY || (oe != SPEC_NIL && sh->masks[oe] != 0ull)) {TERRA_ATOMIC_OR(&sh->blk_special, 1u << i);}
		}
		TERRA_WAVE_SYNC();
		// Resolve every cell of those blocks ONCE, all 64 cells of a block in parallel: which source a cell that enters the window is read from.  (Per entering cell this was a
		// hash probe for the trace's own page plus a walk over the candidates' masks: 16 cells per lane, each a chain of dependent LDS round trips -- 12 of the 19 us of a
		// window move on the critical path of a dense run, profiles/r03_erosion_clock_breakdown.txt.)  Now a cell's look-up is one byte.
		nonempty = wave_uniform(sh->blk_nonempty); special = wave_uniform(sh->blk_special);
		for (uint32_t m = nonempty; m; m &= m - 1) {
			uint32_t const i = (uint32_t)__builtin_ctz(m), cnt = sh->blk_shared[i];
			TERRA_LANES(c, SPEC_PAGE) {
				uint32_t code = SPEC_SRC_GRID; // (a crowded block: the grid value, patched through lookup(); a cell of this trace's own page: patched too -- special_value)
				if (cnt != SPEC_CAND_MANY) {
					uint32_t sel = 0;
					for (uint32_t k = cnt; k-- > 0;) {if ((sh->cand[i][k].mask >> c) & 1ull) {sel = k + 1;}} // the first (highest droplet) whose mask has the cell
					if (sel) {uint32_t const page = sh->cand[i][sel - 1].page; code = (page & 0x80000000u) | ((page & 0x7FFFFFFFu)*SPEC_PAGE + (uint32_t)c);}
				}
				sh->src[i][c] = code;
			}
		}
		TERRA_WAVE_SYNC();
	}
	TERRA_HD void note_written_rect(int, int) {}
	TERRA_HD float base(int X, int Z) const {return *sb->grid.at(X, Z);}
	TERRA_HD bool block_flag(int X, int Z) const { // is the cell's block also in a LOWER droplet's footprint?
		int const bx = (X >> sb->bshift) - wbx0, bz = (Z >> sb->bshift) - wbz0;
		if ((unsigned)bx < (unsigned)wnb && (unsigned)bz < (unsigned)wnb) return sh->blk_shared[bz*wnb + bx] != 0;
		return true; // outside the prepared window (far read): look the block up directly
	}
	TERRA_HD uint32_t block_of(int X, int Z) const {return (uint32_t)(Z >> sb->bshift)*sb->nbx + (uint32_t)(X >> sb->bshift);}
	TERRA_HD bool own_written(int X, int Z, uint32_t &e) const { // has this trace written the cell back to its page?
		e = map_find(block_of(X, Z));
		return e != SPEC_NIL && ((sh->masks[e] >> page_cell(X, Z)) & 1ull);
	}
	// What a window move needs to find where its entering cells are read from, as a SNAPSHOT of plain wave-uniform values taken once per move.  Through `sb` (a pointer to the
	// kernel's argument block) every field was a scalar load + s_waitcnt lgkmcnt(0) per use -- which also drains the LDS reads in flight -- and `sb->page_vals[code >> 31]` was a
	// VECTOR load of the pointer + s_waitcnt vmcnt(0) per cell, i.e. every cell waited for the previous cell's data load (tools/ab_build.sh instr1, TERRA_ERO_DIAG).
	struct src_t {
		grid_view_t g; float *pv0, *pv1; wave_shared_t const *sh;
		int wbx0, wbz0, wnb, bshift; uint32_t nonempty;
		TERRA_HD bool interior_window(int nx0, int nz0) const {return grid_interior_window(g, nx0, nz0);}
		TERRA_HD float const *interior_ptr(int X, int Z) const {return g.interior + (size_t)(Z - EROSION_PAD)*g.xsize + (X - EROSION_PAD);}
		TERRA_HD size_t row_step() const {return (size_t)2*g.xsize;}
		TERRA_HD uint32_t bi(int X, int Z) const {return (uint32_t)(((Z >> bshift) - wbz0)*wnb + ((X >> bshift) - wbx0));} // inside the prepared window by construction
		// SPEC_SRC_GRID, or the float index (bit 31: buffer) of the value in a lower version's page: one LDS word, read whether or not the block has versions (rows of blocks without
		// content are not filled in: the bit decides)
		TERRA_HD uint32_t code(int X, int Z) const {uint32_t const b = bi(X, Z), v = sh->src[b][page_cell(X, Z)]; return ((nonempty >> b) & 1u) ? v : SPEC_SRC_GRID;}
		TERRA_HD float const *ptr(uint32_t cd, float const *gp) const {
			float const *const pp = ((cd >> 31) ? pv1 : pv0) + (cd & 0x7FFFFFFFu);
			return (cd == SPEC_SRC_GRID) ? gp : pp;
		}
	};
	TERRA_HD src_t src_snapshot() const {
		src_t t;
		t.g = sb->grid; t.pv0 = sb->page_vals[0]; t.pv1 = sb->page_vals[1]; t.sh = sh;
		t.wbx0 = wbx0; t.wbz0 = wbz0; t.wnb = wnb; t.bshift = (int)sb->bshift; t.nonempty = nonempty;
		return t;
	}
	// the blocks under the prepared window whose entering cells are not (all) covered by src[]: one lane per cell of such a block
	TERRA_HD uint32_t special_blocks() const {return special;}
	TERRA_HD void special_cell(uint32_t bi, uint32_t c, int &X, int &Z) const {
		X = ((wbx0 + (int)(bi % (uint32_t)wnb)) << sb->bshift) + (int)(c & 7u); Z = ((wbz0 + (int)(bi / (uint32_t)wnb)) << sb->bshift) + (int)(c >> 3);
	}
	// v: what the pass put into the window for the cell (the grid value in a crowded block)
	TERRA_HD float special_value(uint32_t bi, uint32_t c, int X, int Z, float v) const {
		uint32_t const oe = sh->blk_own[bi];
		if (oe != SPEC_OWN_NONE && ((sh->masks[oe] >> c) & 1ull)) {return TERRA_L2_LOAD(&my_pages[(size_t)oe*SPEC_PAGE + c]);} // written earlier in THIS kernel by other lanes: not through L1
		if (sh->blk_shared[bi] == SPEC_CAND_MANY) {return lookup(X, Z, v);}
		return v;
	}
	// own earlier write-backs first, then the value written by the highest-numbered lower droplet, else the grid value `b`
	TERRA_HD float lookup(int X, int Z, float b) const {
		uint32_t const c = page_cell(X, Z);
		uint32_t e;
		if (own_written(X, Z, e)) {

			return TERRA_L2_LOAD(&my_pages[(size_t)e*SPEC_PAGE + c]);
		}
		if (block_flag(X, Z)) {

			uint32_t best = SPEC_NIL; // droplet number of the best writer so far
			float v = b;
			for (uint32_t node = sb->head[block_of(X, Z)]; node != SPEC_NIL;) {
				uint32_t const this_node = node;
				spec_u32x4 const r = sb->node_rec[node]; uint32_t const cb = sb->vbuf[node / sb->maxb];
				node = r.x;
				uint32_t const ij = r.y;
				if (!(cb != SPEC_VIS_NONE && ij < iter)) continue;
				if (best != SPEC_NIL && ij <= best) continue;
				unsigned long long const m = (unsigned long long)r.z | ((unsigned long long)r.w << 32);
				if ((m >> c) & 1ull) {best = ij; v = (cb ? sb->page_vals[1] : sb->page_vals[0])[(size_t)this_node*SPEC_PAGE + c];} // node = slot*maxb + entry = index of the page
			}
			return v;
		}
		return b;
	}
	// called from lanes in parallel, each with a distinct cell (several may share a page: the mask is updated atomically)
	TERRA_HD void store(int X, int Z, float val) {

		uint32_t const e = map_find(block_of(X, Z)), c = page_cell(X, Z);
		if (TERRA_UNLIKELY(e == SPEC_NIL)) {TERRA_ATOMIC_OR(&sh->flags, (uint32_t)SPEC_F_LOG_OVERFLOW); return;} // every written cell lies in a recorded brush box: never happens
		uint32_t const idx = e*SPEC_PAGE + c;
		if (log_undo && ((sh->masks[e] >> c) & 1ull)) { // the cell was written back before, possibly before a checkpoint: remember what it held
			float const old = TERRA_L2_LOAD(&my_pages[idx]);
			uint32_t ob, nbits; memcpy(&ob, &old, 4); memcpy(&nbits, &val, 4);
			if (ob != nbits) {
				uint32_t const q = TERRA_ATOMIC_ADD(&sh->undo_n, 1u);
				if (q < SPEC_UNDO_MAX) {my_undo_idx[q] = idx; my_undo_val[q] = old;}
				else {TERRA_ATOMIC_OR(&sh->flags, (uint32_t)SPEC_F_UNDO_OVERFLOW);}
			}
		}
		my_pages[idx] = val;
		TERRA_ATOMIC_OR(&sh->masks[e], 1ull << c);
		TERRA_ATOMIC_OR(&sh->wrote[e >> 5], 1u << (e & 31u));
	}
};

// ---- wave bodies: one call per droplet-wave (device: one 64-lane workgroup; host: one call)

// LDS scratch a wave body needs; the kernels / the emulator provide it
struct wave_scratch_t {float *win; uint8_t *dirty; wave_shared_t *sh;}; // win / dirty hold EW*EW entries

TERRA_HD bool spec_slot_active(spec_buffers_t const &sb, uint32_t slot, uint32_t &iter) { // holds a droplet that is not paused
	iter = sb.it[slot];
	return iter != SPEC_NIL && iter <= sb.ctl->stop_at;
}

// trace (or continue tracing) the slot's droplet for at most `budget` steps
TERRA_HD void spec_trace_wave(spec_buffers_t const &sb, uint32_t slot, uint32_t budget, wave_scratch_t const &ws) {
	uint32_t iter;
	if (!spec_slot_active(sb, slot, iter)) return;
	uint32_t const ph = sb.phase[slot];
	if (ph != SPEC_FRESH && ph != SPEC_RUNNING) return;
	unsigned long long const clk_a = TERRA_CLOCK();
	// the droplets next in line for the commit finish now (nothing behind them can be committed before they are); the ones further back advance a slice per
	// round, so that a long path has made most of its way by the time it is the one everybody waits for
	if (sb.near_count && iter - sb.ctl->base < sb.near_count) {budget = DROPLET_NO_BUDGET;}
	if ((uint64_t)sb.ctl->base + sb.W >= sb.num_iters) {budget = DROPLET_NO_BUDGET;} // nobody is waiting for a slot any more: run to the end
	window_mem_t<spec_back_t> mem;
	mem.init(ws.win, ws.dirty, sb.ec.NX, sb.ec.NY);
	mem.lead_mode = sb.ec.lead_mode;
	droplet_state_t d;
	bool finished = false;
	uint32_t nck = 0;            // checkpoints of this trace so far
	unsigned steps_before = 0;   // the droplet's step count when this wave took over
	if (ph == SPEC_FRESH) {
		mem.back.init(&sb, slot, iter, ws.sh, nullptr);
		finished = !droplet_start((int)iter, mem, sb.ec, d); // (a re-trace that can resume from a checkpoint was turned into a suspended trace by spec_resume_wave)
	}
	else {
		d = sb.state[slot];
		mem.back.init(&sb, slot, iter, ws.sh, &sb.resume[slot]);
		nck = sb.resume[slot].nck; steps_before = d.numMoves;
	}
	unsigned long long const clk_b = TERRA_CLOCK();
	{ // the trace runs from checkpoint to checkpoint; a checkpoint = all dirty cells of the window written back (the window stays) + the state saved
		unsigned used_total = 0;
		unsigned last_ck = nck ? sb.ck_state[1u - sb.cur[slot]][(size_t)slot*SPEC_CK_MAX + nck - 1].numMoves : 0u;
		bool ck_on = true;
		while (!finished) {
			unsigned seg = DROPLET_NO_BUDGET;
			if (ck_on && nck < sb.ck_max) {unsigned const since = d.numMoves - last_ck; seg = (since >= sb.ck_steps) ? 1u : sb.ck_steps - since;}
			if (budget != DROPLET_NO_BUDGET) {unsigned const left = budget - used_total; seg = (seg < left) ? seg : left;}
			unsigned const before = d.numMoves;
			finished = droplet_run_fast(d, mem, sb.ec, seg);
			used_total += d.numMoves - before;
			if (finished || mem.back.failed()) break;
			if (budget != DROPLET_NO_BUDGET && used_total >= budget) break; // suspended until the next round
			if (!(ck_on && nck < sb.ck_max) || d.numMoves - last_ck < sb.ck_steps) continue;
			mem.flush();
			if (ws.sh->flags & SPEC_F_UNDO_OVERFLOW) {ck_on = false; nck = 0; mem.back.log_undo = false;} // the log is incomplete: no checkpoint of this trace can be restored
			else {mem.back.ck_save(nck, d); ++nck; last_ck = d.numMoves; mem.back.log_undo = true;}
		}
		if (ws.sh->flags & SPEC_F_UNDO_OVERFLOW) {nck = 0;}
	}
	unsigned long long const clk_c = TERRA_CLOCK();
	mem.finish(); // the window's dirty cells go to the pages: a suspended trace keeps nothing but its masks in LDS, and those are saved next
	mem.back.publish_masks();
	if (TERRA_LANE0) {
		uint32_t const ob = sb.cur[slot], nb = 1u - ob;
		uint32_t const fl = ws.sh->flags | (mem.back.blk_overflow ? (uint32_t)SPEC_F_BLK_OVERFLOW : 0u);
		bool const failed = (fl & (SPEC_F_LOG_OVERFLOW | SPEC_F_BLK_OVERFLOW)) != 0;
		sb.run_nblk[slot] = mem.back.nblk;
		sb.ck_cnt[nb][slot] = failed ? 0u : nck;
		sb.undo_n[nb][slot] = (ws.sh->undo_n < SPEC_UNDO_MAX) ? ws.sh->undo_n : SPEC_UNDO_MAX;
		if (ph == SPEC_FRESH) {sb.rsrc[slot] = 0;}
		if (finished || failed) {
			sb.blk_cnt[nb][slot] = mem.back.nblk;
			sb.nsteps[slot]      = d.numMoves;
			sb.flags[slot]       = fl | (d.nan_seen ? SPEC_F_NAN : 0);
			sb.changed[slot]     = 0u; // (spec_post_wave compares the version with the published one, block by block)
			sb.phase[slot]       = failed ? (uint32_t)SPEC_FAILED : (uint32_t)SPEC_DONE_NEW;
		}
		else {
			sb.state[slot] = d;
			mem.back.save(sb.resume[slot], nck);
			sb.phase[slot] = SPEC_RUNNING;
		}
		if (ph == SPEC_FRESH) {TERRA_ATOMIC_ADD(&sb.ctl->traces, 1u);}
		TERRA_ATOMIC_ADD(&sb.ctl->traced_steps, (unsigned long long)(d.numMoves - steps_before));
		if (sb.diag) { // the serial chain and the device-clock breakdown of the traces (TERRA_ERO_DIAG=1): ~20 more atomics on the control block -- ONE cache line for the whole chip -- per trace
			TERRA_ATOMIC_MAX(&sb.ctl->round_max_steps, (uint32_t)(d.numMoves - steps_before));
			TERRA_ATOMIC_MAX(&sb.ctl->round_max_shifts, ws.sh->n_shift);
			if (ws.sh->n_shift) {TERRA_ATOMIC_ADD(&sb.ctl->n_shift, (unsigned long long)ws.sh->n_shift);}
			unsigned long long const clk_d = TERRA_CLOCK();
			TERRA_ATOMIC_ADD(&sb.ctl->clk_wave, clk_d - clk_a); TERRA_ATOMIC_ADD(&sb.ctl->clk_init, clk_b - clk_a);
			TERRA_ATOMIC_ADD(&sb.ctl->clk_shift, mem.clk_shift); TERRA_ATOMIC_ADD(&sb.ctl->clk_tail, clk_d - clk_c);
			TERRA_ATOMIC_ADD(&sb.ctl->clk_sh_flush, mem.clk_sh_flush); TERRA_ATOMIC_ADD(&sb.ctl->clk_sh_prep, mem.clk_sh_prep); TERRA_ATOMIC_ADD(&sb.ctl->clk_sh_load, mem.clk_sh_load);
			TERRA_ATOMIC_MAX(&sb.ctl->round_max_clk, clk_d - clk_a);
			auto sat = [](unsigned long long v, unsigned bits) {unsigned long long const m = (1ull << bits) - 1; return (v < m) ? v : m;};
			TERRA_ATOMIC_MAX(&sb.ctl->round_max_pack2, (sat(clk_d - clk_a, 20) << 44) | (sat(mem.clk_sh_flush, 16) << 28) | (sat(mem.clk_sh_load, 14) << 14) | sat(mem.clk_sh_prep, 14));
			TERRA_ATOMIC_MAX(&sb.ctl->round_max_pack, (sat(clk_d - clk_a, 20) << 44) | (sat(mem.clk_shift, 20) << 24) | (sat((clk_b - clk_a) + (clk_d - clk_c), 14) << 10) | sat((d.numMoves - steps_before) >> 2, 10));
		}
	}
}

// Before the traces of a round: a pending re-trace whose previous trace is still valid up to some checkpoint becomes a suspended trace at that checkpoint
// (state, footprint, masks, pages, undo log, checkpoints as they were then), which the trace kernel resumes like any other suspended trace.
TERRA_HD void spec_resume_wave(spec_buffers_t const &sb, uint32_t slot) {
	uint32_t iter;
	if (!spec_slot_active(sb, slot, iter)) return;
	if (sb.phase[slot] != SPEC_FRESH) return;
	uint32_t const rs = sb.rsrc[slot];
	// a first trace that higher droplets have been reading while it grew (vbuf) is about to be rolled back to a checkpoint, or dropped: what it showed changes NOW, after this
	// round's mark pass -- the blocks concerned are marked for the next round's (dirty_list2).  (A paused slot keeps what it shows until it is active again.)
	bool const vis = !sb.has_ver[slot] && sb.vbuf[slot] != SPEC_VIS_NONE;
	if (!rs && !vis) return;
	uint32_t const nb = 1u - sb.cur[slot];
	uint32_t const sbuf = (rs == 1u) ? sb.cur[slot] : nb, cnt = rs ? sb.ck_cnt[sbuf][slot] : 0u, upto = sb.rat[slot];
	uint32_t k = SPEC_NIL;
	for (uint32_t i = 0; i < cnt; ++i) {if (sb.ck_nblk[sbuf][(size_t)slot*SPEC_CK_MAX + i] <= upto) {k = i;}} // footprint lengths grow with the checkpoint number
	uint32_t const had = sb.run_nblk[slot];
	if (vis) {
		size_t const pbase = (size_t)slot*sb.maxb, cb0 = (size_t)slot*SPEC_CK_MAX;
		uint32_t const *bl = sb.blk_list[nb] + pbase;
		unsigned long long const *msk = sb.page_mask[nb] + pbase;
		bool const inplace = (k != SPEC_NIL); // (rs == 2: a slot without a published version has no other trace to resume from)
		uint32_t const nk = inplace ? sb.ck_nblk[nb][cb0 + k] : 0u, uk = inplace ? sb.ck_undo[nb][cb0 + k] : 0u, un = inplace ? sb.undo_n[nb][slot] : 0u;
		unsigned long long const *ckm = sb.ck_masks[nb] + (cb0 + (inplace ? k : 0u))*sb.maxb;
		uint32_t const *uidx = sb.undo_idx[nb] + (size_t)slot*SPEC_UNDO_MAX;
		bool const odd = (sb.ctl->par & 1u) != 0;
		uint32_t *const cnt2 = odd ? &sb.ctl->nd2[1] : &sb.ctl->nd2[0], *const lst2 = odd ? sb.dirty_list2[1] : sb.dirty_list2[0];
		for (uint32_t q0 = 0; q0 < had; q0 += 64) {
			TERRA_EACH_LANE(l) {
				uint32_t const e = q0 + (uint32_t)l;
				uint32_t b = SPEC_NIL;
				if (e < had) {
					uint32_t const ent = bl[e];
					if (ent & SPEC_BLK_WRITTEN) {
						bool ch = (e >= nk) || (msk[e] != ckm[e]); // gone, or cells written back since the checkpoint
						if (!ch) {for (uint32_t q = uk; q < un; ++q) {if (uidx[q]/SPEC_PAGE == e) {ch = true; break;}}} // or a cell that gets back what it held then
						if (ch) {b = ent & SPEC_BLK_ID;}
					}
				}
				uint32_t const kk = wave_reserve(cnt2, (b != SPEC_NIL) ? 1u : 0u);
				if (b != SPEC_NIL) {TERRA_ATOMIC_MIN(&sb.dirty_min[b], iter); lst2[kk] = b;}
			}
		}
		TERRA_WAVE_SYNC();
	}
	if (k != SPEC_NIL) {
		spec_back_t back;
		back.sb = &sb; back.sh = nullptr; back.slot = slot; back.iter = iter;
		back.my_pages = sb.page_vals[nb] + (size_t)slot*sb.maxb*SPEC_PAGE; back.my_masks = sb.page_mask[nb] + (size_t)slot*sb.maxb; back.my_blks = sb.blk_list[nb] + (size_t)slot*sb.maxb;
		back.my_undo_idx = sb.undo_idx[nb] + (size_t)slot*SPEC_UNDO_MAX; back.my_undo_val = sb.undo_val[nb] + (size_t)slot*SPEC_UNDO_MAX;
		droplet_state_t d;
		back.ck_restore(sbuf, k, d, had);
		if (TERRA_LANE0) {
			sb.state[slot] = d;
			spec_resume_t r; r.nblk = back.nblk; r.flags = 0; r.undo_n = sb.ck_undo[nb][(size_t)slot*SPEC_CK_MAX + k]; r.nck = k + 1;
			sb.resume[slot] = r;
			sb.run_nblk[slot] = back.nblk; sb.ck_cnt[nb][slot] = k + 1; sb.undo_n[nb][slot] = r.undo_n;
			sb.phase[slot] = SPEC_RUNNING;
			TERRA_ATOMIC_ADD(&sb.ctl->traces, 1u); TERRA_ATOMIC_ADD(&sb.ctl->ck_resumes, 1u); TERRA_ATOMIC_ADD(&sb.ctl->ck_steps_saved, (unsigned long long)d.numMoves);
		}
	}
	if (TERRA_LANE0) {
		sb.rsrc[slot] = 0;
		if (vis) {sb.vbuf[slot] = (k != SPEC_NIL) ? nb : SPEC_VIS_NONE;} // rolled back in place: still there, as it was at the checkpoint; else the next trace starts from the spawn
	}
}
// the lowest uncommitted droplet, alone, directly on the grid (overflow fall-back)
TERRA_HD void direct_droplet_wave(grid_view_t const &g, erosion_consts_t const &ec, uint32_t iter, uint32_t *out_steps_nan, wave_scratch_t const &ws,
	uint32_t *touched = nullptr, uint32_t *touched_count = nullptr, uint32_t touched_cap = 0)
{
	window_mem_t<grid_back_t> mem;
	mem.init(ws.win, ws.dirty, ec.NX, ec.NY);
	mem.lead_mode = ec.lead_mode;
	mem.back.g = g; mem.back.touched = touched; mem.back.touched_count = touched_count; mem.back.touched_cap = touched_cap;
	droplet_result_t const r = simulate_droplet((int)iter, mem, ec);
	mem.finish();
	if (TERRA_LANE0 && out_steps_nan) {out_steps_nan[0] = r.steps; out_steps_nan[1] = (uint32_t)r.nan_seen;}
}

// ================================================================== sparse regime: a few droplets on a big map (the headline: 1000 droplets on 16384^2)
// Two droplets of such a run almost never meet (one pair in ~5*10^5), so nearly every trace made on the ORIGINAL grid is already the serial loop's trace.  The machinery above
// pays for the dense regime on every wave -- candidate lists and per-cell source words (13 KB of LDS), version look-ups, checkpoints, an undo log: 22.7 KB of LDS and 264
// registers per droplet wave, which is what another heightmap's noise kernel has to share its CU with.  The sparse scheduler needs none of it:
//   trace    every droplet, one LEAN wave each, reads the grid only and writes to private pages (same page / mask / block-list format as a version above): 9.8 KB of LDS;
//            at its end the wave lowers wmin[b] to its droplet number for every block b it wrote;
//   check    droplet j is CONFLICTED when a block of its footprint was written by a lower droplet (wmin[b] < j); c = the lowest conflicted droplet;
//   commit   droplets [base, c) are exact -- by induction: no lower droplet wrote anything they read, and marks are never taken back, so that holds for committed droplets'
//            writes too -- and their written blocks are pairwise disjoint: their pages go to the grid in parallel;
//   re-trace droplet c is now the lowest uncommitted one: ONE wave traces it again on the grid as it stands (exact by definition; it is FINAL and never checked again),
//            marks what it wrote, and check / commit run again from base = c.
// One round per conflicted droplet: fine for a handful (the headline has one), hopeless for a dense run -- after SPARSE_MAX_CONFLICTS re-traces, or when a lean trace
// overflows its block list, the host hands the rest [base, N) to the multi-version scheduler above, which starts from the grid as committed so far.
struct lean_shared_t { // per-wave LDS of a lean trace, beside its window (4 KB) and dirty bytes (1 KB)
	uint32_t flags, special;                 // SPEC_F_*; bit i: block i under the window holds cells this trace has written back to its pages
	uint16_t blk_own[32];                    // per block under the window: its entry in this trace's block list (SPEC_OWN_NONE: not in the footprint)
	uint32_t map_keys[SPEC_MAP_SLOTS];       // block id (SPEC_NIL: free)
	uint8_t  map_ent[SPEC_MAP_SLOTS];
	unsigned long long masks[SPEC_MAXB];     // per entry: cells written back
};
struct lean_scratch_t {float *win; uint8_t *dirty; lean_shared_t *sh;};

enum {SPARSE_TRACED = 0, SPARSE_FINAL = 1, SPARSE_COMMITTED = 2, SPARSE_FAILED = 3};
struct sparse_ctl_t {
	uint32_t base;        // droplets below are committed
	uint32_t c;           // lowest conflicted droplet >= base as of the last check (N: none -- after the commit that follows, everything is on the grid)
	uint32_t nconf;       // conflicted droplets found by the last check
	uint32_t retraces;    // re-traces made so far
	uint32_t bail;        // the re-trace pass gave up: too many conflicts, or droplet `base` overflowed its block list (the host continues with the general scheduler from `base`)
	uint32_t touched;     // cells recorded for the sparse clamp
	uint32_t nan_droplets;
	uint32_t nwork;       // droplets the probe pass left for the trace waves (sparse_buffers_t::work)
	unsigned long long steps, traced_steps;
};
struct sparse_buffers_t {
	grid_view_t grid;
	erosion_consts_t ec;
	uint32_t N, maxb, nbx, nby, max_retraces;
	float *page_vals[2]; unsigned long long *page_mask[2]; uint32_t *blk_list[2], *blk_cnt[2]; // [N][maxb][64], [N][maxb], [N][maxb], [N]: a droplet's trace, and its re-trace in the other buffer
	uint32_t *cur, *state, *nsteps, *nan; // [N]
	uint32_t *work;       // [N] the droplets that take a step or write (probe pass) or were traced again: only these get trace / commit / unmark waves
	uint32_t *queued;     // [N] 1: the droplet is in work[]
	uint32_t trace_groups; // workgroups of the trace launch: group i takes work[i], work[i + trace_groups], ...
	uint32_t *wmin;       // [nbx*nby] lowest droplet that wrote the block in any of its traces (SPEC_NIL: nobody); reset through the block lists at the end of the run
	uint32_t *touched; uint32_t touched_cap;
	sparse_ctl_t *ctl;
	// a SHARDED run (one grid whose row strips live on several GPUs): every rank probes / traces the droplets that START in its rows [row0, row1) into its own
	// arena (shard = 1: the other droplets are skipped, nobody marks wmin[]); the eroding rank then gathers all traces into its arena, makes the marks and goes on as usual
	uint32_t shard, row0, row1;
};
// the interior row a droplet starts in (droplet_start's first two draws, src/erosion.cpp:67-70)
TERRA_HD uint32_t sparse_start_row(erosion_consts_t const &ec, uint32_t j) {
	rand_gen_t r; r.set_state((int)j + 11, 79*(int64_t)(int)j + 121);
	(void)(r.rand() % ec.xsize);
	return (uint32_t)(r.rand() % ec.ysize);
}

// backing store of a lean trace: reads = the grid (+ what this trace wrote back earlier), writes = private pages
struct lean_back_t {
	grid_view_t g; lean_shared_t *sh;
	float *my_pages; unsigned long long *my_masks; uint32_t *my_blks;
	uint32_t nblk, maxb, nbx, nby;
	int NXm1, NYm1;
	bool blk_overflow;
	int wbx0, wbz0;
	uint32_t special = 0;
	int lx0 = INT_MIN, lx1 = INT_MIN, lz0 = INT_MIN, lz1 = INT_MIN;
	static constexpr int wnb = (EW >> 3) + 1;

	TERRA_HD uint32_t map_find(uint32_t b) const {
		for (uint32_t h = spec_back_t::map_hash(b), n = 0; n < SPEC_MAP_SLOTS; ++n, h = (h + 1) & (SPEC_MAP_SLOTS - 1)) {
			uint32_t const k = sh->map_keys[h];
			if (k == b) return sh->map_ent[h];
			if (k == SPEC_NIL) return SPEC_NIL;
		}
		return SPEC_NIL;
	}
	TERRA_HD void init(sparse_buffers_t const &sb, uint32_t iter, uint32_t buf, lean_shared_t *sh_) {
		g = sb.grid; sh = sh_; maxb = sb.maxb; nbx = sb.nbx; nby = sb.nby; NXm1 = sb.ec.NX - 1; NYm1 = sb.ec.NY - 1;
		my_pages = sb.page_vals[buf] + (size_t)iter*sb.maxb*SPEC_PAGE; my_masks = sb.page_mask[buf] + (size_t)iter*sb.maxb; my_blks = sb.blk_list[buf] + (size_t)iter*sb.maxb;
		nblk = 0; blk_overflow = false; wbx0 = wbz0 = 0; special = 0;
		if (TERRA_LANE0) {sh->flags = 0; sh->special = 0;}
		TERRA_LANES(e, SPEC_MAXB) {sh->masks[e] = 0ull;}
		TERRA_LANES(h, SPEC_MAP_SLOTS) {sh->map_keys[h] = SPEC_NIL;}
		TERRA_WAVE_SYNC();
	}
	TERRA_HD bool failed() const {return blk_overflow || (sh->flags & SPEC_F_LOG_OVERFLOW) != 0;}
	TERRA_HD void touch_block(uint32_t b) {
		if (map_find(b) != SPEC_NIL) return;
		if (TERRA_UNLIKELY(nblk >= maxb)) {blk_overflow = true; return;}
		my_blks[nblk] = b; // (every lane does the same insert in lock step, as in spec_back_t::touch_block)
		uint32_t h = spec_back_t::map_hash(b);
		while (sh->map_keys[h] != SPEC_NIL) {h = (h + 1) & (SPEC_MAP_SLOTS - 1);}
		sh->map_keys[h] = b; sh->map_ent[h] = (uint8_t)nblk;
		++nblk;
		TERRA_WAVE_SYNC();
	}
	TERRA_HD void note_write() {}
	TERRA_HD void note_written_rect(int, int) {}
	TERRA_HD void note_far_read(int X, int Z) {nblk = wave_uniform(nblk); touch_block((uint32_t)(Z >> 3)*nbx + (uint32_t)(X >> 3));}
	TERRA_HD bool begin_step(int xi, int zi) { // the step's 4x4 brush box covers every read and write of the step
		int const x0 = clampi(xi-1, NXm1) >> 3, x1 = clampi(xi+2, NXm1) >> 3, z0 = clampi(zi-1, NYm1) >> 3, z1 = clampi(zi+2, NYm1) >> 3;
		lx0 = wave_uniform(lx0); lx1 = wave_uniform(lx1); lz0 = wave_uniform(lz0); lz1 = wave_uniform(lz1);
		if (x0 == lx0 && x1 == lx1 && z0 == lz0 && z1 == lz1) {return !blk_overflow;}
		lx0 = x0; lx1 = x1; lz0 = z0; lz1 = z1;
		nblk = wave_uniform(nblk);
		touch_block((uint32_t)z0*nbx + x0);
		if (x1 != x0) {touch_block((uint32_t)z0*nbx + x1);}
		if (z1 != z0) {
			touch_block((uint32_t)z1*nbx + x0);
			if (x1 != x0) {touch_block((uint32_t)z1*nbx + x1);}
		}
		return !failed();
	}
	// the only cells of a new window that do not come from the grid: the ones this trace wrote back to its own pages when they left an earlier window
	TERRA_HD void prepare_window(int wx0, int wz0) {
		if (TERRA_LANE0) {sh->special = 0;}
		wbx0 = wx0 >> 3; wbz0 = wz0 >> 3;
		TERRA_LANES(i, wnb*wnb) {
			uint32_t const bx = (uint32_t)(wbx0 + i % wnb), bz = (uint32_t)(wbz0 + i / wnb);
			uint32_t const oe = (bx < nbx && bz < nby) ? map_find(bz*nbx + bx) : SPEC_NIL;
			sh->blk_own[i] = (uint16_t)((oe == SPEC_NIL) ? SPEC_OWN_NONE : oe);
			if (oe != SPEC_NIL && sh->masks[oe] != 0ull) {TERRA_ATOMIC_OR(&sh->special, 1u << i);}
		}
		TERRA_WAVE_SYNC();
		special = wave_uniform(sh->special);
	}
	TERRA_HD float base(int X, int Z) const {return *g.at(X, Z);}
	TERRA_HD float lookup(int X, int Z, float b) const { // (a read outside the window: only after a NaN position)
		uint32_t const e = map_find((uint32_t)(Z >> 3)*nbx + (uint32_t)(X >> 3)), c = spec_back_t::page_cell(X, Z);
		if (e != SPEC_NIL && ((sh->masks[e] >> c) & 1ull)) {return TERRA_L2_LOAD(&my_pages[(size_t)e*SPEC_PAGE + c]);}
		return b;
	}
	typedef grid_back_t::src_t src_t;
	TERRA_HD src_t src_snapshot() const {return src_t{g};}
	TERRA_HD uint32_t special_blocks() const {return special;}
	TERRA_HD void special_cell(uint32_t bi, uint32_t c, int &X, int &Z) const {X = ((wbx0 + (int)(bi % (uint32_t)wnb)) << 3) + (int)(c & 7u); Z = ((wbz0 + (int)(bi / (uint32_t)wnb)) << 3) + (int)(c >> 3);}
	TERRA_HD float special_value(uint32_t bi, uint32_t c, int, int, float v) const {
		uint32_t const oe = sh->blk_own[bi];
		if (oe != SPEC_OWN_NONE && ((sh->masks[oe] >> c) & 1ull)) {return TERRA_L2_LOAD(&my_pages[(size_t)oe*SPEC_PAGE + c]);} // written earlier in THIS kernel by other lanes: not through L1
		return v;
	}
	TERRA_HD void store(int X, int Z, float val) { // lanes in parallel, distinct cells
		uint32_t const e = map_find((uint32_t)(Z >> 3)*nbx + (uint32_t)(X >> 3)), c = spec_back_t::page_cell(X, Z);
		if (TERRA_UNLIKELY(e == SPEC_NIL)) {TERRA_ATOMIC_OR(&sh->flags, (uint32_t)SPEC_F_LOG_OVERFLOW); return;} // every written cell lies in a recorded brush box: never happens
		my_pages[(size_t)e*SPEC_PAGE + c] = val;
		TERRA_ATOMIC_OR(&sh->masks[e], 1ull << c);
	}
};

// one lean trace of droplet `iter` into buffer `buf` (on the grid as it stands); marks what it wrote
TERRA_HD void sparse_trace_droplet(sparse_buffers_t const &sb, uint32_t iter, uint32_t buf, uint32_t new_state, lean_scratch_t const &ws) {
	window_mem_t<lean_back_t> mem;
	mem.init(ws.win, ws.dirty, sb.ec.NX, sb.ec.NY);
	mem.lead_mode = sb.ec.lead_mode;
	mem.back.init(sb, iter, buf, ws.sh);
	droplet_state_t d;
	if (droplet_start((int)iter, mem, sb.ec, d)) {droplet_run_fast(d, mem, sb.ec, DROPLET_NO_BUDGET);}
	mem.finish();
	bool const failed = mem.back.failed();
	uint32_t const n = mem.back.nblk;
	TERRA_LANES(e, n) {
		unsigned long long const m = ws.sh->masks[e];
		uint32_t const b = mem.back.my_blks[e] & SPEC_BLK_ID;
		mem.back.my_masks[e] = m;
		mem.back.my_blks[e] = m ? (b | SPEC_BLK_WRITTEN) : b;
		if (m && !failed && !sb.shard) {TERRA_ATOMIC_MIN(&sb.wmin[b], iter);}
	}
	if (TERRA_LANE0) {
		sb.blk_cnt[buf][iter] = n; sb.cur[iter] = buf;
		sb.nsteps[iter] = d.numMoves; sb.nan[iter] = (uint32_t)d.nan_seen;
		sb.state[iter] = failed ? (uint32_t)SPARSE_FAILED : new_state;
		TERRA_ATOMIC_ADD(&sb.ctl->traced_steps, (unsigned long long)d.numMoves);
	}
	TERRA_WAVE_SYNC();
}
// ---- round 0.  Most droplets of a map that is mostly ocean end at their first step: the cell they would move to lies under water (src/erosion.cpp:98) -- 849 of the
// headline's 1000.  A trace WAVE for such a droplet is all overhead (LDS set-up, a 4 KB window fetched for two corner reads) and, worse, takes the registers of a quarter
// of a CU from another heightmap's noise kernel while it lives (128 registers do not fit beside four 120-register waves on a SIMD: the CU runs three noise blocks instead of
// four).  So one THREAD per droplet first walks that first step directly on the grid, with the reference's own loop (droplet_start + one iteration of droplet_run): a droplet
// that ends there without having written anything is complete -- its footprint (the blocks of its 4x4 box: every cell it read) goes into its block list like a trace's --
// and only the others are queued for a trace wave.
struct probe_mem_t {
	grid_view_t g; uint32_t *blks; uint32_t nblk, nbx; int NXm1, NYm1; bool wrote;
	TERRA_HD void add(uint32_t b) {for (uint32_t i = 0; i < nblk; ++i) {if (blks[i] == b) return;} blks[nblk++] = b;} // (at most four blocks per box)
	TERRA_HD bool begin_step(int xi, int zi) {
		xi = sati(xi, NXm1 + 1); zi = sati(zi, NYm1 + 1);
		int const x0 = clampi(xi-1, NXm1) >> 3, x1 = clampi(xi+2, NXm1) >> 3, z0 = clampi(zi-1, NYm1) >> 3, z1 = clampi(zi+2, NYm1) >> 3;
		add((uint32_t)z0*nbx + x0); add((uint32_t)z0*nbx + x1); add((uint32_t)z1*nbx + x0); add((uint32_t)z1*nbx + x1);
		return true;
	}
	TERRA_HD void corners(int x, int z, float out[4]) const {
		int const x0 = clampi(x, NXm1), x1 = clampi(x+1, NXm1), z0 = clampi(z, NYm1), z1 = clampi(z+1, NYm1);
		out[0] = *g.at(x0, z0); out[1] = *g.at(x1, z0); out[2] = *g.at(x0, z1); out[3] = *g.at(x1, z1);
	}
	TERRA_HD void deposit(int, int, float, float, float) {wrote = true;} // the probe never changes anything: a droplet that would write is left to its trace wave
	TERRA_HD void erode(int, int, float, float, float) {wrote = true;}
};
TERRA_HD void sparse_probe_body(sparse_buffers_t const &sb, uint32_t j) {
	if (sb.shard) { // a tracer of a sharded run: only the droplets that start in its rows
		uint32_t const z = sparse_start_row(sb.ec, j);
		if (z < sb.row0 || z >= sb.row1) {sb.cur[j] = 0; sb.blk_cnt[0][j] = 0; sb.blk_cnt[1][j] = 0; sb.nsteps[j] = 0; sb.nan[j] = 0; sb.state[j] = SPARSE_TRACED; sb.queued[j] = 0; return;}
	}
	probe_mem_t m;
	m.g = sb.grid; m.blks = sb.blk_list[0] + (size_t)j*sb.maxb; m.nblk = 0; m.nbx = sb.nbx; m.NXm1 = sb.ec.NX - 1; m.NYm1 = sb.ec.NY - 1; m.wrote = false;
	droplet_state_t d;
	bool done = !droplet_start((int)j, m, sb.ec, d);
	if (!done) {done = droplet_run(d, m, sb.ec, 1u);} // the first iteration of the step loop; false: it made the step (the budget ended it)
	sb.cur[j] = 0; sb.blk_cnt[1][j] = 0;
	if (done && !m.wrote && d.numMoves == 0 && sb.maxb >= 4) { // complete: nothing written, only the box was read
		sb.blk_cnt[0][j] = m.nblk; sb.nsteps[j] = 0; sb.nan[j] = 0; sb.state[j] = SPARSE_TRACED;
	}
	else {sb.blk_cnt[0][j] = 0; sb.state[j] = SPARSE_TRACED; sb.queued[j] = 1; sb.work[TERRA_ATOMIC_ADD(&sb.ctl->nwork, 1u)] = j; return;}
	sb.queued[j] = 0;
}
// the queued droplets, one trace wave each (workgroup i of trace_groups takes every trace_groups-th)
TERRA_HD void sparse_trace_wave(sparse_buffers_t const &sb, uint32_t group, lean_scratch_t const &ws) {
	uint32_t const n = wave_uniform(sb.ctl->nwork);
	for (uint32_t k = group; k < n; k += sb.trace_groups) {sparse_trace_droplet(sb, wave_uniform(sb.work[k]), 0u, SPARSE_TRACED, ws);}
}
// ---- the eroding rank of a sharded run: one wave per droplet fetches the trace its owner made (block list, masks, the pages of written blocks, the droplet's counters)
// from the owner's arena -- every arena has the layout of this one, rank r's lies (r - self)*stride bytes from it in the mapped range -- makes the marks the trace did not
// make and puts the droplet on the work list.  Afterwards this arena is what a single context would hold after its own probe + trace passes (marks and work list are sets:
// their order does not matter), and check / commit / re-trace run unchanged.
struct sparse_rows_t {uint32_t end[16];}; // rank r owns the interior rows [end[r-1], end[r])
struct sparse_shard_t {int phase; uint32_t row0, row1; uint8_t *arena; uint32_t world, self; long long stride; sparse_rows_t rows;}; // one phase of a sharded run (terra_engine::sparse_erosion)
constexpr uint32_t SPARSE_SHARD_MAX_WORLD = 16;
template<class T> TERRA_HD T *sparse_peer(T *local, long long rel) {return (T *)((char *)local + rel);}
TERRA_HD void sparse_gather_wave(sparse_buffers_t const &sb, uint32_t j, sparse_rows_t const &rows, uint32_t world, uint32_t self, long long stride) {
	uint32_t const z = sparse_start_row(sb.ec, j);
	uint32_t owner = 0;
	while (owner + 1 < world && z >= rows.end[owner]) {++owner;}
	long long const rel = ((long long)owner - (long long)self)*stride;
	size_t const pbase = (size_t)j*sb.maxb;
	uint32_t const n = wave_uniform(*sparse_peer(&sb.blk_cnt[0][j], rel)), st = wave_uniform(*sparse_peer(&sb.state[j], rel)), q = wave_uniform(*sparse_peer(&sb.queued[j], rel));
	uint32_t const ns = wave_uniform(*sparse_peer(&sb.nsteps[j], rel));
	uint32_t const *sbl = sparse_peer(sb.blk_list[0] + pbase, rel);
	if (owner != self) {
		unsigned long long const *smk = sparse_peer(sb.page_mask[0] + pbase, rel);
		float const *spv = sparse_peer(sb.page_vals[0] + pbase*SPEC_PAGE, rel);
		TERRA_LANES(e, n) {sb.blk_list[0][pbase + e] = sbl[e]; sb.page_mask[0][pbase + e] = smk[e];}
		for (uint32_t e = 0; e < n; ++e) {
			if (!(wave_uniform(sbl[e]) & SPEC_BLK_WRITTEN)) continue; // (a page is read for the cells its mask names only)
			TERRA_LANES(c, SPEC_PAGE) {sb.page_vals[0][(pbase + e)*SPEC_PAGE + c] = spv[(size_t)e*SPEC_PAGE + c];}
		}
		if (TERRA_LANE0) {
			sb.blk_cnt[0][j] = n; sb.blk_cnt[1][j] = 0; sb.cur[j] = 0; sb.state[j] = st; sb.nsteps[j] = ns; sb.nan[j] = *sparse_peer(&sb.nan[j], rel); sb.queued[j] = q;
		}
	}
	if (st != (uint32_t)SPARSE_FAILED) {TERRA_LANES(e, n) {uint32_t const ent = sbl[e]; if (ent & SPEC_BLK_WRITTEN) {TERRA_ATOMIC_MIN(&sb.wmin[ent & SPEC_BLK_ID], j);}}}
	if (TERRA_LANE0 && q) {
		sb.work[TERRA_ATOMIC_ADD(&sb.ctl->nwork, 1u)] = j;
		TERRA_ATOMIC_ADD(&sb.ctl->traced_steps, (unsigned long long)ns);
	}
	TERRA_WAVE_SYNC();
}
// a later round (ONE wave): everything below the lowest conflicted droplet is committed; that droplet is traced again on the grid as it stands now
TERRA_HD void sparse_retrace_wave(sparse_buffers_t const &sb, lean_scratch_t const &ws) {
	sparse_ctl_t &c = *sb.ctl;
	uint32_t const base = wave_uniform(c.c), bail = wave_uniform(c.bail), done = wave_uniform(c.retraces), pending = wave_uniform(c.nconf);
	TERRA_WAVE_SYNC(); // (every lane has read the control block)
	if (bail) return;
	if (TERRA_LANE0) {c.base = base; c.c = sb.N; c.nconf = 0;}
	if (base >= sb.N) return;
	// one round per conflicted droplet: when more of them are waiting than the run may still re-trace, the multi-version scheduler is the better tool -- now, not after the last allowed round
	if (wave_uniform(sb.state[base]) == (uint32_t)SPARSE_FAILED || (uint64_t)done + pending > sb.max_retraces) {if (TERRA_LANE0) {c.bail = 1;} return;}
	if (TERRA_LANE0) {
		c.retraces = done + 1;
		if (!sb.queued[base]) {sb.queued[base] = 1; sb.work[c.nwork] = base; c.nwork = c.nwork + 1;} // (settled by the probe pass until now: from here on it has pages to commit and marks to reset)
	}
	sparse_trace_droplet(sb, base, 1u - wave_uniform(sb.cur[base]), SPARSE_FINAL, ws);
	if (wave_uniform(sb.state[base]) == (uint32_t)SPARSE_FAILED) {if (TERRA_LANE0) {c.bail = 1;}} // (its footprint grew past the block list on the changed grid)
}
// one WAVE per droplet, a lane per footprint entry: is it conflicted?  (One thread per droplet walked its list with two dependent loads per entry: ~45 us for the longest
// droplet of the headline run beside another map's noise kernel -- profiles/r05_timeline_sparse_v1.txt -- for a check that is a few hundred loads.)
TERRA_HD void sparse_check_wave(sparse_buffers_t const &sb, uint32_t j) {
	sparse_ctl_t &c = *sb.ctl;
	if (c.bail || j < c.base) return;
	uint32_t const st = sb.state[j];
	if (st == SPARSE_COMMITTED || st == SPARSE_FINAL) return;
	uint32_t const buf = sb.cur[j], n = (st == SPARSE_FAILED) ? 0u : sb.blk_cnt[buf][j];
	uint32_t const *bl = sb.blk_list[buf] + (size_t)j*sb.maxb;
	uint32_t hits[TERRA_LANE_SLOTS] = {};
	TERRA_EACH_LANE(l) {
		uint32_t h = 0;
		for (uint32_t e = (uint32_t)l; e < n; e += 64) {h |= (sb.wmin[bl[e] & SPEC_BLK_ID] < j) ? 1u : 0u;}
		hits[TERRA_LANE_SLOT(l)] = h;
	}
	bool hit = (st == SPARSE_FAILED);
#if defined(__HIP_DEVICE_COMPILE__)
	hit = hit || (__ballot(hits[0] != 0) != 0ull);
#else
	for (int l = 0; l < 64; ++l) {hit = hit || hits[l] != 0;}
#endif
	if (hit && TERRA_LANE0) {TERRA_ATOMIC_MIN(&c.c, j); TERRA_ATOMIC_ADD(&c.nconf, 1u);}
}
// one wave per droplet: the droplets [base, c) are exact and write disjoint blocks -- their pages go to the grid.  A lane takes a page: all of its 64 floats are loaded first
// (independent 16-byte loads: one memory latency), then the written ones are stored
TERRA_HD void sparse_commit_droplet(sparse_buffers_t const &sb, uint32_t j) {
	sparse_ctl_t &c = *sb.ctl;
	if (j < c.base || j >= c.c || sb.state[j] == SPARSE_COMMITTED) return;
	uint32_t const buf = sb.cur[j], n = sb.blk_cnt[buf][j];
	size_t const pbase = (size_t)j*sb.maxb;
	uint32_t const *bl = sb.blk_list[buf] + pbase;
	unsigned long long const *pm = sb.page_mask[buf] + pbase;
	float const *pv = sb.page_vals[buf] + pbase*SPEC_PAGE;
	// Sixty-four entries at a time: lane l fetches entry l and its mask (one memory latency for all of them) and the wave reserves their record slots with one atomic; then
	// the pages one after another, a lane per CELL -- entry, mask and slot come out of lane i's registers (v_readlane: no memory), the 64 values of a page are one coalesced
	// load, independent of the page before.  A handful of registers: the wave is dispatched beside a noise kernel's four waves per SIMD without waiting for one of them to
	// leave (the lane-per-page form, 75 registers, spent 40 of its 45 us beside a noise kernel waiting for slots; a first lane-per-cell form that loaded entry and mask page by
	// page took 77 us: four dependent latencies per page -- profiles/r05_timeline_sparse_v3.txt)
	for (uint32_t e0 = 0; e0 < n; e0 += 64) {
		uint32_t ent_l[TERRA_LANE_SLOTS], mlo_l[TERRA_LANE_SLOTS], mhi_l[TERRA_LANE_SLOTS], k_l[TERRA_LANE_SLOTS];
		TERRA_EACH_LANE(l) {
			uint32_t const e = e0 + (uint32_t)l;
			uint32_t ent = 0; unsigned long long m = 0;
			if (e < n) {ent = bl[e]; if (ent & SPEC_BLK_WRITTEN) {m = pm[e];}}
			ent_l[TERRA_LANE_SLOT(l)] = ent; mlo_l[TERRA_LANE_SLOT(l)] = (uint32_t)m; mhi_l[TERRA_LANE_SLOT(l)] = (uint32_t)(m >> 32);
			k_l[TERRA_LANE_SLOT(l)] = wave_reserve(&c.touched, sb.touched ? (uint32_t)__builtin_popcountll(m) : 0u);
		}
		uint32_t const cnt = (n - e0 < 64u) ? n - e0 : 64u;
		for (uint32_t i = 0; i < cnt; ++i) {
			uint32_t const ent = TERRA_READLANE_U32(ent_l, i), m_lo = TERRA_READLANE_U32(mlo_l, i), m_hi = TERRA_READLANE_U32(mhi_l, i), k0 = TERRA_READLANE_U32(k_l, i);
			unsigned long long const mu = (unsigned long long)m_lo | ((unsigned long long)m_hi << 32);
			if (!mu) continue;
			uint32_t const b = ent & SPEC_BLK_ID, bx = b % sb.nbx, bz = b / sb.nbx;
			float const *page = pv + (size_t)(e0 + i)*SPEC_PAGE;
			TERRA_LANES(cc, SPEC_PAGE) {
				if ((mu >> cc) & 1ull) {
					uint32_t const X = (bx << 3) + ((uint32_t)cc & 7u), Z = (bz << 3) + ((uint32_t)cc >> 3);
					*sb.grid.at_sel((int)X, (int)Z) = page[cc];
					if (sb.touched) {
						uint32_t const k = k0 + (uint32_t)__builtin_popcountll(mu & ((1ull << cc) - 1ull));
						if (k < sb.touched_cap) {sb.touched[k] = Z*(uint32_t)sb.ec.NX + X;}
					}
				}
			}
		}
	}
	TERRA_WAVE_SYNC();
	if (TERRA_LANE0) {
		sb.state[j] = SPARSE_COMMITTED;
		TERRA_ATOMIC_ADD(&c.steps, (unsigned long long)sb.nsteps[j]);
		if (sb.nan[j]) {TERRA_ATOMIC_ADD(&c.nan_droplets, 1u);}
	}
}
// only a droplet of the work list can have pages (the ones the probe pass settled wrote nothing): workgroup `group` of trace_groups takes every trace_groups-th of them
TERRA_HD void sparse_commit_wave(sparse_buffers_t const &sb, uint32_t group) {
	if (sb.ctl->bail) return;
	uint32_t const n = wave_uniform(sb.ctl->nwork);
	for (uint32_t k = group; k < n; k += sb.trace_groups) {sparse_commit_droplet(sb, wave_uniform(sb.work[k]));}
}
// sparse version of "clamp to min_zval" (src/erosion.cpp:158-162) when min_zval <= every untouched cell: one thread per recorded write
TERRA_HD void touched_clamp_body(grid_view_t const &g, uint32_t const *touched, uint32_t i, float min_zval) {
	uint32_t const cell = touched[i];
	int const X = (int)(cell % (uint32_t)g.NX), Z = (int)(cell / (uint32_t)g.NX);
	int const x = X - EROSION_PAD, z = Z - EROSION_PAD;
	if ((unsigned)x < (unsigned)g.xsize && (unsigned)z < (unsigned)g.ysize) {float *p = g.interior + (size_t)z*g.xsize + x; *p = max_std(min_zval, *p);} // idempotent: duplicates are harmless
}

// after a check + commit pair: is everything on the grid?  (c.c = N: the check found no conflicted droplet in [base, N), the commit that followed covered all of it)
TERRA_HD bool sparse_done(sparse_buffers_t const &sb) {return sb.ctl->c >= sb.N && !sb.ctl->bail;}
// end of the run (also before the general scheduler takes over): every mark is reset through the block lists, both traces of a re-traced droplet.  One wave per droplet, a lane
// per entry.  Launched right behind the rounds, before the host has read the control block: when the run is NOT complete yet (more rounds or the hand-over follow) and the
// launch is not the final one (`force`), it does nothing
TERRA_HD void sparse_unmark_wave(sparse_buffers_t const &sb, uint32_t group, bool force) {
	if (!force && !sparse_done(sb)) return;
	uint32_t const nw = wave_uniform(sb.ctl->nwork);
	for (uint32_t k = group; k < nw; k += sb.trace_groups) { // (only a droplet of the work list can have written anything)
		uint32_t const j = wave_uniform(sb.work[k]);
		for (uint32_t buf = 0; buf < 2; ++buf) {
			uint32_t const n = sb.blk_cnt[buf][j];
			uint32_t const *bl = sb.blk_list[buf] + (size_t)j*sb.maxb;
			TERRA_LANES(e, n) {uint32_t const ent = bl[e]; if (ent & SPEC_BLK_WRITTEN) {sb.wmin[ent & SPEC_BLK_ID] = SPEC_NIL;}}
		}
	}
}
// the sparse clamp (touched_clamp_body) over the cells the commits recorded, count taken on the device; thread i of nth.  Idempotent.
TERRA_HD void sparse_clamp_body(sparse_buffers_t const &sb, uint32_t i, uint32_t nth, float min_zval, bool force) {
	if (!sb.touched || (!force && !sparse_done(sb))) return;
	uint32_t const n = (sb.ctl->touched < sb.touched_cap) ? sb.ctl->touched : sb.touched_cap;
	for (uint32_t k = i; k < n; k += nth) {touched_clamp_body(sb.grid, sb.touched, k, min_zval);}
}

// ---- per-logical-thread bodies of the bookkeeping kernels (one round = clear, trace, post, flip, link, mark, scan, flush, admit, advance)

TERRA_HD void spec_undirty_body(spec_buffers_t const &sb, uint32_t i) {
	if (i < sb.ctl->ndirty) {sb.dirty_min[sb.dirty_list[i]] = SPEC_NIL;}
	// (the two lists are selected, not indexed: `sb.dirty_list2[par ^ 1]` -- a run-time index into an array inside the by-value kernel argument -- read a wrong pointer on gfx950
	// with this compiler and reset marks that had not been looked at yet; found by bisecting a GPU-only parity failure)
	bool const odd = (sb.ctl->par & 1u) != 0; // the marks the previous round's resume pass made for this round's mark pass: list par ^ 1
	uint32_t const n2 = odd ? sb.ctl->nd2[0] : sb.ctl->nd2[1];
	uint32_t const *l2 = odd ? sb.dirty_list2[0] : sb.dirty_list2[1];
	if (i < n2) {sb.dirty_min[l2[i]] = SPEC_NIL;}
}
// A version finished this round: which of its blocks differ in content (cells written, values) from the droplet's published version -- those, and the blocks
// the published version wrote and the new one does not even touch, are dirty for every higher droplet.  A re-trace usually repeats most of its predecessor's
// writes bit for bit (every block before the point where its inputs changed); a first version changes every block it wrote.  One wave per slot, no LDS.
// A droplet without a published finished version: its FIRST trace, visible to higher droplets while it grows (vbuf = the buffer it is built in) -- they read the pages
// it has written back as of the end of the previous round.  What changes under them in a round is exactly what the trace wrote back during this round's slice
// (SPEC_BLK_NOW, set by publish_masks), whether the trace stopped at a slice boundary or at its end; a trace that failed is withdrawn, with everything it had shown.
TERRA_HD void spec_post_growing(spec_buffers_t const &sb, uint32_t slot, uint32_t iter) {
	uint32_t const ph = sb.phase[slot], nb = 1u - sb.cur[slot];
	bool const live = sb.live_partial != 0, withdrawn = (ph == SPEC_FAILED && sb.vbuf[slot] != SPEC_VIS_NONE);
	if (!(ph == SPEC_DONE_NEW || (live && ph == SPEC_RUNNING) || withdrawn)) return;
	uint32_t const n = (ph == SPEC_DONE_NEW) ? sb.blk_cnt[nb][slot] : sb.run_nblk[slot];
	uint32_t *bl = sb.blk_list[nb] + (size_t)slot*sb.maxb;
	for (uint32_t q0 = 0; q0 < n; q0 += 64) {
		TERRA_EACH_LANE(l) {
			uint32_t const q = q0 + (uint32_t)l;
			uint32_t b = SPEC_NIL;
			if (q < n) {
				uint32_t const e = bl[q];
				bool const hit = (withdrawn || !live) ? ((e & SPEC_BLK_WRITTEN) != 0) : ((e & SPEC_BLK_NOW) != 0); // (not live: the version appears now, with everything it wrote)
				if (hit) {b = e & SPEC_BLK_ID;}
				if (ph == SPEC_DONE_NEW) {bl[q] = hit ? (e | SPEC_BLK_CHANGED) : (e & ~SPEC_BLK_CHANGED);}
			}
			uint32_t const k = wave_reserve(&sb.ctl->ndirty, (b != SPEC_NIL) ? 1u : 0u);
			if (b != SPEC_NIL) {TERRA_ATOMIC_MIN(&sb.dirty_min[b], iter); sb.dirty_list[k] = b; TERRA_ATOMIC_OR(&sb.changed[slot], 1u);}
		}
	}
	TERRA_WAVE_SYNC();
}
TERRA_HD void spec_post_wave(spec_buffers_t const &sb, uint32_t slot) {
	uint32_t const iter = sb.it[slot];
	if (iter == SPEC_NIL) return;
	if (!sb.has_ver[slot]) {spec_post_growing(sb, slot, iter); return;}
	if (sb.phase[slot] != SPEC_DONE_NEW) return;
	uint32_t const ob = sb.cur[slot], nb = 1u - ob;
	size_t const pbase = (size_t)slot*sb.maxb;
	uint32_t *nblkl = sb.blk_list[nb] + pbase, *oblk = sb.blk_list[ob] + pbase;
	unsigned long long const *nmask = sb.page_mask[nb] + pbase, *omask = sb.page_mask[ob] + pbase;
	float const *nval = sb.page_vals[nb] + pbase*SPEC_PAGE, *oval = sb.page_vals[ob] + pbase*SPEC_PAGE;
	uint32_t const n = sb.blk_cnt[nb][slot], no = sb.has_ver[slot] ? sb.blk_cnt[ob][slot] : 0u;
	TERRA_LANES(e, n) { // one lane per block of the new version (all loads of a lane's comparison are independent)
		unsigned long long const m = nmask[e];
		uint32_t const b = nblkl[e] & SPEC_BLK_ID;
		uint32_t oe = SPEC_NIL;
		for (uint32_t k = 0; k < no; ++k) {uint32_t const q = ((uint32_t)e + k) % no; if ((oblk[q] & SPEC_BLK_ID) == b) {oe = q; break;}} // same path => same position: found at k = 0
		unsigned long long const mo = (oe != SPEC_NIL) ? omask[oe] : 0ull;
		bool differs = (m != mo);
		if (!differs && m) {
			uint32_t acc = 0;
			for (uint32_t c = 0; c < SPEC_PAGE; ++c) {
				float const vn = nval[(size_t)e*SPEC_PAGE + c], vo = oval[(size_t)oe*SPEC_PAGE + c];
				uint32_t x, y; memcpy(&x, &vn, 4); memcpy(&y, &vo, 4);
				acc |= ((m >> c) & 1ull) ? (x ^ y) : 0u;
			}
			differs = (acc != 0);
		}
		if (differs) {nblkl[e] = nblkl[e] | SPEC_BLK_CHANGED;} // (each entry has one lane)
	}
	TERRA_LANES(q, no) { // blocks the published version wrote and the new one does not even touch
		uint32_t const ent = oblk[q];
		bool gone = false;
		if (omask[q]) {
			gone = true;
			uint32_t const b = ent & SPEC_BLK_ID;
			for (uint32_t k = 0; k < n; ++k) {if ((nblkl[(q + k) % n] & SPEC_BLK_ID) == b) {gone = false; break;}}
		}
		oblk[q] = gone ? (ent | SPEC_BLK_CHANGED) : (ent & ~SPEC_BLK_CHANGED);
	}
	TERRA_WAVE_SYNC();
	uint32_t const lim = (n > no) ? n : no;
	for (uint32_t q0 = 0; q0 < lim; q0 += 64) { // the dirty marks (the flags were set by other lanes of this wave: read them at L2); list slots for the whole wave with one atomic
		TERRA_EACH_LANE(l) {
			uint32_t const q = q0 + (uint32_t)l;
			uint32_t b0 = SPEC_NIL, b1 = SPEC_NIL;
			if (q < no) {uint32_t const e = TERRA_L2_LOAD(&oblk[q]); if (e & SPEC_BLK_CHANGED) {b0 = e & SPEC_BLK_ID;}}
			if (q < n)  {uint32_t const e = TERRA_L2_LOAD(&nblkl[q]); if (e & SPEC_BLK_CHANGED) {b1 = e & SPEC_BLK_ID;}}
			uint32_t const cnt = (b0 != SPEC_NIL ? 1u : 0u) + (b1 != SPEC_NIL ? 1u : 0u);
			uint32_t k = wave_reserve(&sb.ctl->ndirty, cnt);
			if (b0 != SPEC_NIL) {TERRA_ATOMIC_MIN(&sb.dirty_min[b0], iter); sb.dirty_list[k++] = b0;}
			if (b1 != SPEC_NIL) {TERRA_ATOMIC_MIN(&sb.dirty_min[b1], iter); sb.dirty_list[k++] = b1;}
			if (cnt) {TERRA_ATOMIC_OR(&sb.changed[slot], 1u);}
		}
	}
	TERRA_WAVE_SYNC();
	if (TERRA_LANE0 && sb.has_ver[slot] && TERRA_L2_LOAD(&sb.changed[slot]) == 0u) {TERRA_ATOMIC_ADD(&sb.ctl->retraces_same, 1u);}
}
// publish the versions finished this round
TERRA_HD void spec_flip_body(spec_buffers_t const &sb, uint32_t slot) {
	if (sb.it[slot] == SPEC_NIL) return;
	if (sb.phase[slot] == SPEC_DONE_NEW) {sb.cur[slot] = 1u - sb.cur[slot]; sb.has_ver[slot] = 1; sb.phase[slot] = SPEC_IDLE;}
	// what higher droplets read from this slot in the next trace pass: the published version, else the pages of a suspended first trace
	sb.vbuf[slot] = sb.has_ver[slot] ? sb.cur[slot] : ((sb.phase[slot] == SPEC_RUNNING && sb.live_partial && sb.run_nblk[slot] != 0) ? 1u - sb.cur[slot] : SPEC_VIS_NONE);
}
// entries of the version higher droplets may read (after spec_flip_body)
TERRA_HD uint32_t spec_visible_count(spec_buffers_t const &sb, uint32_t slot) {
	if (sb.it[slot] == SPEC_NIL || sb.vbuf[slot] == SPEC_VIS_NONE) return 0u;
	return sb.has_ver[slot] ? sb.blk_cnt[sb.cur[slot]][slot] : sb.run_nblk[slot];
}
// rebuild block -> writer lists from the published footprints (head[] was reset to SPEC_NIL before): one thread per (slot, entry)
TERRA_HD void spec_link_body(spec_buffers_t const &sb, uint32_t slot, uint32_t entry) {
	uint32_t const cb = sb.vbuf[slot];
	if (sb.it[slot] == SPEC_NIL || cb == SPEC_VIS_NONE) return;
	if (entry >= spec_visible_count(sb, slot)) return;
	uint32_t const e = sb.blk_list[cb][(size_t)slot*sb.maxb + entry];
	if (!(e & SPEC_BLK_WRITTEN)) return; // the lists answer "who wrote here": read-only entries stay out
	uint32_t const node = slot*sb.maxb + entry, b = e & SPEC_BLK_ID;
	sb.node_blk[node] = b;
	uint32_t const prev = TERRA_ATOMIC_EXCH(&sb.head[b], node);
	unsigned long long const m = (cb ? sb.page_mask[1] : sb.page_mask[0])[node];
	sb.node_rec[node] = spec_u32x4{prev, sb.it[slot], (uint32_t)m, (uint32_t)(m >> 32)};
}
// take the lists apart again (before they are rebuilt, and at the end of the run: head[] is left all-NIL for the next run)
TERRA_HD void spec_unlink_body(spec_buffers_t const &sb, uint32_t node) {
	uint32_t const b = sb.node_blk[node];
	if (b != SPEC_NIL) {sb.head[b] = SPEC_NIL; sb.node_blk[node] = SPEC_NIL;}
}
// who must start over: any droplet whose footprint (so far) contains a block dirtied by a lower droplet.  One thread per (slot, entry).
TERRA_HD void spec_mark_body(spec_buffers_t const &sb, uint32_t slot, uint32_t entry) {
	uint32_t const iter = sb.it[slot];
	if (iter == SPEC_NIL) return;
	uint32_t const ph = sb.phase[slot], cb = sb.cur[slot], rs = sb.rsrc[slot];
	uint32_t const *bl; uint32_t cnt;
	if (ph == SPEC_RUNNING || ph == SPEC_FAILED || (ph == SPEC_FRESH && rs == 2u)) {bl = sb.blk_list[1u - cb] + (size_t)slot*sb.maxb; cnt = sb.run_nblk[slot];} // (a pending re-trace that has not run yet -- the slot is
	else if ((ph == SPEC_IDLE || (ph == SPEC_FRESH && rs == 1u)) && sb.has_ver[slot]) {bl = sb.blk_list[cb] + (size_t)slot*sb.maxb; cnt = sb.blk_cnt[cb][slot];}  // paused -- still answers for the trace it wants to resume)
	else return;
	if (entry < cnt && sb.dirty_min[bl[entry] & SPEC_BLK_ID] < iter) {sb.restart[slot] = 1; TERRA_ATOMIC_MIN(&sb.rentry[slot], entry);} // entries are in first-touch order: the trace is valid below the lowest one hit
}
// apply the restarts, find the commit point and the lowest failed droplet: one thread per slot
TERRA_HD void spec_scan_body(spec_buffers_t const &sb, uint32_t slot) {
	uint32_t const iter = sb.it[slot];
	if (iter == SPEC_NIL) return;
	if (sb.restart[slot]) {
		uint32_t const hit = sb.rentry[slot], ph0 = sb.phase[slot];
		sb.restart[slot] = 0; sb.rentry[slot] = SPEC_NIL;
		if (ph0 == SPEC_FRESH) {if (hit < sb.rat[slot]) {sb.rat[slot] = hit;}} // a re-trace that is still waiting to run: its valid prefix got shorter
		else {sb.rsrc[slot] = (ph0 == SPEC_IDLE) ? 1u : ((ph0 == SPEC_RUNNING) ? 2u : 0u); sb.rat[slot] = hit; sb.phase[slot] = SPEC_FRESH;}
	}
	uint32_t const ph = sb.phase[slot];
	if (ph == SPEC_FAILED && iter < TERRA_L2_LOAD(&sb.ctl->new_stop)) {TERRA_ATOMIC_MIN(&sb.ctl->new_stop, iter);}
	if (!(ph == SPEC_IDLE && sb.has_ver[slot]) && iter < TERRA_L2_LOAD(&sb.ctl->new_base)) {TERRA_ATOMIC_MIN(&sb.ctl->new_base, iter);} // look first: thousands of slots fold into one word, and only the lowest few can lower it
}
// flush the committed droplets [base, new_base): the highest-numbered committed writer of a cell stores it.  One wave per slot; the
// flushed buffer is left empty for the slot's next droplet.
TERRA_HD void spec_flush_wave(spec_buffers_t const &sb, uint32_t slot) {
	uint32_t const iter = sb.it[slot], nbase = sb.ctl->new_base;
	if (iter == SPEC_NIL || iter >= nbase) return;
	uint32_t const cb = sb.cur[slot], n = sb.blk_cnt[cb][slot];
	size_t const pbase = (size_t)slot*sb.maxb;
	for (uint32_t e0 = 0; e0 < n; e0 += 64) { // one page per lane: the dependent loads of the ownership test (list walk, masks) of different pages overlap; the cell stores need no reply
		TERRA_EACH_LANE(l) {
			uint32_t const e = e0 + (uint32_t)l;
			unsigned long long mine = 0;
			uint32_t b = 0;
			if (e < n) {
				uint32_t const ent = sb.blk_list[cb][pbase + e];
				if (ent & SPEC_BLK_WRITTEN) {
					b = ent & SPEC_BLK_ID;
					mine = sb.page_mask[cb][pbase + e];
					for (uint32_t node = sb.head[b]; node != SPEC_NIL && mine;) { // a later committed droplet owns the final value of the cells it wrote
						spec_u32x4 const r = sb.node_rec[node]; // (written by this round's link pass: the droplet its slot holds now; a droplet below new_base is idle with its finished version published)
						node = r.x;
						if (r.y <= iter || r.y >= nbase) continue;
						mine &= ~((unsigned long long)r.z | ((unsigned long long)r.w << 32));
					}
				}
			}
			uint32_t k = wave_reserve(&sb.ctl->touched, sb.touched ? (uint32_t)__builtin_popcountll(mine) : 0u); // (the record of written cells for the sparse final clamp)
			uint32_t const bx = b % sb.nbx, bz = b / sb.nbx;
			float const *page = sb.page_vals[cb] + (pbase + e)*SPEC_PAGE;
			for (; mine; mine &= mine - 1, ++k) {
				uint32_t const c = (uint32_t)__builtin_ctzll(mine);
				uint32_t const X = (bx << 3) + (c & 7u), Z = (bz << 3) + (c >> 3);
				*sb.grid.at((int)X, (int)Z) = page[c];
				if (sb.touched && k < sb.touched_cap) {sb.touched[k] = Z*(uint32_t)sb.ec.NX + X;}
			}
		}
	}
}
// hand a slot to the next droplet of the ring
TERRA_HD void spec_reassign(spec_buffers_t const &sb, uint32_t slot, uint32_t iter) {
	uint64_t const nit = (uint64_t)iter + sb.W;
	sb.it[slot] = (nit < sb.num_iters) ? (uint32_t)nit : SPEC_NIL;
	sb.phase[slot] = SPEC_FRESH; sb.has_ver[slot] = 0; sb.vbuf[slot] = SPEC_VIS_NONE; sb.blk_cnt[0][slot] = 0; sb.blk_cnt[1][slot] = 0; sb.run_nblk[slot] = 0; sb.restart[slot] = 0;
	sb.rsrc[slot] = 0; sb.rentry[slot] = SPEC_NIL; sb.rat[slot] = SPEC_NIL; sb.ck_cnt[0][slot] = 0; sb.ck_cnt[1][slot] = 0;
}
// committed droplets leave, their slots are handed to the next droplets: one thread per slot
TERRA_HD void spec_admit_body(spec_buffers_t const &sb, uint32_t slot) {
	uint32_t const iter = sb.it[slot];
	if (iter == SPEC_NIL || iter >= sb.ctl->new_base) return;
	TERRA_ATOMIC_ADD(&sb.ctl->steps, (unsigned long long)sb.nsteps[slot]);
	if (sb.flags[slot] & SPEC_F_NAN) {TERRA_ATOMIC_ADD(&sb.ctl->nan_droplets, 1u);}
	spec_reassign(sb, slot, iter);
}
// end of round (one thread)
TERRA_HD void spec_advance_body(spec_buffers_t const &sb) {
	spec_ctl_t &c = *sb.ctl;
	if (c.base < sb.num_iters && c.stop_at != c.base) {++c.rounds;} // (a round after the end, or while the lowest droplet waits for its serial fall-back, is empty)
	c.base = c.new_base;
	uint64_t const nb = (uint64_t)c.base + sb.W;
	c.new_base = (nb < sb.num_iters) ? (uint32_t)nb : sb.num_iters;
	c.stop_at = c.new_stop; c.new_stop = SPEC_NIL; c.unfinished = 0; c.ndirty = 0;
	if (c.par & 1u) {c.nd2[0] = 0;} else {c.nd2[1] = 0;}
	c.par ^= 1u; // the marks of dirty_list2[par ^ 1] were consumed and reset in this round; the next round consumes the ones its resume pass has just made
	c.crit_steps += c.round_max_steps; c.crit_shifts += c.round_max_shifts; c.round_max_steps = 0; c.round_max_shifts = 0;
	c.clk_crit += c.round_max_clk; c.round_max_clk = 0;
	c.crit_own_shift += (c.round_max_pack >> 24) & 0xFFFFFu; c.crit_own_edge += (c.round_max_pack >> 10) & 0x3FFFu; c.crit_own_steps += (c.round_max_pack & 0x3FFu) << 2; c.round_max_pack = 0;
	c.crit_own_flush += (c.round_max_pack2 >> 28) & 0xFFFFu; c.crit_own_load += (c.round_max_pack2 >> 14) & 0x3FFFu; c.crit_own_prep += c.round_max_pack2 & 0x3FFFu; c.round_max_pack2 = 0;
}
// after the fall-back droplet `base` ran directly on the grid: it is committed and every other in-flight trace starts over (the grid
// changed under them without a version to compare against).  One thread per slot, then spec_fallback_advance_body.
TERRA_HD void spec_fallback_reset_body(spec_buffers_t const &sb, uint32_t slot) {
	uint32_t const iter = sb.it[slot];
	if (iter == SPEC_NIL) return;
	if (iter == sb.ctl->base) {spec_reassign(sb, slot, iter); return;}
	sb.phase[slot] = SPEC_FRESH; sb.has_ver[slot] = 0; sb.vbuf[slot] = SPEC_VIS_NONE; sb.blk_cnt[0][slot] = 0; sb.blk_cnt[1][slot] = 0; sb.run_nblk[slot] = 0; sb.restart[slot] = 0;
	sb.rsrc[slot] = 0; sb.rentry[slot] = SPEC_NIL; sb.rat[slot] = SPEC_NIL; sb.ck_cnt[0][slot] = 0; sb.ck_cnt[1][slot] = 0; // the grid changed under every trace: nothing to resume from
}
TERRA_HD void spec_fallback_advance_body(spec_buffers_t const &sb) {
	spec_ctl_t &c = *sb.ctl;
	c.base += 1;
	uint64_t const nb = (uint64_t)c.base + sb.W;
	c.new_base = (nb < sb.num_iters) ? (uint32_t)nb : sb.num_iters;
	c.stop_at = SPEC_NIL; c.new_stop = SPEC_NIL; c.unfinished = 0;
	c.steps += c.fb_steps; c.traced_steps += c.fb_steps; c.nan_droplets += c.fb_nan;
}
// ---- the bookkeeping launches of a round, fused where no other slot's pass has to lie in between (profiles/r06_erosion_round_anatomy.txt: every launch of a round costs
// 4-5 us whatever it does, and the five small ones did 1-3 us of work each):
//   after the traces: the slot's comparison with its published version (dirty marks), its nodes taken out of the writer lists, its finished version published -- post reads
//     nothing another slot's unlink or flip writes (its own lists, masks and pages only), so the three passes of a slot run back to back in the slot's wave;
//   at the end: the slot's commit, its checkpoint resume, its hand-over to the next droplet (all of them touch the slot's own arrays; other slots' commits read the writer
//     lists' node records, which keep the droplet numbers of the link pass) -- and the wave that finishes LAST closes the round (spec_advance_body).
TERRA_HD void spec_post_unlink_flip_wave(spec_buffers_t const &sb, uint32_t slot) {
	spec_post_wave(sb, slot);
	uint32_t const n = wave_uniform(sb.linked[slot]);
	TERRA_LANES(e, n) {spec_unlink_body(sb, slot*sb.maxb + (uint32_t)e);}
	if (TERRA_LANE0) {spec_flip_body(sb, slot);}
}
TERRA_HD void spec_close_wave(spec_buffers_t const &sb, uint32_t slot) {
	spec_flush_wave(sb, slot);
	spec_resume_wave(sb, slot);
	TERRA_WAVE_SYNC();
	if (TERRA_LANE0) {
		spec_admit_body(sb, slot);
		// The closing thread needs nothing another wave of this launch WROTE (they change the control block through atomics only, and not the words it rewrites): what it
		// needs is that every wave has finished READING new_base / par / nd2 before they change -- a wave counts itself done after its own reads have returned.  (An
		// agent-scope fence per wave here -- a cache write-back each, 32768 of them on a 16384^2 ring -- made the round 2.5x longer.)
		// Two levels of counters (64 slots each, then the groups): thousands of atomics on ONE word take ~10 ns each, one after the other -- 0.4 ms of a round on a 32768-slot ring.
		TERRA_ORDER_FENCE();
		uint32_t const grp = slot >> 6, ngrp = (sb.W + 63u) >> 6, in_grp = (grp + 1 == ngrp) ? sb.W - (grp << 6) : 64u;
		if (TERRA_ATOMIC_ADD(&sb.done_cnt[grp], 1u) == in_grp - 1) {
			sb.done_cnt[grp] = 0;
			if (TERRA_ATOMIC_ADD(&sb.ctl->waves_done, 1u) == ngrp - 1) {
				sb.ctl->waves_done = 0;
				spec_advance_body(sb);
			}
		}
	}
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
