// terra_kernels.hpp -- hand-written gfx950 kernels for the hot loops (device only; included by terra_hip.hip).
//
//   k_sine_grid     K1+K4: z = sum_k X[x][k]*Y[y][k] as an LDS-tiled rank-90 outer-product contraction + fused
//                   shape/glaciate/island epilogue (mesh_xy_grid_cache_t::eval_index, src/mesh_gen.cpp:766-790)
//   k_noise_grid    K2/K3: per-cell simplex/Perlin fBm and domain warp (get_noise_zval, src/mesh_gen.cpp:734-751)
//   k_tile_erosion  K5 (tile mode): one wave per tile, the clamp-padded 138x138 grid resident in LDS, droplets in serial order
//   k_voxel_sines   K8: rank-60 3-D sine field, z-fastest output (noise_gen_3d::get_val, src/upsurface.cpp:60-70)
//
// No MFMA: the sums must round exactly like the CPU's mul-then-add chain (an MFMA/fma chain rounds once per term).
// Wave = 64 lanes; block sizes are multiples of 64; LDS tiles are read with 16-byte ds_read_b128 and broadcast.
#pragma once
#include "terra_driver.hpp"
#include "terra_noise_kernels.hpp"

namespace terra {

typedef float    st_f4 __attribute__((ext_vector_type(4))); // native vector types: what the nontemporal load / store builtins take
typedef uint32_t st_u4 __attribute__((ext_vector_type(4)));

template<class F> __global__ void k_generic(size_t n, F f) {
	size_t const i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	if (i < n) {f(i);}
}

// one 64-lane workgroup (= one wave) per item, with the per-wave LDS scratch of the droplet window (terra_erosion.hpp)
template<class F> __global__ __launch_bounds__(64) void k_waves(F f, unsigned first) {
	__builtin_amdgcn_s_setprio(3); // (see k_waves_lean)
	__shared__ __attribute__((aligned(16))) float win[EW*EW];
	__shared__ uint8_t dirty[EW*EW];
	__shared__ wave_shared_t sh;
	wave_scratch_t const ws{win, dirty, &sh};
	f((size_t)blockIdx.x + first, ws);
}

template<class F> __global__ __launch_bounds__(64) void k_waves_nolds(F f) {f((size_t)blockIdx.x);}

// the same for the lean traces of the sparse erosion scheduler (terra_erosion.hpp: lean_back_t): window + dirty bytes + 4.6 KB of footprint bookkeeping = 9.8 KB of LDS, and
// at most 128 registers (four waves per SIMD requested) -- beside a k_sine_grid block of another heightmap (29.7 KB, 120 registers per wave) a SIMD then holds three of that
// kernel's waves and one of these, and a CU four of its blocks and four of these; the general trace wave (22.7 KB, 260 registers) leaves room for two and two
template<class F> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_waves_lean(F f) {
	__builtin_amdgcn_s_setprio(3); // a droplet is a chain of dependent instructions: beside three waves of another kernel on its SIMD it should never wait for an issue slot (it asks for one every ~5 cycles)
	__shared__ __attribute__((aligned(16))) float win[EW*EW];
	__shared__ uint8_t dirty[EW*EW];
	__shared__ lean_shared_t sh;
	lean_scratch_t const ws{win, dirty, &sh};
	f((size_t)blockIdx.x, ws);
}


// ------------------------------------------------------------------ K1: sine-sum grid
// z[y][x] = sum_k X[k][x]*Y[k][y] is a rank-(90-kstart) outer-product contraction.  It is fp32-VALU bound: every term costs one
// v_mul and one v_add (no FMA: the CPU reference rounds the product before adding), i.e. 2*terms VALU ops for 4 B written.
// Tiling: 128 x 128 cells per 256-thread block, 8 x 8 cells per thread (64 independent accumulator chains), the K range staged
// through LDS in chunks of <= KC terms (template parameter; the heightmap kernel uses 27: three chunks for 8 octaves, 29.7 KB per block,
// 118 VGPRs -> 4 blocks = 16 waves per CU alone, and two waves per SIMD beside a 264-register droplet wave of another map's erosion;
// the plain tile variant uses 27 as well, the general one keeps SG_KC = 45: two chunks, 48 KB), operands of the next step prefetched from LDS while this one is multiplied.
// Per k a thread issues four ds_read_b128 (conflict-free / broadcast) for 64 mul + 64 add.  The f32 matrix instructions cannot take the
// products: exact with K = 1, C = 0, but they share the packed-f32 datapath (profiles/r04_sine_matrix_pipe.txt).
constexpr int SG_BX = 128, SG_BY = 128, SG_TX = 8, SG_TY = 8, SG_THREADS = 256, SG_KC = 45; // 16 x 16 threads
// XCD-aware tile order: the dispatcher places block b on XCD b % 8 (observed, speed only); give every XCD one contiguous
// band of tile rows so its private L2 keeps that band's Y tiles, and walk the band in groups of `rowgroup` tile rows
// (an X slice is reused by `rowgroup` consecutive blocks before the walk moves to the next tile column).
__device__ __forceinline__ bool sg_tile_of_block(unsigned b, unsigned ntx, unsigned nty, unsigned rowgroup, unsigned &bx, unsigned &by) {
	unsigned const nb = ntx*nty, per_xcd = (nb + 7)/8;
	unsigned const lin = (b & 7)*per_xcd + (b >> 3); // position in the global tile order
	if (lin >= nb) return false;
	unsigned const group = lin/(rowgroup*ntx), r = lin % (rowgroup*ntx);
	unsigned const rows_here = (nty - group*rowgroup < rowgroup) ? nty - group*rowgroup : rowgroup;
	bx = r/rows_here; by = group*rowgroup + r % rows_here;
	return bx < ntx;
}

struct sg_tiles_t {int32_t const *tile_map; float const *m0; uint32_t nux; uint32_t rowgroup; uint32_t tw; // tw: cells per tile edge of the virtual grid (130 zvals, 201 AO context)
	// a BAND of the tiles' fields (round 6: the AO context without the centre nobody reads): tw x twy virtual cells per tile, written into the tile's ostride x ostride field at
	// (c + (c >= split ? gap : 0)) -- the square case: twy = ostride = tw, splits at 0xFFFFFFFF.  Only the packed epilogue of the plain kernel knows about it (the launcher checks).
	uint32_t twy, ostride, xsplit, xgap, ysplit, ygap;};

typedef float sg_v2f __attribute__((ext_vector_type(2)));
// Two rows x eight columns per call: r0[jp] += {x0*y0, x1*y0}, r1[jp] += {x0*y1, x1*y1} with (y0,y1) = one register pair straight from ds_read_b128.
// v_pk_mul_f32 broadcasts the row's y from either half of the pair through op_sel, so no operand is ever copied (the compiler's own selection
// moves half of the broadcast operands into fresh registers every step), and the mul/mul/add/add order keeps one independent instruction
// between each product and the add that consumes it (no hazard nops).  Product rounded by v_pk_mul, then added by v_pk_add: never fused.
__device__ __forceinline__ void sg_mul_add_2x8(sg_v2f (&r0)[SG_TX/2], sg_v2f (&r1)[SG_TX/2], sg_v2f const (&xp)[4], sg_v2f yp) {
	sg_v2f t0, t1;
#define TERRA_SG_2X2(A0, A1, X) \
	"v_pk_mul_f32 %8, " X ", %13 op_sel:[0,0] op_sel_hi:[1,0]\n\t" \
	"v_pk_mul_f32 %9, " X ", %13 op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
	"v_pk_add_f32 " A0 ", " A0 ", %8\n\t" \
	"v_pk_add_f32 " A1 ", " A1 ", %9\n\t"
	asm(TERRA_SG_2X2("%0", "%4", "%10") TERRA_SG_2X2("%1", "%5", "%11") TERRA_SG_2X2("%2", "%6", "%12") TERRA_SG_2X2("%3", "%7", "%14")
	    : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]), "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r1[3]), "=&v"(t0), "=&v"(t1)
	    : "v"(xp[0]), "v"(xp[1]), "v"(xp[2]), "v"(yp), "v"(xp[3]));
#undef TERRA_SG_2X2
}
// acc[i][jp] holds cells (row i, columns 2*jp, 2*jp+1) of the thread's 8 x 8 patch; one call = the four rows that one 16-byte Y read feeds (rows 4 h .. 4 h + 3)
struct sg_xop_t {float4 a, b;};
__device__ __forceinline__ sg_xop_t sg_load_x(float const *px, int k) {return sg_xop_t{*(float4 const *)(px + k*SG_BX), *(float4 const *)(px + k*SG_BX + 64)};}
__device__ __forceinline__ void sg_accumulate_half(sg_v2f (&acc)[SG_TY][SG_TX/2], sg_xop_t const &x, float4 const &y, int h) {
	sg_v2f const xp[4] = {{x.a.x, x.a.y}, {x.a.z, x.a.w}, {x.b.x, x.b.y}, {x.b.z, x.b.w}};
	sg_mul_add_2x8(acc[4*h], acc[4*h + 1], xp, sg_v2f{y.x, y.y});
	sg_mul_add_2x8(acc[4*h + 2], acc[4*h + 3], xp, sg_v2f{y.z, y.w});
}

// GENERAL = false: the host proved that every cell takes the short epilogue (terra_engine::sine_plain_only), so finish_cell() -- 64 inlined
// copies of the plateau / crater / crack / volcano / powf code, ~340 KB of instructions the hot path would otherwise be threaded through -- is left out
template<bool TILES, bool GENERAL, int KC = SG_KC> __global__ __launch_bounds__(SG_THREADS) void k_sine_grid(grid_job_t job, noise_consts_t nc, sin_lut_t L,
	float const *__restrict__ xt, float const *__restrict__ yt, float const *__restrict__ smx, float const *__restrict__ smy, float *__restrict__ out, unsigned ntx, unsigned nty, uint32_t *__restrict__ mm, sg_tiles_t tiles)
{
	__shared__ __attribute__((aligned(16))) float sX[(KC + 2)*SG_BX];
	__shared__ __attribute__((aligned(16))) float sY[(KC + 2)*SG_BY];
	unsigned bxi, byi;
	if (!sg_tile_of_block(blockIdx.x, ntx, nty, tiles.rowgroup, bxi, byi)) return;
	uint32_t mm_lo = 0xFFFFFFFFu, mm_hi = 0xFFFFFFFFu; // fused min(vals)/max(vals) (heightmap_t::run_erosion, get_heightmap_z_range): saves a 4 B/cell read pass
	unsigned const tid = threadIdx.x, bx0 = bxi*SG_BX, by0 = byi*SG_BY;
	unsigned const tx = tid & 15, ty = tid >> 4;
	// thread (tx,ty) owns columns {tx*4..+3} u {64+tx*4..+3} and rows {ty*4..+3} u {64+ty*4..+3}: every LDS read is one aligned ds_read_b128,
	// 16 distinct 16-byte slots per 16-lane group for X (conflict-free) and a broadcast for Y
	float const *px = sX + tx*4, *py = sY + ty*4;
	sg_v2f acc[SG_TY][SG_TX/2];
#pragma unroll
	for (int i = 0; i < SG_TY; ++i) {
#pragma unroll
		for (int jp = 0; jp < SG_TX/2; ++jp) {acc[i][jp] = sg_v2f{0.0f, 0.0f};}
	}
	int const nk = F_TABLE_SIZE - job.kstart, nchunks = (nk + KC - 1)/KC, per_chunk = (nk + nchunks - 1)/nchunks;
	for (int c = 0; c < nchunks; ++c) { // terms are summed in k order across chunks, exactly like the CPU loop
		int const k0 = job.kstart + c*per_chunk, kn = ((k0 + per_chunk > F_TABLE_SIZE) ? F_TABLE_SIZE - k0 : per_chunk);
		if (c > 0) {__syncthreads();} // everyone is done reading the previous chunk
		{	// stage the chunk: thread (kr, q) copies 16-byte group q of rows kr, kr + 8, ... of both tables.  All loads of a table's chunk are issued before its first LDS write (two
			// HBM / L2 latencies per chunk instead of one per row group), addresses advance by a uniform stride (tables are zero-padded to a multiple of 128 columns / rows)
			constexpr int ITERS = (KC*(SG_BX/4) + SG_THREADS - 1)/SG_THREADS;
			unsigned const kr = tid >> 5, q4 = (tid & 31u)*4u;
			float const *const gx = xt + (size_t)(k0 + (int)kr)*job.nxp + bx0 + q4, *const gy = yt + (size_t)(k0 + (int)kr)*job.nyp + by0 + q4;
#pragma unroll
			for (int tab = 0; tab < 2; ++tab) { // X, then Y: ITERS 16-byte loads in flight per thread (both tables at once would not fit the register budget of four waves per SIMD)
				float const *const g = tab ? gy : gx; size_t const stride = tab ? job.nyp : job.nxp; float *const sT = tab ? sY : sX;
				float4 v[ITERS];
#pragma unroll
				for (int it = 0; it < ITERS; ++it) {v[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); if ((int)kr + 8*it < kn) {v[it] = *(float4 const *)(g + (size_t)(8*it)*stride);}}
#pragma unroll
				for (int it = 0; it < ITERS; ++it) {if ((int)kr + 8*it < kn) {*(float4 *)&sT[(kr + 8u*it)*SG_BX + q4] = v[it];}}
			}
		}
		__syncthreads();
		// Operand registers: X (the eight columns, used by all eight rows of a step) in two sets that alternate between steps; Y (four rows per 16-byte read) in ONE set --
		// a Y read is reloaded with the next step's values right after the four rows that use it, half a step (64 packed instructions, ~256 cycles) before it is needed
		// again, an X set right after its step, a whole step ahead.  24 operand registers instead of 32: the kernel stays under 120 VGPRs, so that two of its waves fit on a
		// SIMD beside a droplet wave of another heightmap's erosion (264 registers), where one did before.  The scheduling barriers keep the loads where they are written
		// (the compiler would hoist them above the multiplies whose operands they replace, and copy).  Rows kn, kn + 1 may be read but are never used (KC + 2 rows are allocated).
		sg_xop_t XA = sg_load_x(px, 0), XB = sg_load_x(px, 1);
		float4 ya = *(float4 const *)py, yb = *(float4 const *)(py + 64);
		int k = 0;
		for (; k + 1 < kn; k += 2) {
			sg_accumulate_half(acc, XA, ya, 0); __builtin_amdgcn_sched_barrier(0); ya = *(float4 const *)(py + (k + 1)*SG_BY);
			sg_accumulate_half(acc, XA, yb, 1); __builtin_amdgcn_sched_barrier(0); yb = *(float4 const *)(py + (k + 1)*SG_BY + 64); XA = sg_load_x(px, k + 2);
			sg_accumulate_half(acc, XB, ya, 0); __builtin_amdgcn_sched_barrier(0); ya = *(float4 const *)(py + (k + 2)*SG_BY);
			sg_accumulate_half(acc, XB, yb, 1); __builtin_amdgcn_sched_barrier(0); yb = *(float4 const *)(py + (k + 2)*SG_BY + 64); XB = sg_load_x(px, k + 3);
		}
		if (k < kn) {sg_accumulate_half(acc, XA, ya, 0); sg_accumulate_half(acc, XA, yb, 1);}
	}
	// ---- epilogue (eval_index's tail, src/mesh_gen.cpp:781-790): shape / post-process, glaciate, sine-mag islands, volcano.
	// The common configuration (linear shape, no plateau/crater/crack, no volcano) takes a short path with the island terms of the
	// thread's 8 columns / 8 rows loaded once; anything else goes through the general finish_cell().  Same arithmetic either way.
	bool const vec_ok = ((job.nx & 3u) == 0) && !TILES;
	if (!GENERAL && TILES && (job.nx & 3u) == 0) {
		// the tile batch's common case: the packed epilogue of the grid (below), then the scatter into the per-tile layout -- a 4-cell group that lies in one tile goes out as ONE
		// 16-byte store (tile rows are 4-byte aligned only: tw = 130, 201, 129), a group across a tile boundary cell by cell.  (The per-cell epilogue with 8-byte stores made this
		// variant run at 0.7 of the grid kernel's rate: profiles/r06_tiles_kernel_stats.txt.)
		typedef sg_v2f v2f;
		typedef float st_f4u __attribute__((ext_vector_type(4), aligned(4)));
		bool const gl = job.glaciate && nc.glaciate, sm = job.glaciate && job.use_sine_mag;
		float4 sx[2], sy[2];
		sx[0] = sx[1] = sy[0] = sy[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (sm) { // (the batch's island tables end with the virtual grid: nothing is read behind nx / ny; nx is a multiple of 4 here, ny need not be)
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				unsigned const x = bx0 + h*64 + tx*4, y = by0 + h*64 + ty*4;
				if (x < job.nx) {sx[h] = *(float4 const *)(smx + x);}
				sy[h] = make_float4((y < job.ny) ? smy[y] : 0.0f, (y + 1 < job.ny) ? smy[y + 1] : 0.0f, (y + 2 < job.ny) ? smy[y + 2] : 0.0f, (y + 3 < job.ny) ? smy[y + 3] : 0.0f);
			}
		}
		v2f const zme = {nc.zmax_est, nc.zmax_est}, inv = {nc.zmax_est2_inv, nc.zmax_est2_inv}, z2 = {nc.zmax_est2, nc.zmax_est2}, off = {job.sine_offset, job.sine_offset};
		unsigned const tw = tiles.tw, twy = tiles.twy, os = tiles.ostride, nuy = job.ny/twy; // (the virtual grid is nux x nuy tiles of tw x twy cells)
		unsigned ux0[2], cx0[2], uy0[2], cy0[2];
#pragma unroll
		for (int half = 0; half < 2; ++half) {unsigned const x = bx0 + half*64 + tx*4; ux0[half] = x/tw; cx0[half] = x - ux0[half]*tw;}
#pragma unroll
		for (int g = 0; g < 2; ++g) {unsigned const y = by0 + g*64 + ty*4; uy0[g] = y/twy; cy0[g] = y - uy0[g]*twy;}
		// the tiles under the thread's cells: two tile columns per half at most (a group may cross into the next), two tile rows per 4-row group
		int tmap[2][2][2][2];
#pragma unroll
		for (int g = 0; g < 2; ++g) {
#pragma unroll
			for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
				for (int half = 0; half < 2; ++half) {
#pragma unroll
					for (int dx = 0; dx < 2; ++dx) {
						unsigned const uy = uy0[g] + dy, ux = ux0[half] + dx;
						tmap[g][dy][half][dx] = (uy < nuy && ux < tiles.nux) ? tiles.tile_map[uy*tiles.nux + ux] : -1;
					}
				}
			}
		}
#pragma unroll
		for (int i = 0; i < SG_TY; ++i) {
			unsigned const y = by0 + (i >> 2)*64 + ty*4 + (i & 3);
			if (y >= job.ny) continue;
			float const syq[4] = {sy[i >> 2].x, sy[i >> 2].y, sy[i >> 2].z, sy[i >> 2].w};
			v2f const syi = {syq[i & 3], syq[i & 3]};
			unsigned cy = cy0[i >> 2] + (unsigned)(i & 3); int dy = 0;
			if (cy >= twy) {cy -= twy; dy = 1;}
			unsigned const oy = cy + ((cy >= tiles.ysplit) ? tiles.ygap : 0u); // row of the tile's field
#pragma unroll
			for (int half = 0; half < 2; ++half) {
				unsigned const x = bx0 + half*64 + tx*4;
				if (x >= job.nx) continue;
				v2f z01 = acc[i][half*2], z23 = acc[i][half*2 + 1];
				if (gl) {
					v2f const r01 = (z01 + zme)*inv, r23 = (z23 + zme)*inv;
					z01 = ((r01*r01)*r01)*z2 - zme; z23 = ((r23*r23)*r23)*z2 - zme;
				}
				if (sm) {
					v2f const s01 = {sx[half].x, sx[half].y}, s23 = {sx[half].z, sx[half].w};
					z01 = z01 + (s01*syi + off); z23 = z23 + (s23*syi + off);
				}
				unsigned const cx = cx0[half];
				int const ta = dy ? tmap[i >> 2][1][half][0] : tmap[i >> 2][0][half][0], tb = dy ? tmap[i >> 2][1][half][1] : tmap[i >> 2][0][half][1];
				if (cx + 3 < tw && (cx + 3 < tiles.xsplit || cx >= tiles.xsplit)) { // the whole group in one tile row (and on one side of a band's gap)
					unsigned const ox = cx + ((cx >= tiles.xsplit) ? tiles.xgap : 0u);
					if (ta >= 0) {*(st_f4u *)(out + (size_t)ta*os*os + oy*os + ox) = st_f4u{z01.x, z01.y, z23.x, z23.y};}
				}
				else {
					float const v[4] = {z01.x, z01.y, z23.x, z23.y};
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						bool const next = cx + (unsigned)j >= tw;
						int const t = next ? tb : ta;
						unsigned const cxx = next ? cx + (unsigned)j - tw : cx + (unsigned)j, ox = cxx + ((cxx >= tiles.xsplit) ? tiles.xgap : 0u);
						if (t >= 0) {out[(size_t)t*os*os + oy*os + ox] = v[j];}
					}
				}
			}
		}
		return; // (a tile batch has no fused min / max)
	}
	if (!GENERAL && !TILES && vec_ok) {
		// the common case, two cells per instruction (v_pk_mul_f32 / v_pk_add_f32): the island tables are zero-padded to the tile grid, rows of 4 cells
		// are either wholly inside the grid or wholly outside
		typedef sg_v2f v2f;
		bool const gl = job.glaciate && nc.glaciate, sm = job.glaciate && job.use_sine_mag;
		float4 sx[2], sy[2];
		sx[0] = sx[1] = sy[0] = sy[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (sm) {
			sx[0] = *(float4 const *)(smx + bx0 + tx*4); sx[1] = *(float4 const *)(smx + bx0 + 64 + tx*4);
			sy[0] = *(float4 const *)(smy + by0 + ty*4); sy[1] = *(float4 const *)(smy + by0 + 64 + ty*4);
		}
		v2f const zme = {nc.zmax_est, nc.zmax_est}, inv = {nc.zmax_est2_inv, nc.zmax_est2_inv}, z2 = {nc.zmax_est2, nc.zmax_est2}, off = {job.sine_offset, job.sine_offset};
		float fmn = INFINITY, fmx = -INFINITY;
#pragma unroll
		for (int i = 0; i < SG_TY; ++i) {
			unsigned const y = by0 + (i >> 2)*64 + ty*4 + (i & 3);
			if (y >= job.ny) continue;
			float const syq[4] = {sy[i >> 2].x, sy[i >> 2].y, sy[i >> 2].z, sy[i >> 2].w};
			v2f const syi = {syq[i & 3], syq[i & 3]};
#pragma unroll
			for (int half = 0; half < 2; ++half) {
				unsigned const x = bx0 + half*64 + tx*4;
				if (x >= job.nx) continue;
				v2f z01 = acc[i][half*2], z23 = acc[i][half*2 + 1];
				if (gl) {
					v2f const r01 = (z01 + zme)*inv, r23 = (z23 + zme)*inv;
					z01 = ((r01*r01)*r01)*z2 - zme; z23 = ((r23*r23)*r23)*z2 - zme;
				}
				if (sm) {
					v2f const s01 = {sx[half].x, sx[half].y}, s23 = {sx[half].z, sx[half].w};
					z01 = z01 + (s01*syi + off); z23 = z23 + (s23*syi + off);
				}
				fmn = sg_min3(sg_min3(fmn, z01.x, z01.y), z23.x, z23.y); // (v_min3 / v_max3 by hand: fminf / fmaxf on values that come out of the inline-asm sum cost a canonicalising
				fmx = sg_max3(sg_max3(fmx, z01.x, z01.y), z23.x, z23.y); //  v_max x, x per operand -- 12 instructions per 4 cells instead of 4; NaNs are skipped either way)
				*(float4 *)(out + (size_t)y*job.nx + x) = make_float4(z01.x, z01.y, z23.x, z23.y);
			}
		}
		if (mm) {
			if (fmn <= fmx) {mm_lo = f2ord(fmn); mm_hi = ~f2ord(fmx);}
			wave_minmax_publish(mm_lo, mm_hi, mm);
		}
		return;
	}
	hmap_params_t const &hp = nc.hp;
	bool const plain = !GENERAL || ((job.shape == 0) && !(hp.crack_lo < hp.crack_hi) && !(hp.volcano_width > 0.0f && hp.volcano_height > 0.0f));
	float const pp_limit = min_std(hp.plat_bot, hp.crat_h); // below this the post-process is the identity
	float smxv[SG_TX], smyv[SG_TY];
#pragma unroll
	for (int j = 0; j < SG_TX; ++j) {unsigned const x = bx0 + (j >> 2)*64 + tx*4 + (j & 3); smxv[j] = (job.use_sine_mag && x < job.nx) ? smx[x] : 0.0f;}
#pragma unroll
	for (int i = 0; i < SG_TY; ++i) {unsigned const y = by0 + (i >> 2)*64 + ty*4 + (i & 3); smyv[i] = (job.use_sine_mag && y < job.ny) ? smy[y] : 0.0f;}
	float fmn = INFINITY, fmx = -INFINITY;
	unsigned t_ux[2] = {0, 0}, t_cx[2] = {0, 0}, t_uy[4] = {0, 0, 0, 0}, t_cy[4] = {0, 0, 0, 0}; // TILES: tile column / row and offset of the thread's two 4-cell groups and four 4-row groups
	if (TILES) {
#pragma unroll
		for (int half = 0; half < 2; ++half) {unsigned const x = bx0 + half*64 + tx*4; t_ux[half] = x/tiles.tw; t_cx[half] = x - t_ux[half]*tiles.tw;}
#pragma unroll
		for (int g = 0; g < 4; ++g) {unsigned const y = by0 + g*64 + ty*4; t_uy[g] = y/tiles.tw; t_cy[g] = y - t_uy[g]*tiles.tw;}
	}
#pragma unroll
	for (int i = 0; i < SG_TY; ++i) {
		unsigned const y = by0 + (i >> 2)*64 + ty*4 + (i & 3);
		if (y >= job.ny) continue;
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			unsigned const x = bx0 + half*64 + tx*4;
			if (x >= job.nx) continue;
			float v[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				float z = acc[i][half*2 + (j >> 1)][j & 1];
				if (!GENERAL || (plain && !(z > pp_limit))) {
					if (job.glaciate) {
						if (nc.glaciate) {float const relh = (z + nc.zmax_est)*nc.zmax_est2_inv; z = (GENERAL ? glaciate_exp_fn(relh, nc.custom_glaciate_exp) : relh*relh*relh)*nc.zmax_est2 - nc.zmax_est;} // !GENERAL: custom_glaciate_exp == 0
						if (job.use_sine_mag) {z += smxv[half*4 + j]*smyv[i] + job.sine_offset;}
					}
				}
				else if (GENERAL && x + j < job.nx) {
					if (TILES) { // general epilogue in tile-local coordinates (volcano term needs the tile's own origin)
						unsigned const tw = tiles.tw, ux = (x + j)/tw, uy = y/tw;
						grid_job_t jt = job; jt.mx0 = tiles.m0[ux]; jt.my0 = tiles.m0[tiles.nux + uy];
						z = finish_cell(z, jt, nc, L, smx + ux*tw, smy + uy*tw, (x + j) - ux*tw, y - uy*tw);
					}
					else {z = finish_cell(z, job, nc, L, smx, smy, x + j, y);}
				}
				else {z = 0.0f;}
				v[j] = z;
				if (x + j < job.nx) {fmn = fminf(fmn, z); fmx = fmaxf(fmx, z);} // fminf/fmaxf skip NaNs, like min_eq/max_eq never let a NaN win
			}
			if (TILES) { // scatter into the per-tile layout: tile column / row and the offsets inside come from the thread's precomputed bases (no division per cell)
				unsigned const tw = tiles.tw;
				unsigned uy = t_uy[i >> 2], cy = t_cy[i >> 2] + (unsigned)(i & 3);
				if (cy >= tw) {cy -= tw; ++uy;}
				if ((tw & 1u) == 0) { // even tile width (130): x is a multiple of 4, so cell pairs (j, j+1) share a tile row and are 8-byte aligned there
#pragma unroll
					for (int j = 0; j < 4; j += 2) {
						if (x + j >= job.nx) continue; // nx = nux*tw is even as well: the pair is inside or outside together
						unsigned ux = t_ux[half], cx = t_cx[half] + (unsigned)j;
						if (cx >= tw) {cx -= tw; ++ux;}
						int const t = tiles.tile_map[uy*tiles.nux + ux];
						if (t >= 0) {*(float2 *)(out + (size_t)t*tw*tw + cy*tw + cx) = make_float2(v[j], v[j + 1]);}
					}
					continue;
				}
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					if (x + j >= job.nx) continue;
					unsigned ux = t_ux[half], cx = t_cx[half] + (unsigned)j;
					if (cx >= tw) {cx -= tw; ++ux;}
					int const t = tiles.tile_map[uy*tiles.nux + ux];
					if (t >= 0) {out[(size_t)t*tw*tw + cy*tw + cx] = v[j];}
				}
				continue;
			}
			float *o = out + (size_t)y*job.nx + x;
			if (vec_ok) {*(float4 *)o = make_float4(v[0], v[1], v[2], v[3]);}
			else {
#pragma unroll
				for (int j = 0; j < 4; ++j) {if (x + j < job.nx) o[j] = v[j];}
			}
		}
	}
	if (mm) {
		if (fmn <= fmx) {mm_lo = f2ord(fmn); mm_hi = ~f2ord(fmx);}
		wave_minmax_publish(mm_lo, mm_hi, mm);
	}
}

// ------------------------------------------------------------------ K5 tile mode: LDS-resident padded grid, serial droplets
// how much of a tile is land: droplets that start under water stop at once, the others walk ~50 steps, so this predicts a tile's serial chain length
__global__ __launch_bounds__(256) void k_tile_land_cells(float const *__restrict__ zvals, uint32_t cells, float water_thresh, uint32_t *__restrict__ land) {
	float const *z = zvals + (size_t)blockIdx.x*cells;
	uint32_t cnt = 0;
	for (uint32_t i = threadIdx.x; i < cells; i += 256) {cnt += !(z[i] < water_thresh) ? 1u : 0u;} // a NaN counts as land: it does not pass the ocean test either
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {cnt += __shfl_down(cnt, off, 64);}
	if ((threadIdx.x & 63) == 0 && cnt) {atomicAdd(&land[blockIdx.x], cnt);}
}
// the stable descending order of the tiles by land count, without a sort: tile i goes to position #{j : land[j] > land[i]} + #{j < i : land[j] == land[i]}
__global__ __launch_bounds__(256) void k_tile_order_by_land(uint32_t const *__restrict__ land, uint32_t n, uint32_t *__restrict__ order) {
	__shared__ uint32_t s_land[256];
	uint32_t const i = blockIdx.x*256 + threadIdx.x;
	uint32_t const mine = (i < n) ? land[i] : 0u;
	uint32_t pos = 0;
	for (uint32_t j0 = 0; j0 < n; j0 += 256) {
		__syncthreads();
		s_land[threadIdx.x] = (j0 + threadIdx.x < n) ? land[j0 + threadIdx.x] : 0u;
		__syncthreads();
		uint32_t const cnt = (n - j0 < 256u) ? n - j0 : 256u;
		for (uint32_t e = 0; e < cnt; ++e) {uint32_t const l = s_land[e]; pos += (l > mine || (l == mine && j0 + e < i)) ? 1u : 0u;}
	}
	if (i < n) {order[pos] = i;}
}
// `order` (or null): the tile each block takes.  Blocks are dispatched in index order and only two tiles fit a CU, so the batch is a list-scheduling
// problem: handing out the longest chains first keeps the heavy land tiles from forming the tail of the launch.
__global__ __launch_bounds__(64) void k_tile_erosion(float *__restrict__ zvals, erosion_consts_t ec, uint32_t iters, uint32_t const *__restrict__ order, uint32_t const *__restrict__ land) {
	extern __shared__ __attribute__((aligned(16))) float te_pad[];
	int const NX = ec.NX, NY = ec.NY, xs = ec.xsize, ys = ec.ysize;
	uint32_t const tile = order ? order[blockIdx.x] : blockIdx.x;
	float *z = zvals + (size_t)tile*xs*ys;
	if (land && land[tile] == 0) { // every cell (and so every clamp-padded copy) is below the ocean threshold: each droplet stops at its first step without a write
		// (src/erosion.cpp:98), so apply_erosion reduces to its final clamp -- no need to hold one of the chip's 512 LDS tile slots for a thousand such droplets
		for (int i = threadIdx.x; i < xs*ys; i += 64) {z[i] = max_std(ec.min_zval, z[i]);}
		return;
	}
	for (int i = threadIdx.x; i < NX*NY; i += 64) { // clamp-padded copy (src/erosion.cpp:31-37), coalesced rows
		int const X = i % NX, Z = i / NX;
		te_pad[i] = z[(size_t)imax(imin(Z - EROSION_PAD, ys-1), 0)*xs + imax(imin(X - EROSION_PAD, xs-1), 0)];
	}
	__syncthreads();
	{ // droplet order is the semantics: droplets run one after another; within a droplet the wave's lanes share the brush / corner accesses (LDS latency, not HBM, bounds a step)
		wave_lds_mem_t m; m.pad = te_pad; m.NX = NX; m.NY = NY;
		for (uint32_t it = 0; it < iters; ++it) {simulate_droplet((int)it, m, ec);}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < xs*ys; i += 64) { // unpad + clamp (src/erosion.cpp:158-162)
		int const x = i % xs, y = i / xs;
		z[i] = max_std(ec.min_zval, te_pad[(y + EROSION_PAD)*NX + (x + EROSION_PAD)]);
	}
}

// ------------------------------------------------------------------ K6+K7: tile post-pass, one 256-thread block per tile
// sub-block z ranges, water bbox (ints), mzmin/mzmax/radius (src/tiled_mesh.cpp:517-541) and RGBA8 normals + min_normal_z (src/tiled_mesh.cpp:865-880).
// HBM-bound: 4 B read + 4 B written per cell; reductions go through LDS atomics on order-preserving uints.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {uint32_t const o = __shfl_down(v, off, 64); v = (o < v) ? o : v;}
	return v; // valid in lane 0
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {int const o = __shfl_down(v, off, 64); v = (o < v) ? o : v;}
	return v;
}
// One block per (tile, band of 32 texel rows): 4 x more blocks than tiles and 17.7 KB of LDS each.  Band yy owns sub-block row yy completely (rows 32*yy .. 32*yy + 32), so the
// 4 x 4 sub-block ranges are written by exactly one block each; what spans the tile (the water bbox, min_normal_z) is folded through 8 words of global scratch per tile with
// atomics and written by a one-thread-per-tile kernel afterwards.
//
// Round 5: the kernel was instruction-bound (127 per texel: three IEEE divisions, a square root, three double-precision byte conversions, an index division).  Now a WAVE walks
// rows -- lane = column, both halves of a row per step, column 128 in one extra pass -- so the row / water bookkeeping is scalar, and a texel's three bytes come from ~40 instructions:
//   * the bytes are floor(127*(n_i/|n| + 1)).  t_i = fma(127, n_i*rsq(s), 127) is within 4.6e-5 of the double-precision value the reference truncates (rsq <= 2^-23 relative:
//     terra_selftest_hot_sqrt checks every fp32 input; RN(n/RN(sqrt(s))) is within 2^-23 relative of n/sqrt(s); the fma rounds once, <= 2^-17), so when every t_i of a wave's step
//     is further than TP_EPS = 2^-14 from an integer, floor(t_i) IS the reference's byte.  Otherwise (a few percent of the steps, and every NaN / Inf) the whole step is redone
//     with the reference's own statements (tile_normal_v + the double conversion).  Exactly flat texels (n_x = n_y = 0: ocean floor) would always land on an integer: their word
//     is a launch constant, computed by the host with the same statements.
//   * min_normal_z = min over texels of RN(dxdy/RN(sqrt(s))) is monotone in s: the wave keeps max(s) (NaN never wins, as in std::min) and k_tile_post_final takes the one square
//     root and division.  The reference's "mag < TOLERANCE: leave unnormalized" branch cannot be taken when sqrtf(dxdy*dxdy) >= TOLERANCE, which the host checks (else: simple path).
constexpr unsigned TP_THREADS = 256, TP_BAND_ROWS = 34, TP_ACC = 8; // acc: {-, -, bbox x1, y1, x2, y2, max |n|^2 bits, -}
constexpr float TP_EPS = 0x1p-14f;
__global__ __launch_bounds__(256) void k_tile_post_init(uint32_t *__restrict__ acc, uint32_t n) {
	uint32_t const i = blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= n*TP_ACC) return;
	uint32_t const f = i % TP_ACC;
	acc[i] = (f < 2) ? 0xFFFFFFFFu : ((f < 4) ? 0x7FFFFFFFu : ((f < 6) ? 0x80000000u : 0u));
}
// the reference's texel, statement by statement (src/tiled_mesh.cpp:865-880)
__device__ __forceinline__ uint32_t tp_word_exact(float zc, float zr, float zd, float dxv, float dyv, float dxy) {
	float nv[3];
	tile_normal_v(zc, zr, zd, dxv, dyv, dxy, nv);
	uint32_t const b0 = (uint8_t)(127.0*((double)nv[0] + 1.0)), b1 = (uint8_t)(127.0*((double)nv[1] + 1.0)), b2 = (uint8_t)(127.0*((double)nv[2] + 1.0));
	return b0 | (b1 << 8) | (b2 << 16); // A = 0
}
// the short form: word and |n|^2; returns false when a byte is not certain
__device__ __forceinline__ bool tp_word_fast(float zc, float zr, float zd, float dxv, float dyv, float dxy, float c2, uint32_t flat_word, uint32_t &word, float &s) {
	float const n0 = dyv*(zc - zr), n1 = dxv*(zc - zd);
	s = n0*n0 + n1*n1 + c2; // the reference's sum, in its order; c2 = dxdy*dxdy
	float const r = rsq_approx(s);
	float const t0 = __builtin_fmaf(127.0f, n0*r, 127.0f), t1 = __builtin_fmaf(127.0f, n1*r, 127.0f), t2 = __builtin_fmaf(127.0f, dxy*r, 127.0f);
	float const lim = 0.5f - TP_EPS;
	bool const sure = (__builtin_fabsf(__builtin_amdgcn_fractf(t0) - 0.5f) < lim) & (__builtin_fabsf(__builtin_amdgcn_fractf(t1) - 0.5f) < lim) & (__builtin_fabsf(__builtin_amdgcn_fractf(t2) - 0.5f) < lim); // false for NaN
	bool const flat = (n0 == 0.0f) & (n1 == 0.0f);
	uint32_t const w = (uint32_t)t0 | ((uint32_t)t1 << 8) | ((uint32_t)t2 << 16);
	word = flat ? flat_word : w;
	return sure | flat;
}
__global__ __launch_bounds__(TP_THREADS) void k_tile_post(tile_ref_pod_t const *__restrict__ refs, float const *__restrict__ zvals, terra_tile_stats *__restrict__ stats,
	uint8_t *__restrict__ normals, uint32_t *__restrict__ acc, float wpz_max, float dxv, float dyv, float dxy, float c2, uint32_t flat_word)
{
	__shared__ __attribute__((aligned(16))) float tp_z[TP_BAND_ROWS*130];
	__shared__ uint32_t s_lo[4], s_hi[4], s_smax;
	__shared__ int s_bb[4];
	unsigned const t = blockIdx.x >> 2, yy = blockIdx.x & 3u, tid = threadIdx.x, zv = 130, stride = 129, bs = 32, row0 = yy*bs;
	tile_ref_pod_t const r = refs[t];
	int const x1 = r.tx*128, y1 = r.ty*128;
	{
		st_f4 const *src = (st_f4 const *)(zvals + (size_t)t*zv*zv + (size_t)row0*zv); // 67600 bytes per tile, 16640 per band: 16-byte aligned
		for (unsigned i = tid; i < TP_BAND_ROWS*zv/4; i += TP_THREADS) {((st_f4 *)tp_z)[i] = __builtin_nontemporal_load(src + i);} // read once (measured: 127.2 -> 125.5 us with the stores below)
	}
	if (tid < 4) {s_lo[tid] = f2ord(100.0f); s_hi[tid] = ~f2ord(-100.0f);} // folds start at szmin = FAR_DISTANCE, szmax = -FAR_DISTANCE
	if (tid == 0) {s_smax = 0u; s_bb[0] = x1 + 128; s_bb[1] = y1 + 128; s_bb[2] = x1; s_bb[3] = y1;} // water bbox starts denormalized
	__syncthreads();
	unsigned const w = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63; // the wave index as a scalar: rows, row tests and the water rows stay on the scalar unit
	unsigned const nrows = (yy == 3) ? bs + 1 : bs; // texel rows of this band: 32, the last band also row 128
	uint32_t *nout = normals ? (uint32_t *)(normals + (size_t)t*stride*stride*4) + (size_t)row0*stride : nullptr;
	bool const want_stats = stats != nullptr;
	// rows [ya, yb) of the band's 33 cell rows belong to this wave; lane = cell column x (first half) and 64 + x (second half)
	unsigned const ya = w*8, yb = (w == 3) ? bs + 1 : ya + 8;
	float loA = 100.0f, hiA = -100.0f, loB = 100.0f, hiB = -100.0f, smax = 0.0f;
	unsigned long long wetA = 0, wetB = 0; // columns with a cell under wpz_max
	int wy0 = 0x7FFFFFFF, wy1 = -1; // first / last such row (band coordinates)
	float const *zl = tp_z + lane;
	float cA = zl[ya*zv], cB = zl[ya*zv + 64];
	for (unsigned y = ya; y < yb; ++y) {
		float const *row = zl + y*zv;
		float const rA = row[1], rB = row[65], dA = row[zv], dB = row[zv + 64];
		if (want_stats) {
			loA = (cA < loA) ? cA : loA; hiA = (cA > hiA) ? cA : hiA; loB = (cB < loB) ? cB : loB; hiB = (cB > hiB) ? cB : hiB; // std::min / std::max: a NaN never wins
			unsigned long long const mA = __builtin_amdgcn_ballot_w64(cA < wpz_max), mB = __builtin_amdgcn_ballot_w64(cB < wpz_max);
			wetA |= mA; wetB |= mB;
			if (mA | mB) {wy0 = (wy0 < (int)y) ? wy0 : (int)y; wy1 = (int)y;}
		}
		if (nout && y < nrows) {
			uint32_t wA, wB; float sA, sB;
			bool const okA = tp_word_fast(cA, rA, dA, dxv, dyv, dxy, c2, flat_word, wA, sA), okB = tp_word_fast(cB, rB, dB, dxv, dyv, dxy, c2, flat_word, wB, sB);
			if (__builtin_amdgcn_ballot_w64(!(okA & okB)) != 0) {wA = tp_word_exact(cA, rA, dA, dxv, dyv, dxy); wB = tp_word_exact(cB, rB, dB, dxv, dyv, dxy);}
			__builtin_nontemporal_store(wA, &nout[y*stride + lane]); __builtin_nontemporal_store(wB, &nout[y*stride + lane + 64]); // written once, read by nobody here
			smax = __builtin_fmaxf(smax, __builtin_fmaxf(sA, sB));
		}
		cA = dA; cB = dB;
	}
	if (want_stats) { // sub-block xx covers columns [32*xx, 32*xx + 32]: the shared columns 32, 64, 96 count on both sides
		uint32_t la = f2ord(loA), ha = ~f2ord(hiA), lb = f2ord(loB), hb = ~f2ord(hiB);
		uint32_t const la0 = la, ha0 = ha, lb0 = lb, hb0 = hb;
#pragma unroll
		for (int off = 16; off > 0; off >>= 1) { // minima over each half of the wave
			uint32_t o;
			o = __shfl_xor(la, off, 64); la = (o < la) ? o : la; o = __shfl_xor(ha, off, 64); ha = (o < ha) ? o : ha;
			o = __shfl_xor(lb, off, 64); lb = (o < lb) ? o : lb; o = __shfl_xor(hb, off, 64); hb = (o < hb) ? o : hb;
		}
		if (lane == 0)  {atomicMin(&s_lo[0], la); atomicMin(&s_hi[0], ha); atomicMin(&s_lo[2], lb); atomicMin(&s_hi[2], hb); atomicMin(&s_lo[1], lb0); atomicMin(&s_hi[1], hb0);} // column 64 also closes sub-block 1
		if (lane == 32) {atomicMin(&s_lo[1], la); atomicMin(&s_hi[1], ha); atomicMin(&s_lo[3], lb); atomicMin(&s_hi[3], hb); atomicMin(&s_lo[0], la0); atomicMin(&s_hi[0], ha0); atomicMin(&s_lo[2], lb0); atomicMin(&s_hi[2], hb0);} // columns 32 and 96 close sub-blocks 0 and 2
	}
	int wx0 = 0x7FFFFFFF, wx1 = -1;
	if (wetA) {wx0 = __builtin_ctzll(wetA); wx1 = 63 - __builtin_clzll(wetA);}
	if (wetB) {int const f = 64 + __builtin_ctzll(wetB); wx0 = (wx0 < f) ? wx0 : f; wx1 = 127 - __builtin_clzll(wetB);}
	if (w == 1) { // column 128: a lane per row
		unsigned const y = lane;
		bool const in = y <= bs;
		float const *c = tp_z + (in ? y : 0u)*zv + 128;
		float const zc = c[0], zr = c[1], zd = c[zv];
		if (want_stats) {
			if (in && zc == zc) {uint32_t const o = f2ord(zc); atomicMin(&s_lo[3], o); atomicMin(&s_hi[3], ~o);}
			unsigned long long const m = __builtin_amdgcn_ballot_w64(in && zc < wpz_max);
			if (m) {
				int const f = __builtin_ctzll(m), l = 63 - __builtin_clzll(m);
				wx1 = 128; wx0 = (wx0 < 128) ? wx0 : 128; wy0 = (wy0 < f) ? wy0 : f; wy1 = (wy1 > l) ? wy1 : l;
			}
		}
		if (nout) {
			uint32_t wd; float s;
			bool const ok = tp_word_fast(zc, zr, zd, dxv, dyv, dxy, c2, flat_word, wd, s);
			if (__builtin_amdgcn_ballot_w64(!ok) != 0) {wd = tp_word_exact(zc, zr, zd, dxv, dyv, dxy);}
			if (y < nrows) {nout[y*stride + 128] = wd; smax = __builtin_fmaxf(smax, s);}
		}
	}
	if (want_stats && lane == 0 && wx1 >= 0) {atomicMin(&s_bb[0], x1 + wx0); atomicMin(&s_bb[1], y1 + (int)row0 + wy0); atomicMax(&s_bb[2], x1 + wx1); atomicMax(&s_bb[3], y1 + (int)row0 + wy1);}
	if (nout) { // s >= 0: positive floats order like their bit patterns
		uint32_t u; memcpy(&u, &smax, 4);
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {uint32_t const o = __shfl_xor(u, off, 64); u = (o > u) ? o : u;}
		if (lane == 0) {atomicMax(&s_smax, u);}
	}
	__syncthreads();
	if (tid == 0) { // the band's results; the tile's totals are folded by k_tile_post_final after this kernel (a device-wide fence per block, the
		// alternative, writes back the XCD's L2 every time on this chip: measured 3 x slower than the whole pass)
		uint32_t *a = acc + (size_t)t*TP_ACC;
		if (want_stats) {
			for (int k = 0; k < 4; ++k) {stats[t].sub_zmin[yy*4 + k] = ord2f(s_lo[k]); stats[t].sub_zmax[yy*4 + k] = ord2f(~s_hi[k]);}
			atomicMin((int *)&a[2], s_bb[0]); atomicMin((int *)&a[3], s_bb[1]); atomicMax((int *)&a[4], s_bb[2]); atomicMax((int *)&a[5], s_bb[3]);
		}
		if (nout) {atomicMax(&a[6], s_smax);}
	}
}
// one thread per tile: mzmin / mzmax fold the 16 sub-block ranges in the reference's order with its std::min / std::max (src/tiled_mesh.cpp:536-538), radius, water bbox, min_normal_z
__global__ __launch_bounds__(256) void k_tile_post_final(tile_ref_pod_t const *__restrict__ refs, uint32_t n, terra_tile_stats *__restrict__ stats, float *__restrict__ min_nz, uint32_t const *__restrict__ acc, float rad_c, float dxy, int have_normals) {
	uint32_t const t = blockIdx.x*blockDim.x + threadIdx.x;
	if (t >= n) return;
	uint32_t const *a = acc + (size_t)t*TP_ACC;
	if (stats) {
		tile_ref_pod_t const r = refs[t];
		int const x1 = r.tx*128, y1 = r.ty*128;
		float mzmin = 100.0f, mzmax = -100.0f;
		for (int k = 0; k < 16; ++k) {mzmin = min_std(mzmin, stats[t].sub_zmin[k]); mzmax = max_std(mzmax, stats[t].sub_zmax[k]);}
		stats[t].mzmin = mzmin; stats[t].mzmax = mzmax;
		stats[t].radius = (float)(0.5*sqrt((double)(rad_c + (mzmax - mzmin)*(mzmax - mzmin))));
		stats[t].wx1 = imin((int)a[2], x1 + 128); stats[t].wy1 = imin((int)a[3], y1 + 128); stats[t].wx2 = imax((int)a[4], x1); stats[t].wy2 = imax((int)a[5], y1);
	}
	if (min_nz && have_normals) { // min_normal_z = min(1.0, min over texels of dxdy/mag) = dxdy / (the largest mag), both roundings monotone; a NaN never wins (src/tiled_mesh.cpp:874)
		float smax; uint32_t const u = a[6]; memcpy(&smax, &u, 4);
		float const nz = dxy/sqrtf(smax);
		min_nz[t] = (nz < 1.0f) ? nz : 1.0f;
	}
}

// ------------------------------------------------------------------ row f1: tile AO lighting (tile_t::calc_mesh_ao_lighting, src/tiled_mesh.cpp:634-659)
// One block = one band of AO_BAND texel rows of one tile (33: four bands cover the 129 rows; with 32 a fifth block staged 74 context rows for one texel row).  The context rows the rays of the band can reach are staged in LDS in two passes:
// rays going up or sideways need context rows [y0, y0 + band + 35], rays going down rows [y0 + 36, y0 + band + 71] (context coordinates =
// texel + 36) -- 68 rows x 201 floats = 54.7 KB each time, so two blocks share a CU.  A thread owns up to 17 texels and keeps their
// attenuation sums in registers across the passes.  Same integer sums as the one-thread-per-texel version (tile_ao_simple).
constexpr unsigned AO_BAND = 33, AO_CS = 201, AO_RL = 36, AO_TEX = 129, AO_THREADS = 256, AO_ROWS_W = (AO_BAND + 3)/4, AO_HALF = (AO_BAND + AO_RL + 1)/2;
// one ray, branch-free: the eight samples sit at fixed offsets 1,3,6,...,36 steps from the texel (immediate ds_read offsets after unrolling),
// are all requested before the first compare, and the first hit is selected backwards (hit at step s attenuates by 8 - s)
template<int DX, int DY> __device__ __forceinline__ unsigned ao_march(float const *s_base, float const (&zr)[8]) {
	float smp[8];
#pragma unroll
	for (int s = 0; s < 8; ++s) {int const off = (s + 1)*(s + 2)/2; smp[s] = s_base[off*(DY*(int)AO_CS + DX)];}
	unsigned att = 0;
#pragma unroll
	for (int s = 7; s >= 0; --s) {att = (smp[s] > zr[s]) ? (unsigned)(8 - s) : att;}
	return att;
}
// the rays of one pass from the texel at sb (pass 0: up and sideways, pass 1: down)
template<int PASS> __device__ __forceinline__ unsigned ao_rays(float const *sb, float z0, float dz) {
	float zr[8];
#pragma unroll
	for (int s = 0; s < 8; ++s) {z0 += dz; zr[s] = z0;} // every ray rises by dz per step: sequential float adds, as in the reference
	if (PASS == 0) {return ao_march<-1, -1>(sb, zr) + ao_march<0, -1>(sb, zr) + ao_march<1, -1>(sb, zr) + ao_march<-1, 0>(sb, zr) + ao_march<1, 0>(sb, zr);}
	return ao_march<-1, 1>(sb, zr) + ao_march<0, 1>(sb, zr) + ao_march<1, 1>(sb, zr);
}
__device__ __forceinline__ uint8_t ao_byte(unsigned atten) {float const ao_scale = (float)(1.0 - (double)((float)atten/(float)64)); return (uint8_t)(255.0*(double)ao_scale);}
// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: xcd_ordered() renumbers the blocks so that an XCD walks a CONTIGUOUS range of logical blocks (a tile's
// four bands, whose staging passes overlap each other's rows, run on one XCD one after another).  (Measured equal for this kernel: its context rows come out of the
// infinity cache either way; kept because it is free.)
__device__ __forceinline__ unsigned xcd_ordered(unsigned b, unsigned nb) {
	unsigned const xcd = b & 7u, j = b >> 3, per = nb >> 3, rem = nb & 7u; // XCD k owns blocks k, k + 8, ...: per + (k < rem) of them
	return ((xcd < rem) ? xcd*(per + 1u) : rem*(per + 1u) + (xcd - rem)*per) + j;
}
// (Taking the tile's own zvals from `zvals` while staging, instead of the caller's copy pass into the context, was measured: the extra index work in the staging loop costs more
// than the 170 us copy kernel saves -- 1.95 vs 1.87 ms for the row.)
// The kernel is bound by its LDS reads (64 per texel).  A WAVE owns texel rows (every fourth row of the band); a lane is a column: x = lane and 64 + lane, column 128 is one
// extra pass with a lane per row (row stride 201 = 9 mod 32: distinct banks).  With texels dealt out linearly (p = tid + 256 k) a wave's 64 texels crossed a row end and
// the part behind the wrap met the part before it in the banks: half of the LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 0.53).
// OWN: the context cells inside the tile are taken from zvals (the tile's own, possibly eroded / edited heights) instead of ctx while the context is staged.
template<bool OWN> __global__ __launch_bounds__(AO_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void k_tile_ao(float const *__restrict__ zvals, float const *__restrict__ ctx, uint8_t *__restrict__ ao, float dz) {
	extern __shared__ __attribute__((aligned(16))) float s_ao_ctx[];
	unsigned const lb = xcd_ordered(blockIdx.x, gridDim.x);
	unsigned const nbands = (AO_TEX + AO_BAND - 1)/AO_BAND, t = lb/nbands, band = lb % nbands, tid = threadIdx.x;
	unsigned const w = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63u;
	unsigned const y0 = band*AO_BAND, rows = (AO_TEX - y0 < AO_BAND) ? AO_TEX - y0 : AO_BAND;
	float const *c = ctx + (size_t)t*AO_CS*AO_CS, *z = zvals + (size_t)t*130*130;
	uint8_t *out = ao + (size_t)t*AO_TEX*AO_TEX;
	unsigned attA[AO_ROWS_W], attB[AO_ROWS_W], attC = 0; // rows w, w + 4, ...: columns lane and 64 + lane; column 128 of row `lane` (wave 3, which has a row less than wave 0)
	float zA[AO_ROWS_W], zB[AO_ROWS_W], zC = 0.0f;
#pragma unroll
	for (unsigned i = 0; i < AO_ROWS_W; ++i) {
		unsigned const yl = w + 4*i;
		attA[i] = attB[i] = 0;
		float const *zr = z + (size_t)(y0 + ((yl < rows) ? yl : 0u))*130;
		zA[i] = zr[lane]; zB[i] = zr[64 + lane];
	}
	if (w == 3) {zC = z[(size_t)(y0 + ((lane < rows) ? lane : 0u))*130 + 128];}
	// Staging: a pass's rows (<= 69 x 201 floats), thread = context column (threads 201 .. 255 idle), the rows in two halves of <= 35 loads per thread, each half ALL in
	// flight before its first LDS store (as a plain copy loop, a few loads at a time between LDS stores, two blocks per CU kept ~8 KB in flight per CU: 775 -> 740 us).
	// By rows, which source a cell has (OWN) is one compare per row and one per thread.  Measured for 4096 tiles: linear staging (thread = element i, i + 256, ...) without OWN
	// 663 us + the 168 us copy pass it needs = 831; this form 763; linear staging with an index division per element for OWN: 2813 (its 55 unrolled divisions spill).
	unsigned const nrow = rows + AO_RL;
	bool const col_ok = tid < AO_CS, col_own = OWN && (tid - AO_RL) < 130u;
	for (int pass = 0; pass < 2; ++pass) {
		unsigned const row0 = pass ? y0 + AO_RL : y0;
#pragma unroll
		for (unsigned h = 0; h < 2; ++h) {
			float stg[AO_HALF];
#pragma unroll
			for (unsigned k = 0; k < AO_HALF; ++k) {
				unsigned const r = h*AO_HALF + k, gr = row0 + r;
				bool const ok = col_ok && r < nrow;
				float const *p = c + (size_t)(ok ? gr : 0u)*AO_CS + (col_ok ? tid : 0u);
				if (OWN) {bool const in = col_own && (gr - AO_RL) < 130u && ok; p = in ? z + (size_t)(gr - AO_RL)*130 + (tid - AO_RL) : p;}
				stg[k] = *p;
			}
			if (h == 0) {__syncthreads();} // everybody is done with the previous pass's rows
#pragma unroll
			for (unsigned k = 0; k < AO_HALF; ++k) {unsigned const r = h*AO_HALF + k; if (col_ok && r < nrow) {s_ao_ctx[r*AO_CS + tid] = stg[k];}}
		}
		__syncthreads();
		unsigned const lrow = pass ? 0u : AO_RL; // LDS row of the band's first texel row
#pragma unroll
		for (unsigned i = 0; i < AO_ROWS_W; ++i) {
			unsigned const yl = w + 4*i;
			if (yl >= rows) break; // (wave-uniform)
			float const *sb = s_ao_ctx + (yl + lrow)*AO_CS + AO_RL + lane; // the texel itself in the staged context
			if (pass == 0) {attA[i] += ao_rays<0>(sb, zA[i], dz); attB[i] += ao_rays<0>(sb + 64, zB[i], dz);}
			else           {attA[i] += ao_rays<1>(sb, zA[i], dz); attB[i] += ao_rays<1>(sb + 64, zB[i], dz);}
		}
		if (w == 3) {
			float const *sb = s_ao_ctx + (((lane < rows) ? lane : 0u) + lrow)*AO_CS + AO_RL + 128;
			attC += (pass == 0) ? ao_rays<0>(sb, zC, dz) : ao_rays<1>(sb, zC, dz);
		}
	}
#pragma unroll
	for (unsigned i = 0; i < AO_ROWS_W; ++i) {
		unsigned const yl = w + 4*i;
		if (yl >= rows) break;
		uint8_t *o = out + (size_t)(y0 + yl)*AO_TEX;
		o[lane] = ao_byte(attA[i]); o[64 + lane] = ao_byte(attB[i]);
	}
	if (w == 3 && lane < rows) {out[(size_t)(y0 + lane)*AO_TEX + 128] = ao_byte(attC);}
}

// ---- the whole tile's context in LDS (k_tile_ao_tile): one block of 16 waves per tile, the 201 x 201 context staged ONCE -- 201 rows x 202 floats = 162 408 of the CU's
// 163 840 bytes -- instead of 68 rows twice per 33-row band (2.7 x 201 rows per tile, and the staging of a band was as long as its rays).  A lane owns the texel PAIR
// (2 lane, 2 lane + 1) of a row, a wave a row: the two texels' samples of a ray step are neighbours in LDS, one 8-byte read (256 B per clock against the 4-byte read's 128)
// where the step's x offset is even; where it is odd the pair straddles two aligned 8-byte words and takes both (the 4-byte reads of a lane pair would be 2-way bank
// conflicts at a stride of two dwords).  Per texel pair 40 + 2 x 24 eight-byte reads = 176 LDS cycles instead of 256, four waves per SIMD instead of two to hide them;
// the compares (2 instructions per sample) are what is left.  Column 128 is a lane per row.  Same integer sums as k_tile_ao / tile_ao_simple.
constexpr unsigned AOT_THREADS = 1024, AOT_S = 202, AOT_LDS = AO_CS*AOT_S*4;
typedef float ao_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float ao_lds_f;   // (the volatile reads below must know they are LDS reads: a generic volatile pointer becomes a flat load)
typedef __attribute__((address_space(3))) ao_f2 ao_lds_f2;
// Half a ray of a texel pair (steps 4 H .. 4 H + 3): the unit the row loop keeps two of in flight.  sb: the sample 36 steps up and left of the pair's first texel -- an even
// dword: every offset below is non-negative and known at compile time.  volatile LDS reads: each stays ONE ds_read_b64 -- the compiler otherwise pairs them into ds_read2_b64,
// half the rate per byte, and narrows the odd case to 4-byte reads -- and they keep their place between the `pins` of ao_pair_hits4 (an empty asm volatile on the ray's sums):
// all eight rays' reads moved to the front spill at 128 registers.  (One whole ray in flight and two half rays in flight measure the same, 382 / 380 us: with four waves per
// SIMD the LDS latency is hidden either way.)
template<int DX, int DY, int H> __device__ __forceinline__ void ao_pair_load4(ao_lds_f const *sb, float (&a)[4], float (&b)[4]) {
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		int const s = 4*H + i, T = (s + 1)*(s + 2)/2, off = ((int)AO_RL + T*DY)*(int)AOT_S + (int)AO_RL + T*DX;
		if ((off & 1) == 0) {ao_f2 const v = *(ao_lds_f2 const volatile *)(sb + off); a[i] = v.x; b[i] = v.y;}
		else {ao_f2 const u = *(ao_lds_f2 const volatile *)(sb + off - 1), v = *(ao_lds_f2 const volatile *)(sb + off + 1); a[i] = u.y; b[i] = v.x;}
	}
}
template<int H> __device__ __forceinline__ void ao_pair_hits4(float const (&a)[4], float const (&b)[4], float const (&zr0)[8], float const (&zr1)[8], unsigned &r0, unsigned &r1) {
#pragma unroll
	for (int i = 3; i >= 0; --i) {int const s = 4*H + i; r0 = (a[i] > zr0[s]) ? (unsigned)(8 - s) : r0; r1 = (b[i] > zr1[s]) ? (unsigned)(8 - s) : r1;}
	asm volatile("" : "+v"(r0), "+v"(r1)); // the pin
}
template<int DX, int DY> __device__ __forceinline__ unsigned ao_march_one(float const *sb, float const (&zr)[8]) { // one texel, context rows AOT_S apart
	float smp[8];
#pragma unroll
	for (int s = 0; s < 8; ++s) {int const T = (s + 1)*(s + 2)/2; smp[s] = sb[((int)AO_RL + T*DY)*(int)AOT_S + (int)AO_RL + T*DX];}
	unsigned att = 0;
#pragma unroll
	for (int s = 7; s >= 0; --s) {att = (smp[s] > zr[s]) ? (unsigned)(8 - s) : att;}
	return att;
}
// Staging: thread = context column, the block's four 256-thread quarters take every fourth row: 51 values per thread.  They are LOADED into registers while the tile before is
// still being computed (the block is persistent: tiles blockIdx.x, + gridDim.x, ...) and stored to LDS when its rays are done: a CU holds one workgroup (the LDS is full), so
// nobody else would hide the loads.  Which source a cell has (OWN: the tile's own heights inside the tile) is a scalar test per row and a per-thread base + stride.
constexpr unsigned AOT_STG = 51;
template<bool OWN> __device__ __forceinline__ void aot_stage_load(float const *__restrict__ zvals, float const *__restrict__ ctx, unsigned t, unsigned col, unsigned rq, float (&stg)[AOT_STG]) {
	float const *c = ctx + (size_t)t*AO_CS*AO_CS, *z = zvals + (size_t)t*130*130;
	bool const col_ok = col < AO_CS, col_own = OWN && (col - AO_RL) < 130u;
	float const *base_out = c + (col_ok ? col : 0u);
	float const *base_in = col_own ? z + (col - AO_RL) - (size_t)AO_RL*130 : base_out; // centre rows: r*130 from here is row r - 36 of the tile
	unsigned const stride_in = col_own ? 130u : AO_CS;
	// rows rq + 4k: for k = 9 .. 40 they lie inside the tile whatever rq is (36 .. 163 + rq), for k = 41 (164 .. 167) it depends on rq, the others never do -- no
	// per-element choice between two pointers but for that one
#pragma unroll
	for (unsigned k = 0; k < AOT_STG; ++k) {
		unsigned const r = rq + 4u*k, rr = (r < AO_CS) ? r : 0u; // (scalar: rq is wave-uniform)
		float const *p;
		if (OWN && k >= 9 && k <= 40) {p = base_in + __umul24(rr, stride_in);}
		else if (OWN && k == 41) {p = ((rr - AO_RL) < 130u) ? base_in + __umul24(rr, stride_in) : base_out + rr*AO_CS;}
		else {p = base_out + rr*AO_CS;}
		stg[k] = *p;
	}
}
__device__ __forceinline__ void aot_stage_store(float *s, unsigned col, unsigned rq, float const (&stg)[AOT_STG]) {
	if (col >= AO_CS) return;
#pragma unroll
	for (unsigned k = 0; k < AOT_STG; ++k) {unsigned const r = rq + 4u*k; if (r < AO_CS) {s[r*AOT_S + col] = stg[k];}}
}
template<bool OWN> __global__ __launch_bounds__(AOT_THREADS) void k_tile_ao_tile(float const *__restrict__ zvals, float const *__restrict__ ctx, uint8_t *__restrict__ ao, float dz, unsigned n) {
	extern __shared__ __attribute__((aligned(16))) float s_aot[];
	__shared__ unsigned s_item;
	unsigned const tid = threadIdx.x, lane = tid & 63u, col = tid & 255u;
	unsigned const w = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), rq = w >> 2;
	float stg[AOT_STG];
	if (blockIdx.x < n) {aot_stage_load<OWN>(zvals, ctx, blockIdx.x, col, rq, stg);}
	for (unsigned t = blockIdx.x; t < n; t += gridDim.x) {
	float const *z = zvals + (size_t)t*130*130;
	uint8_t *out = ao + (size_t)t*AO_TEX*AO_TEX;
	aot_stage_store(s_aot, col, rq, stg);
	if (tid == 0) {s_item = 0u;}
	__syncthreads();
	// in flight behind this tile's rays -- which then must not wait for a global load of their own (a wave's loads return in order): with OWN the texels' heights are the
	// centre of the staged context; without, they are the caller's zvals (not the context's centre) and the next tile is loaded after the rays instead
	if (OWN && t + gridDim.x < n) {aot_stage_load<OWN>(zvals, ctx, t + gridDim.x, col, rq, stg);}
	for (;;) { // work items of the tile, dealt out by a counter in LDS (a fixed deal leaves one wave with a ninth row while the others wait at the barrier): 129 rows of 64 texel pairs, then column 128 in three pieces
		unsigned item = 0;
		if (lane == 0) {item = atomicAdd(&s_item, 1u);}
		item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
		if (item >= AO_TEX) {
			if (item >= AO_TEX + 3u) break;
			unsigned const y = (item - AO_TEX)*64u + lane; // column 128: a lane per row
			if (y < AO_TEX) {
				float z0 = OWN ? s_aot[(y + AO_RL)*AOT_S + AO_RL + 128] : z[(size_t)y*130 + 128];
				float zr[8];
#pragma unroll
				for (int s = 0; s < 8; ++s) {z0 += dz; zr[s] = z0;}
				float const *sb = s_aot + y*AOT_S + 128;
				unsigned const att = ao_march_one<-1, -1>(sb, zr) + ao_march_one<0, -1>(sb, zr) + ao_march_one<1, -1>(sb, zr) + ao_march_one<-1, 0>(sb, zr) + ao_march_one<1, 0>(sb, zr)
					+ ao_march_one<-1, 1>(sb, zr) + ao_march_one<0, 1>(sb, zr) + ao_march_one<1, 1>(sb, zr);
				out[(size_t)y*AO_TEX + 128] = ao_byte(att);
			}
			continue;
		}
		unsigned const y = item, x0 = 2u*lane;
		float z0, z1;
		if (OWN) {ao_f2 const v = *(ao_f2 const *)(s_aot + (y + AO_RL)*AOT_S + AO_RL + x0); z0 = v.x; z1 = v.y;}
		else {z0 = z[(size_t)y*130 + x0]; z1 = z[(size_t)y*130 + x0 + 1];}
		float zr0[8], zr1[8];
#pragma unroll
		for (int s = 0; s < 8; ++s) {z0 += dz; z1 += dz; zr0[s] = z0; zr1[s] = z1;} // every ray rises by dz per step: sequential float adds, as in the reference
		ao_lds_f const *sb = (ao_lds_f const *)s_aot + y*AOT_S + x0;
		unsigned a0 = 0, a1 = 0;
		// two HALF rays in flight: the far half (steps 4 .. 7, scanned first: the nearest hit wins) of a ray is loaded behind the near half of the ray before it
		float pa[4], pb[4], qa[4], qb[4];
		unsigned r0 = 0, r1 = 0;
#define TERRA_AO_RAY(DX, DY, NEXT) ao_pair_load4<DX, DY, 0>(sb, qa, qb); ao_pair_hits4<1>(pa, pb, zr0, zr1, r0, r1); NEXT; ao_pair_hits4<0>(qa, qb, zr0, zr1, r0, r1); a0 += r0; a1 += r1; r0 = r1 = 0;
		ao_pair_load4<-1, -1, 1>(sb, pa, pb);
		TERRA_AO_RAY(-1, -1, (ao_pair_load4< 0, -1, 1>(sb, pa, pb)))
		TERRA_AO_RAY( 0, -1, (ao_pair_load4< 1, -1, 1>(sb, pa, pb)))
		TERRA_AO_RAY( 1, -1, (ao_pair_load4<-1,  0, 1>(sb, pa, pb)))
		TERRA_AO_RAY(-1,  0, (ao_pair_load4< 1,  0, 1>(sb, pa, pb)))
		TERRA_AO_RAY( 1,  0, (ao_pair_load4<-1,  1, 1>(sb, pa, pb)))
		TERRA_AO_RAY(-1,  1, (ao_pair_load4< 0,  1, 1>(sb, pa, pb)))
		TERRA_AO_RAY( 0,  1, (ao_pair_load4< 1,  1, 1>(sb, pa, pb)))
		TERRA_AO_RAY( 1,  1, (void)0)
#undef TERRA_AO_RAY
		uint8_t *o = out + (size_t)y*AO_TEX + x0;
		o[0] = ao_byte(a0); o[1] = ao_byte(a1);
	}
	if (!OWN && t + gridDim.x < n) {aot_stage_load<OWN>(zvals, ctx, t + gridDim.x, col, rq, stg);}
	__syncthreads(); // (the next tile's staging overwrites the context)
	}
}

// ------------------------------------------------------------------ row f2: mesh shadows, one launch per dependency level
// An outgoing edge height travels between tiles as (dependency order << 32 | float bits) under a 64-bit max; 0 = nothing arrived (MESH_MIN_Z).
constexpr unsigned long long SHADOW_EDGE_PUB = 1ull << 63; // (k_tile_shadows_flow) this word of a finished tile's edge array has been published
struct shadow_edge_t {
	__device__ static float decode(unsigned long long v) {if (v == 0) return -1.0E6f; uint32_t const b = (uint32_t)(v & 0xFFFFFFFFull); float f; memcpy(&f, &b, 4); return f;}
	__device__ static unsigned long long pack(uint32_t order, float v) {uint32_t b; memcpy(&b, &v, 4); return ((unsigned long long)order << 32) | b;}
};
constexpr unsigned SH_LEVEL_THREADS = 576; // 9 waves: all 520 sweeps of a tile in one round

// One dependency level of the tile mesh shadows: one block per tile of the level, one thread per sweep.  A sweep is a chain of ~260 dependent steps, so
// what a step costs is latency: everything it reads AND writes lives in LDS -- the tile's heights, the two incoming edge arrays (decoded), the tile's
// shadow bytes and its two outgoing edge arrays (ordered 64-bit max, as in the other variants).  The walk touches no global memory at all; with the
// per-step shadow atomics going to L2 a step took ~270 ns (a wave's atomics drain one after another), with global reads in the loop it also stalled on
// `s_waitcnt vmcnt(0)` every step.  The block copies its results out at the end: it is the only writer of its tile's mask and edges.
struct shadow_lds_in_t {
	float const *ix, *iy; // LDS: decoded incoming edge heights (MESH_MIN_Z where there is none)
	__device__ float x(int i) const {return ix[i];}
	__device__ float y(int i) const {return iy[i];}
};
struct shadow_lds_out_t {
	uint8_t *sm; unsigned long long *ox, *oy; int xsize; // all LDS
	int spare_b; unsigned long long *spare_q; // this lane's spare byte (an index into sm) and spare 8-byte slot: where the branch-free steps write when they have nothing to write
	__device__ void shadow(int x, int y) {sm[y*xsize + x] = 0x02;} // MESH_SHADOW: every writer stores the same byte
	__device__ void shadow_at(int idx) {sm[idx] = 0x02;}            // (the mask and the heights have the same row length)
	__device__ void shadow_if(bool sh, int idx) {sm[sh ? idx : spare_b] = 0x02;} // branch-free: a selected address instead of a skipped store
	__device__ void out_x_if(bool on, int i, unsigned long long v) {atomicMax(on ? &ox[i] : spare_q, v);}
	__device__ void out_y_if(bool on, int i, unsigned long long v) {atomicMax(on ? &oy[i] : spare_q, v);}
	__device__ void out_x(int i, uint32_t order, float v) {atomicMax(&ox[i], shadow_edge_t::pack(order, v));}
	__device__ void out_y(int i, uint32_t order, float v) {atomicMax(&oy[i], shadow_edge_t::pack(order, v));}
};
// The sweep of the LDS kernels (same arithmetic as shadow_trace_path, which the per-thread cross-check kernels and the emulator keep): a tile is a chain of <= 131 dependent
// steps per sweep and the batch a chain of tiles, so what a step costs is what the row costs.  Cut out of the step: the incoming edge heights are looked at only while a sweep
// is still on its first column / row (x and y move away from xa / ya monotonically: a wave-uniform branch that is not taken after the first steps), the outgoing ones only
// where a shadowed step sits on the last column / row; the cell index and the coordinate along the light's dominant axis are carried instead of multiplied out; the last
// unshadowed height is carried as the double it is used as; the rest of the step is selects, not branches.
__device__ __forceinline__ int shadow_wave_max(int v) {for (int off = 32; off; off >>= 1) {v = max(v, __shfl_xor(v, off));} return __builtin_amdgcn_readfirstlane(v);} // (all 64 lanes active; the result in a scalar register: loop bounds)
// A lane's sweep and its wave's phases: the same for every tile of a batch (the light and the tile geometry are), so a persistent block makes it once, not per tile (the line
// clip with its divisions and three wave reductions were ~10 % of a tile's time).  Called by every lane of a wave, sweep or not (p >= npaths: none): the phases are the wave's.
struct shadow_plan_t {shadow_path_t w; unsigned p; bool has; int n1, e, lmax;};
__device__ __forceinline__ shadow_plan_t shadow_sweep_plan(shadow_consts_t const &c, unsigned p, unsigned npaths) {
	shadow_plan_t pl;
	pl.w = shadow_path_t{0, 0, 0, 0, -1, 0, 0, 0, 0, 0}; pl.p = p;
	pl.has = p < npaths && shadow_path_setup(c, p, pl.w);
	if (!pl.has) {pl.w.longest = -1;}
	// x and y move away from xa / ya and toward xb / yb monotonically, and where a walk leaves its first column / row and reaches its last is known in closed form
	// (shadow_path_zones): three phases with wave-uniform trip counts -- until the last lane has left its first zone; the middle, which no lane's edge words can touch and whose
	// cells lie strictly inside the tile; from where the first lane reaches its last zone
	int first_end = 0, last_begin = 0x7FFFFFFF;
	if (pl.has) {shadow_path_zones(pl.w.longest, pl.w.shortest, first_end, last_begin);}
	pl.n1 = shadow_wave_max(first_end); pl.e = -shadow_wave_max(-last_begin); pl.lmax = shadow_wave_max(pl.w.longest);
	return pl;
}
template<class IN, class OUT> __device__ __forceinline__ void shadow_trace_path_lean(shadow_consts_t const &c, float const *mh, IN const &in, shadow_plan_t const &pl, OUT &out) {
	if (!pl.has) return;
	shadow_path_t const &w = pl.w; unsigned const p = pl.p;
	int const xa = w.xa, ya = w.ya, xb = w.xb, yb = w.yb, longest = w.longest, shortest = w.shortest, dx1 = w.dx1, dy1 = w.dy1, dx2 = w.dx2, dy2 = w.dy2;
	bool const dim = (fabsf(c.dirx) < fabsf(c.diry));
	double const dir_ratio = (double)(c.dirz/(dim ? c.diry : c.dirx));
	float const org_d = dim ? -c.Y_SCENE_SIZE : -c.X_SCENE_SIZE, step_d = dim ? c.DY_VAL : c.DX_VAL;
	int x = xa, y = ya, numerator = longest >> 1;
	int const xs = c.xsize, di1 = dy1*xs + dx1, di2 = dy2*xs + dx2, dc1 = dim ? dy1 : dx1, dc2 = dim ? dy2 : dx2, last_cell = xs*c.ysize - 1;
	int idx = y*xs + x, cc = dim ? y : x, i = 0; // the cell, its coordinate along the dominant light axis, the step
	float cur_d; double cur_zd; // the caster (the last unshadowed point: its coordinate along the light's axis, its height)
	float nxt_z = mh[min(max(idx, 0), last_cell)]; // the height of the cell a step works on is read one step ahead: its address does not depend on the shadow state
	// One step of the first or the last phase (on the first / last column or row of the walk: edge heights come in / go out, the cell may lie just outside the tile), straight-line:
	// lanes whose sweep has ended and cells outside the tile are folded into `valid`, `not yet inited` into the caster (a caster at -inf shadows nothing: the step then takes the
	// cell as the new caster, which is what the reference's `inited` does), the incoming edge heights are read whether needed or not and selected, the outgoing ones go through LDS
	// atomics on selected addresses (the lane's spare slot when the step has nothing to hand on).  EDGE_IN: some lane may still be on its first column / row.
	cur_d = 0.0f; cur_zd = -(double)INFINITY;
	auto zone_step = [&](auto in_tag) {
		constexpr bool EDGE_IN = decltype(in_tag)::value;
		int const x0 = x, y0 = y, idx0 = idx, cc0 = cc, i0 = i;
		float const pt_z = nxt_z;
		double const pt_zd = (double)pt_z;
		numerator += shortest;
		bool const both = numerator >= longest;
		numerator -= both ? longest : 0;
		x += both ? dx1 : dx2; y += both ? dy1 : dy2; idx += both ? di1 : di2; cc += both ? dc1 : dc2; ++i;
		nxt_z = mh[min(max(idx, 0), last_cell)];
		bool const valid = i0 <= longest && (unsigned)x0 < (unsigned)c.xsize && (unsigned)y0 < (unsigned)c.ysize;
		float const pt_d = org_d + step_d*(float)cc0;
		if (EDGE_IN) {
			float const siy = in.y(min(max(y0, 0), c.ysize - 1)), six = in.x(min(max(x0, 0), c.xsize - 1));
			bool const take_y = valid && x0 == xa && siy > -1.0E6f, take_x = valid && !take_y && y0 == ya && six > -1.0E6f;
			float const siv = take_y ? siy : six;
			cur_d = (take_y || take_x) ? pt_d : cur_d; cur_zd = (take_y || take_x) ? (double)siv : cur_zd;
		}
		float const shadow_z = (float)((double)(pt_d - cur_d)*dir_ratio + cur_zd);
		bool const sh = valid && shadow_z > pt_z;
		out.shadow_if(sh, idx0);
		unsigned long long const word = shadow_edge_t::pack(p*1024u + (uint32_t)i0 + 1u, shadow_z);
		out.out_y_if(sh && x0 == xb, y0, word);
		out.out_x_if(sh && y0 == yb, x0, word);
		bool const upd = valid && !sh;
		cur_d = upd ? pt_d : cur_d; cur_zd = upd ? pt_zd : cur_zd;
	};
	int const n1 = pl.n1, e = pl.e, lmax = pl.lmax; // the wave's phases (shadow_sweep_plan): the middle has no loop test, no zone test, no bounds test per step -- the wave's lanes all have a step to make
	int k = 0;
	for (; k < n1; ++k) {zone_step(std::true_type());}
	if (k < e) { // the middle: straight-line steps -- no x / y (recomputed behind it), no bounds, `not yet inited` folded into the caster (a caster at -inf shadows nothing), the
		// shadow byte stored with a selected address instead of a branch
		int const k0 = k;
		for (; k < e; ++k) {
			int const idx0 = idx, cc0 = cc;
			float const pt_z = nxt_z;
			double const pt_zd = (double)pt_z;
			numerator += shortest;
			bool const both = numerator >= longest;
			numerator -= both ? longest : 0;
			idx += both ? di1 : di2; cc += both ? dc1 : dc2;
			nxt_z = mh[idx]; // (a walk's cell: at worst one column / row past the tile, still this block's LDS)
			float const pt_d = org_d + step_d*(float)cc0;
			float const shadow_z = (float)((double)(pt_d - cur_d)*dir_ratio + cur_zd);
			bool const sh = shadow_z > pt_z;
			out.shadow_if(sh, idx0);
			cur_d = sh ? cur_d : pt_d; cur_zd = sh ? cur_zd : pt_zd;
		}
		i += k - k0;
		int const m = (longest > 0) ? ((longest >> 1) + i*shortest)/longest : 0; // how often the minor coordinate has moved in i steps
		x = xa + dx2*i + (dx1 - dx2)*m; y = ya + dy2*i + (dy1 - dy2)*m;
	}
	for (; k <= lmax; ++k) {zone_step(std::false_type());}
}
struct shadow_lanes_t {uint16_t path[SH_LEVEL_THREADS];}; // which sweep a lane takes (shadow_lane_order), 0xFFFF = none; travels as a kernel argument
// LDS of the shadow kernels: the shadow bytes FIRST (the sweep's byte stores and height reads then carry their array's offset as the instruction's 16-bit immediate: behind
// 70 KB of heights the mask's offset did not fit and cost an addition per step: 1.71 -> 1.65 ms), a spare byte per lane behind them, the heights (16-byte aligned), the
// decoded incoming edges, the outgoing edge words, a spare 8-byte slot per lane.  (The cell's coordinate along the light's axis from a 132-entry table in LDS instead of a
// conversion, a multiplication and an addition per step was measured and LOST, 1.65 -> 1.72 ms: a second LDS read per step to wait for.)
constexpr unsigned SH_SPARE_B = 130*130, SH_OFF_MH = (130*130 + SH_LEVEL_THREADS + 15)/16*16; // byte offsets: the lanes' spare bytes, the heights
constexpr unsigned SH_LEVEL_LDS = SH_OFF_MH + 130*130*4 + 2*130*4 + 2*130*8 + SH_LEVEL_THREADS*8; // = 17 488 + 67 600 + 1 040 + 2 080 + 4 608 = 92 816 bytes
__global__ __launch_bounds__(SH_LEVEL_THREADS) void k_tile_shadows_level(shadow_consts_t c, uint32_t n, uint32_t const *__restrict__ order, int32_t const *__restrict__ adj,
	float const *__restrict__ zvals, unsigned long long *out, uint8_t *smask, uint32_t npaths, shadow_lanes_t lanes)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t s_sh_raw[];
	unsigned const zv = 130, tid = threadIdx.x;
	uint32_t const t = order[blockIdx.x];
	int32_t const ax = adj[2*t], ay = adj[2*t + 1];
	float const *z = zvals + (size_t)t*zv*zv;
	uint32_t *s_mask = (uint32_t *)s_sh_raw;
	float *s_sh_mh = (float *)(s_sh_raw + SH_OFF_MH), *s_in = s_sh_mh + zv*zv;
	unsigned long long *s_out = (unsigned long long *)(s_in + 2*zv); // (8-byte aligned: 17 488 + 67 600 + 1 040)
	if (((uintptr_t)z & 15) == 0) {for (unsigned i = tid; i < zv*zv/4; i += SH_LEVEL_THREADS) {((float4 *)s_sh_mh)[i] = ((float4 const *)z)[i];}} // 67 600 bytes per tile
	else {for (unsigned i = tid; i < zv*zv; i += SH_LEVEL_THREADS) {s_sh_mh[i] = z[i];}}
	for (unsigned i = tid; i < zv*zv/4; i += SH_LEVEL_THREADS) {s_mask[i] = 0u;}
	if (tid < 2*zv) { // in.x(i) = the y-neighbour's out_x, in.y(i) = the x-neighbour's out_y (src/tiled_mesh.cpp:676-687); earlier levels have finished
		bool const isx = tid < zv; unsigned const i = isx ? tid : tid - zv; int32_t const a = isx ? ay : ax;
		s_in[tid] = (a >= 0) ? shadow_edge_t::decode(out[((size_t)(isx ? 0 : 1)*n + a)*zv + i]) : -1.0E6f;
		s_out[tid] = 0ull; // 0 = never written
	}
	__syncthreads();
	shadow_lds_in_t const in{s_in, s_in + zv};
	shadow_lds_out_t o{(uint8_t *)s_mask, s_out, s_out + zv, (int)zv, (int)(SH_SPARE_B + tid), s_out + 2*zv + tid};
	shadow_trace_path_lean(c, s_sh_mh, in, shadow_sweep_plan(c, lanes.path[tid], npaths), o);
	__syncthreads();
	uint32_t *gm = (uint32_t *)(smask + (size_t)t*zv*zv); // 16 900 bytes per tile: word-aligned
	for (unsigned i = tid; i < zv*zv/4; i += SH_LEVEL_THREADS) {gm[i] = s_mask[i] | c.mask_fill;} // plain stores: nobody else writes this tile's mask
	if (tid < 2*zv) {
		unsigned long long const v = s_out[tid];
		if (v) {out[((size_t)((tid < zv) ? 0 : 1)*n + t)*zv + ((tid < zv) ? tid : tid - zv)] = v;}
	}
}

// The whole batch as ONE dataflow launch.  A tile needs the edge arrays of its two neighbours toward the light (src/tiled_mesh.cpp:664-692) and nothing else: a launch per
// dependency LEVEL (127 of them for a 64 x 64 batch) makes every tile wait for the slowest tile of the level before -- and for two launch gaps.  Here blocks take tiles from
// a queue in level order (an atomic ticket; `order` lists a tile's dependencies before it), stage the tile's heights, and only then wait for the upstream tiles' `done`
// words.  Progress: a waiting block holds a ticket, the tile it waits for has a LOWER ticket and was therefore taken by a block that is running -- whatever the grid size and
// however many blocks are resident; the lowest ticket in flight never waits.
// Hand-off between workgroups (they sit on other CUs, maybe other XCDs: neither their L1 nor their L2 is ours): the edge words ARE the flags.  A finished tile stores ALL
// 260 words of its two edge arrays as 8-byte agent-scope (write-through) stores with SHADOW_EDGE_PUB set -- `nothing arrived at this cell` is published as 0 | PUB -- and a
// consumer lane polls exactly the word it needs with relaxed agent-scope loads until the bit is there: no flag word, no fence, no second round trip (a `done` word per tile
// with a release / acquire pair was measured first: ~5 us more per tile of a 127-tile chain).  While a tile is far from its turn only ONE lane per upstream tile polls,
// sleeping in between; the 260 lanes join when that lane has seen the first word.  `ticket` is zeroed by the caller per call, the edge arrays too (no stale PUB bit);
// the virtual halo slots (index >= ntiles) were written before the launch and are read without polling.
__global__ __launch_bounds__(SH_LEVEL_THREADS) void k_tile_shadows_flow(shadow_consts_t c, uint32_t n, uint32_t ntiles, uint32_t const *__restrict__ order, int32_t const *__restrict__ adj,
	float const *__restrict__ zvals, unsigned long long *out, uint8_t *smask, uint32_t npaths, uint32_t *ticket, shadow_lanes_t lanes)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t s_sh_raw[];
	__shared__ uint32_t s_ticket;
	unsigned const zv = 130, tid = threadIdx.x;
	uint32_t *s_mask = (uint32_t *)s_sh_raw;
	float *s_sh_mh = (float *)(s_sh_raw + SH_OFF_MH), *s_in = s_sh_mh + zv*zv;
	unsigned long long *s_out = (unsigned long long *)(s_in + 2*zv);
	shadow_plan_t const plan = shadow_sweep_plan(c, lanes.path[tid], npaths); // (the same for every tile)
	for (;;) {
		if (tid == 0) {s_ticket = atomicAdd(ticket, 1u);}
		__syncthreads();
		uint32_t const k = s_ticket;
		if (k >= ntiles) return; // (block-uniform)
		uint32_t const t = order[k];
		int32_t const ax = adj[2*t], ay = adj[2*t + 1];
		float const *z = zvals + (size_t)t*zv*zv;
		// everything that does not depend on the neighbours first
		if (((uintptr_t)z & 15) == 0) {for (unsigned i = tid; i < zv*zv/4; i += SH_LEVEL_THREADS) {((float4 *)s_sh_mh)[i] = ((float4 const *)z)[i];}}
		else {for (unsigned i = tid; i < zv*zv; i += SH_LEVEL_THREADS) {s_sh_mh[i] = z[i];}}
		for (unsigned i = tid; i < zv*zv/4; i += SH_LEVEL_THREADS) {s_mask[i] = 0u;}
		if (tid < 64) { // far from this tile's turn: lane 0 watches the x neighbour's first word, lane 1 the y neighbour's
			int32_t const a = (tid == 0) ? ax : ((tid == 1) ? ay : -1);
			unsigned long long const *w = (a >= 0 && (uint32_t)a < ntiles) ? &out[((size_t)((tid == 0) ? 1 : 0)*n + a)*zv] : nullptr;
			for (;;) {
				bool const ready = !w || (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SHADOW_EDGE_PUB) != 0ull;
				if (__all(ready)) break;
				__builtin_amdgcn_s_sleep(8);
			}
		}
		__syncthreads();
		if (tid < 2*zv) { // in.x(i) = the y-neighbour's out_x, in.y(i) = the x-neighbour's out_y (src/tiled_mesh.cpp:676-687): every lane waits for its own word
			bool const isx = tid < zv; unsigned const i = isx ? tid : tid - zv; int32_t const a = isx ? ay : ax;
			float v = -1.0E6f;
			if (a >= 0) {
				unsigned long long const *w = &out[((size_t)(isx ? 0 : 1)*n + a)*zv + i];
				unsigned long long e = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((uint32_t)a < ntiles) {while (!(e & SHADOW_EDGE_PUB)) {__builtin_amdgcn_s_sleep(1); e = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);}}
				v = shadow_edge_t::decode(e & ~SHADOW_EDGE_PUB);
			}
			s_in[tid] = v;
			s_out[tid] = 0ull; // 0 = never written
		}
		__syncthreads();
		shadow_lds_in_t const in{s_in, s_in + zv};
		shadow_lds_out_t o{(uint8_t *)s_mask, s_out, s_out + zv, (int)zv, (int)(SH_SPARE_B + tid), s_out + 2*zv + tid};
		shadow_trace_path_lean(c, s_sh_mh, in, plan, o);
		__syncthreads();
		if (tid < 2*zv) { // the edges first, every word: somebody may be polling it
			__hip_atomic_store(&out[((size_t)((tid < zv) ? 0 : 1)*n + t)*zv + ((tid < zv) ? tid : tid - zv)], s_out[tid] | SHADOW_EDGE_PUB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		uint32_t *gm = (uint32_t *)(smask + (size_t)t*zv*zv); // 16 900 bytes per tile: word-aligned; read by later launches only
		for (unsigned i = tid; i < zv*zv/4; i += SH_LEVEL_THREADS) {gm[i] = s_mask[i] | c.mask_fill;}
		__syncthreads(); // (the next tile's staging overwrites the mask words being copied out)
	}
}

// ------------------------------------------------------------------ min / max reduction (run_erosion's min(vals), get_heightmap_z_range): HBM-bound, 4 B read per cell
// Every thread keeps MM_UNROLL independent 16-byte loads in flight per trip (the loop-carried state is only the running min / max), the grid is sized
// to the chip (256 CUs x 8 blocks of 256 threads) so a 16384^2 grid is ~8 trips of 64 bytes per thread; wave shuffle reduction, then look-before-atomic
// like the fused variant.  d[0] = min f2ord(v), d[1] = min ~f2ord(v); NaNs are skipped.
constexpr int MM_UNROLL = 4;
__device__ __forceinline__ void minmax_acc4(st_f4 const v, float &lo, float &hi) {
	// fminf / fmaxf drop NaNs (v_min_f32 / v_max_f32 with IEEE mode return the non-NaN operand): the same set of values as the `v == v` filter
	lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
	hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
}
__global__ __launch_bounds__(256) void k_minmax(float const *__restrict__ vals, size_t n, uint32_t *__restrict__ d) {
	float lo = INFINITY, hi = -INFINITY;
	bool any = false; // +-inf are legitimate values: remember whether anything but NaNs was seen
	size_t const n4 = n/4, stride = (size_t)gridDim.x*blockDim.x;
	st_f4 const *v4 = (st_f4 const *)vals;
	size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	for (; i + (MM_UNROLL - 1)*stride < n4; i += MM_UNROLL*stride) {
		st_f4 r[MM_UNROLL];
#pragma unroll
		for (int u = 0; u < MM_UNROLL; ++u) {r[u] = __builtin_nontemporal_load(&v4[i + u*stride]);}
#pragma unroll
		for (int u = 0; u < MM_UNROLL; ++u) {minmax_acc4(r[u], lo, hi);}
	}
	for (; i < n4; i += stride) {minmax_acc4(v4[i], lo, hi);}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {float const t = vals[n4*4 + threadIdx.x]; lo = fminf(lo, t); hi = fmaxf(hi, t);}
	any = (lo <= hi); // false only when every value this thread saw was a NaN (or it saw none)
	uint32_t mlo = any ? f2ord(lo) : 0xFFFFFFFFu, mhi = any ? ~f2ord(hi) : 0xFFFFFFFFu;
	wave_minmax_publish(mlo, mhi, d);
}

// ------------------------------------------------------------------ f3: landscape weights texture (tile_t::create_texture, src/tiled_mesh.cpp:1071-1240) + grass blocks
// One block per 32-row band of a tile (the fourth band takes row 128 as well): the band's 33 / 34 height rows are staged in LDS once -- a texel reads its cell's four corners
// from there -- the texels are walked in linear order (consecutive lanes = consecutive texels: the noise field is read and the RGBA words are written in full lines), and the
// band's 8 x 32 grass blocks (add_grass_block_at, src/tiled_mesh.cpp:1354-1371) are folded from the flags the texels left in LDS: no flag array in HBM, no second launch.
constexpr uint32_t WK_BAND = 32, WK_BANDS = 4;
__global__ __launch_bounds__(256) void k_tile_weights(landscape_consts_t c, tile_ref_pod_t const *__restrict__ refs, float const *__restrict__ zvals, float const *__restrict__ noise,
	float const *__restrict__ params, uint32_t *__restrict__ w32, grass_block_pod_t *__restrict__ blocks, uint8_t *__restrict__ any_grass)
{
	__shared__ float s_z[(WK_BAND + 2)*WT_ZV];
	__shared__ uint8_t s_flag[(WK_BAND + 1)*WT_TEX + 3];
	uint32_t const t = blockIdx.x/WK_BANDS, band = blockIdx.x % WK_BANDS, y0 = band*WK_BAND, rows = (band == WK_BANDS - 1) ? WK_BAND + 1 : WK_BAND;
	float const *zt = zvals + (size_t)t*WT_ZV*WT_ZV + (size_t)y0*WT_ZV;
	for (uint32_t i = threadIdx.x; i < (rows + 1)*WT_ZV; i += 256) {s_z[i] = zt[i];}
	biome_corners_t bio;
#pragma unroll
	for (int k = 0; k < 12; ++k) {bio.v[k] = params[(size_t)t*12 + k];} // (block-uniform: scalar loads)
	__syncthreads();
	size_t const tex0 = (size_t)t*WT_TEX*WT_TEX + (size_t)y0*WT_TEX;
	bool grass_here = false;
	for (uint32_t p = threadIdx.x; p < rows*WT_TEX; p += 256) {
		uint32_t const yy = p/WT_TEX, x = p - yy*WT_TEX;
		float const *zc = s_z + yy*WT_ZV + x;
		unsigned flags;
		w32[tex0 + p] = weights_texel_v(c, corner_heights_t{zc[0], zc[1], zc[WT_ZV], zc[WT_ZV + 1]}, bio, noise[tex0 + p], x, y0 + yy, flags);
		s_flag[p] = (uint8_t)flags;
		grass_here |= (flags & 1u) != 0;
	}
	if (grass_here) {any_grass[t] = 1;} // has_any_grass: every writer stores the same value
	if (!blocks) return;
	__syncthreads();
	// the band's grass blocks: 8 rows of 32 blocks of 4 x 4 texels = one per thread, texels in the reference's row-major order (the first contributor picks the block's ix)
	uint32_t const bx = threadIdx.x & 31u, byl = threadIdx.x >> 5;
	tile_ref_pod_t const r = refs[t];
	grass_block_pod_t gb = {0u, 0.0f, 0.0f};
	for (uint32_t yy = byl*GRASS_BLOCK_SZ; yy < (byl + 1)*GRASS_BLOCK_SZ; ++yy) {
		for (uint32_t x = bx*GRASS_BLOCK_SZ; x < (bx + 1)*GRASS_BLOCK_SZ; ++x) {
			if (!(s_flag[yy*WT_TEX + x] & 2u)) continue;
			float const *zc = s_z + yy*WT_ZV + x;
			corner_heights_t const h{zc[0], zc[1], zc[WT_ZV], zc[WT_ZV + 1]};
			float const lowest = min4_std(h), highest = max4_std(h);
			if (gb.ix == 0) {
				gb.ix = ((((uint32_t)(r.tx*128) + x) + 1567u*((uint32_t)(r.ty*128) + y0 + yy)) % c.num_rnd_grass_blocks) + 1; // int + unsigned: unsigned arithmetic
				gb.zmin = lowest; gb.zmax = highest;
			}
			else {gb.zmin = min_std(gb.zmin, lowest); gb.zmax = max_std(gb.zmax, highest);}
		}
	}
	blocks[(size_t)t*GRASS_BLOCK_DIM*GRASS_BLOCK_DIM + (size_t)(band*(WK_BAND/GRASS_BLOCK_SZ) + byl)*GRASS_BLOCK_DIM + bx] = gb;
}

// ------------------------------------------------------------------ K10: 16-bit quantise (heightmap_t::from_floats + write_pixel_16_bits, src/heightmap.cpp:205-215, src/Textures.cpp:1889-1893)
// HBM-bound, 4 B read + 2 B written per cell: eight cells per thread = two 16-byte loads and one 16-byte store of {fraction, integer} byte pairs
__device__ __forceinline__ uint32_t q16_pair(float z, float val_add, float val_div) {
	float const v = (z - val_add)*val_div;
	uint8_t const hi = (uint8_t)v;
	uint8_t const lo = (uint8_t)(256.0f*(v - (float)hi));
	return (uint32_t)lo | ((uint32_t)hi << 8);
}
__global__ __launch_bounds__(256) void k_quantize16(float const *__restrict__ vals, size_t n8, float val_add, float val_div, st_u4 *__restrict__ pix) {
	size_t const i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= n8) return;
	st_f4 const a = __builtin_nontemporal_load((st_f4 const *)vals + 2*i), b = __builtin_nontemporal_load((st_f4 const *)vals + 2*i + 1);
	st_u4 o;
	o.x = q16_pair(a.x, val_add, val_div) | (q16_pair(a.y, val_add, val_div) << 16);
	o.y = q16_pair(a.z, val_add, val_div) | (q16_pair(a.w, val_add, val_div) << 16);
	o.z = q16_pair(b.x, val_add, val_div) | (q16_pair(b.y, val_add, val_div) << 16);
	o.w = q16_pair(b.z, val_add, val_div) | (q16_pair(b.w, val_add, val_div) << 16);
	__builtin_nontemporal_store(o, &pix[i]);
}

// ------------------------------------------------------------------ K8: voxel sine field (noise_gen_3d::get_val, src/upsurface.cpp:60-70)
// val[y][x][z] = sum_k (xv[x][k]*yv[y][k])*zv[z][k], z fastest.  fp32-VALU bound (60 mul + 60 add per voxel for 4 B written).
// One lane = one z: its 60 zv values stay in registers (30 register pairs); the block walks VX_PER_BLOCK (x,y) columns TWO AT A TIME: the products
// P[k] = xv*yv of a column pair are stored interleaved, so one scalar load delivers {P_a[k], P_b[k]} as an SGPR pair and one v_pk_mul_f32 / v_pk_add_f32
// (zv broadcast from either half of its pair by op_sel) advances both columns -- same multiplies and adds, half the instructions.
// (Measured with tools/pk_rate.hip: gfx950 issues a plain fp32 VALU op in ~1.0 ns per SIMD and a packed one in ~1.8 ns, so packing buys ~10%, not 2x.)
// Output rows are contiguous along z: coalesced stores.
constexpr int VX_PER_BLOCK = 32;
constexpr int VX_PSTRIDE = 128; // floats per column pair in P: 60 interleaved pairs, padded to whole 16-float scalar loads
typedef float vx_v16f __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_voxel_P(float *__restrict__ P, float *__restrict__ ZT, uint32_t nx, uint32_t ny, uint32_t nz, float const *__restrict__ tab) {
	// P[(col/2)*128 + k*2 + (col & 1)], columns padded to an even count with zeros; ZT[k][z] = the z table transposed, so that lane z of the main kernel reads it coalesced
	// a thread makes BOTH products of a column pair for one k and writes them as one 8-byte word: consecutive threads write consecutive words (one product per thread wrote
	// every other float of a line: 34 us for the 63 MB of a 512 x 512 field, half of what the 512 x 512 x 64 field's own kernel takes)
	size_t i = (size_t)blockIdx.x*256 + threadIdx.x; size_t const ncol = (size_t)nx*ny, ncol2 = (ncol + 1) & ~(size_t)1;
	if (i < (ncol2/2)*VOX_SINES) {
		uint32_t const k = (uint32_t)(i % VOX_SINES); size_t const pr = i / VOX_SINES;
		sg_v2f v = {0.0f, 0.0f};
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			size_t const c = 2*pr + (size_t)h;
			if (c < ncol) {uint32_t const x = (uint32_t)(c % nx), y = (uint32_t)(c / nx); v[h] = __fmul_rn(tab[(size_t)x*VOX_SINES + k], tab[((size_t)nx + y)*VOX_SINES + k]);}
		}
		*(sg_v2f *)&P[pr*VX_PSTRIDE + k*2] = v;
		return;
	}
	if (i < ncol2*VOX_SINES) return; // (the launch is sized for one thread per product: the upper half of those threads has nothing to do)
	i -= ncol2*VOX_SINES;
	if (i >= (size_t)nz*VOX_SINES) return;
	uint32_t const z = (uint32_t)(i % nz), k = (uint32_t)(i / nz);
	ZT[i] = tab[((size_t)nx + ny + z)*VOX_SINES + k];
}
template<int E> __device__ __forceinline__ void vx_mul_add(sg_v2f &val, sg_v2f p2, sg_v2f zpair) { // val += {P_a*z, P_b*z}: product rounded, then added
	sg_v2f t;
	if (E == 0) {asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(val), "=&v"(t) : "s"(p2), "v"(zpair));}
	else        {asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(val), "=&v"(t) : "s"(p2), "v"(zpair));}
}
__global__ __launch_bounds__(256) void k_voxel_sines(float *__restrict__ out, uint32_t nx, uint32_t ny, uint32_t nz, float const *__restrict__ ZT, float const *__restrict__ P, float zscale, int normalize) {
	uint32_t const z = blockIdx.y*blockDim.x + threadIdx.x;
	bool const active = z < nz;
	sg_v2f zv[VOX_SINES/2];
	float const *zcol = ZT + (active ? z : 0);
#pragma unroll
	for (unsigned k = 0; k < VOX_SINES/2; ++k) {zv[k] = sg_v2f{zcol[(size_t)(2*k)*nz], zcol[(size_t)(2*k + 1)*nz]};}
	float const zterm = __fmul_rn((float)z, zscale);
	size_t const c0 = (size_t)blockIdx.x*VX_PER_BLOCK, ncol = (size_t)nx*ny;
	if (c0 >= ncol) return;
	// The P stream is software-pipelined by hand.  Scalar loads return out of order, so the only usable wait is lgkmcnt(0) -- left to the compiler, every second 64-byte load was
	// issued directly in front of such a wait (its whole latency exposed: SQ_WAIT_ANY 0.65 of the wave cycles).  Here a stage is TWO 64-byte loads (16 k of a column pair), in
	// flight while the 32 packed instructions of the previous stage run: wait, issue the next stage into the other buffers, compute.  The buffers are asm operands from issue
	// to use, nothing else touches them.  P itself streams from HBM (63 MB at 512^3, no reuse): each wave first touches its share of the block's 8 KB slice with ONE vector load
	// (a lane per 64-byte line), so that the scalar loads find the lines in the L2 instead of paying the HBM latency chunk by chunk.
	size_t const last_pair = (ncol - 1) >> 1;
	float const *pp = P + (c0 >> 1)*VX_PSTRIDE; // chunk 0 of this block's first column pair
	float touched, touched2;
	{
		// 128 lines of 64 B = 16 column pairs x 512 B.  A block of four waves (nz >= 256) touches 32 lines per wave; a block of ONE wave (the reference's own 64-deep field)
		// must touch them all itself -- with lines 32 .. 127 left to the scalar loads the 512 x 512 x 64 field took 83 us instead of ~45
		// (two loads per lane at most; both destination registers stay allocated until the waits at the end of the kernel: the loads are asynchronous to the compiler)
		unsigned const nw = blockDim.x >> 6, per_wave = 128u/nw, base = (threadIdx.x >> 6)*per_wave, l0 = threadIdx.x & 63u;
		unsigned const la = base + ((l0 < per_wave) ? l0 : 0u), lb = base + ((l0 + 64u < per_wave) ? l0 + 64u : ((l0 < per_wave) ? l0 : 0u));
		size_t const pa = (c0 >> 1) + (la >> 3), pb = (c0 >> 1) + (lb >> 3);
		float const *ta = P + ((pa <= last_pair) ? pa : last_pair)*VX_PSTRIDE + (la & 7u)*16u, *tb = P + ((pb <= last_pair) ? pb : last_pair)*VX_PSTRIDE + (lb & 7u)*16u;
		asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(touched), "=&v"(touched2) : "v"(ta), "v"(tb));
	}
	vx_v16f b0, b1, b2, b3;
	asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(b0), "=&s"(b1) : "s"(pp));
#define VX_NEXT(CA, CB, NA, NB, ADDR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %2, %4, 0x0\n\ts_load_dwordx16 %3, %4, 0x40" : "+s"(CA), "+s"(CB), "=&s"(NA), "=&s"(NB) : "s"(ADDR))
#define VX_STEP(BUF, Q, J, LO, HI) if (Q*8 + J < VOX_SINES) {vx_mul_add<(J & 1)>(val, sg_v2f{BUF.LO, BUF.HI}, zv[(Q*8 + J < VOX_SINES) ? ((Q*8 + J) >> 1) : 0]);}
#define VX_CHUNK(BUF, Q) VX_STEP(BUF, Q, 0, s0, s1) VX_STEP(BUF, Q, 1, s2, s3) VX_STEP(BUF, Q, 2, s4, s5) VX_STEP(BUF, Q, 3, s6, s7) VX_STEP(BUF, Q, 4, s8, s9) VX_STEP(BUF, Q, 5, sa, sb) VX_STEP(BUF, Q, 6, sc, sd) VX_STEP(BUF, Q, 7, se, sf)
	for (int c = 0; c < VX_PER_BLOCK; c += 2) {
		size_t const col = c0 + c; // = x + y*nx of the pair's first column (even), wave-uniform
		if (col >= ncol) break;
		size_t const np = ((col >> 1) + 1 <= last_pair) ? (col >> 1) + 1 : last_pair; // the pair whose first stage is fetched under this pair's last one (past the end: any valid address)
		float const *pn = P + np*VX_PSTRIDE;
		sg_v2f val = {0.0f, 0.0f}; // (xv*yv)*zv, summed in k order; 8 pairs per 64-byte scalar load
		VX_NEXT(b0, b1, b2, b3, pp + 32); VX_CHUNK(b0, 0) VX_CHUNK(b1, 1)
		VX_NEXT(b2, b3, b0, b1, pp + 64); VX_CHUNK(b2, 2) VX_CHUNK(b3, 3)
		VX_NEXT(b0, b1, b2, b3, pp + 96); VX_CHUNK(b0, 4) VX_CHUNK(b1, 5)
		VX_NEXT(b2, b3, b0, b1, pn);      VX_CHUNK(b2, 6) VX_CHUNK(b3, 7)
		pp = pn;
		float va = __fadd_rn(val.x, zterm), vb = __fadd_rn(val.y, zterm);
		if (normalize) {va = clip_pm1(va); vb = clip_pm1(vb);}
		if (active) {
			__builtin_nontemporal_store(va, &out[col*nz + z]); // written once, never read back by this kernel: keep it out of the L2's way
			if (col + 1 < ncol) {__builtin_nontemporal_store(vb, &out[(col + 1)*nz + z]);}
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(touched), "+v"(touched2)); // the last prefetch and the touch loads land before the wave ends
#undef VX_CHUNK
#undef VX_STEP
#undef VX_NEXT
}

// ---- the same field with a LANE PER COLUMN (round 6).  k_voxel_sines streams a 63 MB array of products P = xv*yv (for a 512 x 512 field) through the scalar unit: as much
// traffic as a 64-deep field's own output, a launch to make it, and per wave a chain of 64 waits on scalar loads -- the reference's own 512 x 512 x 64 field
// (scene_config/config_voxel_params.txt:1-3) ran at a third of the 512^3 field's rate per voxel.  Here a lane owns a column: its 60 products live in 30 register pairs (made from
// the x / y tables once per block and 64 z: no P array, no launch for it) and the z table -- uniform over the wave -- arrives as SCALAR operands (s_load_dwordx8 of the
// transposed table ZT[k][z]: 15 KB per block, shared by its four waves and hot in the scalar cache), eight z per pass = four independent packed accumulator chains.  (The z
// slice in LDS, read as 16-byte broadcasts, was LDS-bound: a broadcast still moves 1 KB per wave and read.)  The results leave through a 64 x 16 transposition piece in LDS so
// that the stores are 16-byte words of whole 64-byte segments.  Same arithmetic, same order as k_voxel_sines: product rounded, then added, k ascending.
// (Pipelining the scalar loads by hand -- stages of two loads, double-buffered, the next stage issued behind the wait for this one -- was measured EQUAL, 47.4 vs 45.7 us for
// 512 x 512 x 64, with the scalar registers at their limit: the compiler's schedule stays.  What is left against the 29 us of packed instructions at full issue is not the waits.)
constexpr unsigned VC_COLS = 256, VC_Z = 64, VC_PIECE = 16, VC_TSTRIDE = 20; // columns per block (a lane each), z per block, z per transposition piece, floats per column in it
typedef float vc_v8f __attribute__((ext_vector_type(8)));
// the table is read through the CONSTANT address space: nothing writes it while the kernel runs, and a uniform load from there is a scalar load whatever stores lie around it
// (as a plain global pointer the compiler kept the loads of the first pass scalar and made vector loads + readfirstlane of the rest, behind the kernel's own stores)
typedef float vc_v16f __attribute__((ext_vector_type(16)));
typedef vc_v16f const __attribute__((address_space(4))) *vc_zt_ptr;
template<int ODD> __device__ __forceinline__ void vc_mul_add4(sg_v2f (&acc)[4], sg_v2f pp, vc_v8f zz) {
	sg_v2f t0, t1, t2, t3;
	sg_v2f const z0 = {zz.s0, zz.s1}, z1 = {zz.s2, zz.s3}, z2 = {zz.s4, zz.s5}, z3 = {zz.s6, zz.s7};
	if (ODD == 0) {
		asm("v_pk_mul_f32 %4, %8, %9 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %5, %8, %10 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %6, %8, %11 op_sel:[0,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %7, %8, %12 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
		    "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
		    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(pp), "s"(z0), "s"(z1), "s"(z2), "s"(z3));
	}
	else {
		asm("v_pk_mul_f32 %4, %8, %9 op_sel:[1,0] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %5, %8, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %6, %8, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %7, %8, %12 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
		    "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
		    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(pp), "s"(z0), "s"(z1), "s"(z2), "s"(z3));
	}
}
// ZT[z/8][k][z%8]: the z table in passes of eight z (written by the table launch of terra_engine::voxel_fill_dev; nzp/8 passes: whole 64-z slices) -- the 60 rows of a pass are
// 1920 contiguous bytes, two rows per 64-byte scalar load; what lies behind nz is never stored
__global__ __launch_bounds__(256, 4) void k_voxel_sines_cols(float *__restrict__ out, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t nzp, float const *__restrict__ tab, float const *__restrict__ ZT, float zscale, int normalize) {
	__shared__ __attribute__((aligned(16))) float s_tp[4][64*VC_TSTRIDE];         // per wave: 64 columns x 16 z, 20 floats per column (16-byte aligned rows)
	size_t const ncol = (size_t)nx*ny, c = (size_t)blockIdx.x*VC_COLS + threadIdx.x;
	uint32_t const z0 = blockIdx.y*VC_Z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	sg_v2f pp[VOX_SINES/2];
	{
		bool const live = c < ncol;
		uint32_t const x = live ? (uint32_t)(c % nx) : 0u, y = live ? (uint32_t)(c / nx) : 0u;
		float const *xt = tab + (size_t)x*VOX_SINES, *yt = tab + ((size_t)nx + y)*VOX_SINES;
#pragma unroll
		for (unsigned j = 0; j < VOX_SINES/2; ++j) {pp[j] = sg_v2f{__fmul_rn(xt[2*j], yt[2*j]), __fmul_rn(xt[2*j + 1], yt[2*j + 1])};}
	}
	float *tp = s_tp[wave];
	size_t const wc0 = (size_t)blockIdx.x*VC_COLS + (size_t)wave*64; // the wave's first column
#pragma unroll 1
	for (unsigned piece = 0; piece < VC_Z/VC_PIECE; ++piece) {
		if (z0 + piece*VC_PIECE >= nz) break;
#pragma unroll
		for (unsigned half = 0; half < 2; ++half) { // eight z per pass
			unsigned const zl = piece*VC_PIECE + half*8;
			float const *zrow = ZT + (size_t)((z0 + zl) >> 3)*(VOX_SINES*8); // (wave-uniform: scalar loads)
			sg_v2f acc[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
			for (unsigned j = 0; j < VOX_SINES/2; ++j) {
				vc_v16f const zz = *(vc_zt_ptr)(uintptr_t)(zrow + 16*j); // the eight z of this pass for k = 2j and k = 2j + 1: one 64-byte scalar load
				vc_mul_add4<0>(acc, pp[j], vc_v8f{zz.s0, zz.s1, zz.s2, zz.s3, zz.s4, zz.s5, zz.s6, zz.s7});
				vc_mul_add4<1>(acc, pp[j], vc_v8f{zz.s8, zz.s9, zz.sa, zz.sb, zz.sc, zz.sd, zz.se, zz.sf});
			}
			float v[8] = {acc[0].x, acc[0].y, acc[1].x, acc[1].y, acc[2].x, acc[2].y, acc[3].x, acc[3].y};
#pragma unroll
			for (unsigned q = 0; q < 8; ++q) {
				float r = __fadd_rn(v[q], __fmul_rn((float)(z0 + zl + q), zscale));
				if (normalize) {r = clip_pm1(r);}
				v[q] = r;
			}
			*(st_f4 *)&tp[lane*VC_TSTRIDE + half*8]     = st_f4{v[0], v[1], v[2], v[3]};
			*(st_f4 *)&tp[lane*VC_TSTRIDE + half*8 + 4] = st_f4{v[4], v[5], v[6], v[7]};
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
		// the piece leaves transposed: a lane stores four consecutive z of one column (16 bytes), four lanes a column's 64-byte segment
		uint32_t const zp = z0 + piece*VC_PIECE;
#pragma unroll
		for (unsigned q = 0; q < 4; ++q) {
			unsigned const idx = q*64 + lane, col = idx >> 2, zq = (idx & 3u)*4u;
			st_f4 const w = *(st_f4 const *)&tp[col*VC_TSTRIDE + zq];
			if (wc0 + col < ncol && zp + zq < nz) {__builtin_nontemporal_store(w, (st_f4 *)&out[(wc0 + col)*nz + zp + zq]);}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
	}
}
} // namespace terra
