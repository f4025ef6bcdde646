// terra_fused.hpp -- the TOLERANCE mode of the generator kernels (TERRA_GEN_FUSED): one rounding per term instead of two.
//
// The reference's sums (mesh_xy_grid_cache_t::eval_index, src/mesh_gen.cpp:766-781; noise_gen_3d::get_val, src/upsurface.cpp:60-70) are compiled without FMA, so the
// bit-exact kernels of terra_kernels.hpp pay a multiply AND an add per term and can never pass half of the chip's fp32 rate.  The reference's own GPU off-load of the same
// function (src/mesh_gen.cpp:666-673, shaders/simplex_noise.part) is not bit-equal to its CPU path either, and BASELINE's bar for the z values is 1e-5 relative: a caller
// that does not feed the heights to the erosion (or accepts other droplet paths) can ask for the fused form.
//
//   k_sine_grid_mx   z[y][x] = sum_k Y[k][y]*X[k][x] on the f32 matrix pipe: v_mfma_f32_32x32x2_f32 accumulates D = A(32 x 2)*B(2 x 32) + C with fp32 inputs in k order,
//                    each term one fused multiply-add -- the value is EXACTLY fmaf(x_k*y_k, acc) chained over k (checked bit for bit against that restatement
//                    in the tests' checker, orc_set_fused), which is within (terms)*2^-24 * sum|x_k y_k| of the reference's mul-then-add chain.  The tables are the
//                    exact kernels' tables (SINF indices stay pinned).  The matrix pipe runs at the vector FMA rate (64 flop / clk / SIMD) but takes its operands as one
//                    register per 32 x 2 slice: 4 operand loads per 8192 multiply-adds, so no LDS staging, no barrier, and the vector ALU is free for the epilogue.
//
// Device only; included by terra_hip.hip (and by tools/sine_mx_probe.hip, which is why this header depends on nothing but the HIP runtime).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace terra {

typedef float sgf_v16 __attribute__((ext_vector_type(16)));

struct sgf_job_t {
	float const *xt, *yt;    // sine tables, k-major, rows padded to nxp / nyp (multiples of 128, zero padded)
	float const *smx, *smy;  // sine-mag island tables (zero padded like the grid), or null
	float *out;
	uint32_t *mm;            // order-preserving {min, ~max} words, or null
	uint32_t nx, ny, nxp, nyp, ntx, nty, rowgroup;
	int32_t kstart, kend;    // terms kstart .. kend - 1 (rows of xt / yt)
	int32_t glaciate, sine_mag;
	float zmax_est, zmax_est2, zmax_est2_inv, sine_offset;
	int32_t const *tile_map; uint32_t nux, tw; // SGF_TILES: scatter into the per-tile layout [tile][tw][tw]
	float zscale; int32_t normalize;           // SGF_VOXELS: the tail of voxel_manager::create_procedural (src/voxels.cpp:340-343)
	void const *xh, *yh; uint32_t nchunks; float unscale; // k_sine_grid_h3: the split half-precision tables [chunk][row][16] and 1 / (their scale factors)
	uint32_t narrow;         // k_sine_grid_h3: the grid is at most 64 columns wide (a voxel field with nz <= 64): a block's four waves take 64 x 256 cells instead of 128 x 128 (nty counts 256-row tiles)
};
enum {SGF_GRID = 0, SGF_TILES = 1, SGF_VOXELS = 2};

#ifndef SGF_STAGES_N
#define SGF_STAGES_N 4
#endif
constexpr int SGF_STAGES = SGF_STAGES_N; // operand register sets in flight (k pairs): 4 or 8

__device__ __forceinline__ uint32_t sgf_f2ord(float f) {uint32_t const u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u);}
__device__ __forceinline__ bool sgf_tile_of_block(unsigned b, unsigned ntx, unsigned nty, unsigned rowgroup, unsigned &bx, unsigned &by) { // (as sg_tile_of_block, terra_kernels.hpp)
	unsigned const nb = ntx*nty, per_xcd = (nb + 7)/8;
	unsigned const lin = (b & 7)*per_xcd + (b >> 3);
	if (lin >= nb) return false;
	unsigned const group = lin/(rowgroup*ntx), r = lin % (rowgroup*ntx);
	unsigned const rows_here = (nty - group*rowgroup < rowgroup) ? nty - group*rowgroup : rowgroup;
	bx = r/rows_here; by = group*rowgroup + r % rows_here;
	return bx < ntx;
}

// 128 x 128 cells per 256-thread block and step, 64 x 64 per wave = 2 x 2 matrix tiles (64 accumulator registers).  Lane l of a wave supplies, for a k pair (k, k+1):
//   A (rows = y): Y[k + (l >> 5)][y0 + 32 i + (l & 31)]     B (columns = x): X[k + (l >> 5)][x0 + 32 j + (l & 31)]
// i.e. every operand load is two 128-byte runs of a table row, straight from the L2 (the tables are 2 x 5.2 MB at 16384^2; an XCD works on one band of tile rows).
// Accumulator register v of tile (i, j) in lane l is the cell (row 32 i + 8 (v >> 2) + 4 (l >> 5) + (v & 3), column 32 j + (l & 31)): a store instruction writes two
// 128-byte runs of two output rows.
// An odd number of terms gets a zero pair member IN FRONT (fma(0, 0, +0) = +0 exactly; behind the last term it would turn a -0 sum into +0).
//
// The blocks are PERSISTENT (block b takes tiles b, b + gridDim.x, ...; gridDim.x is a multiple of 8, so a block stays on its XCD's band) and a wave's stores are DEFERRED:
// the 64 finished values of a tile wait in registers and leave eight at a time between the matrix instructions of the NEXT tile's sum.  Measured on MI355X (16384^2, 80
// terms): one tile per block with the stores at the end 0.62 ms -- every wave of the chip starts together and takes equally long, so all of them reach their stores at the
// same time and the matrix pipe idles while 64 MB drain, 16 times per launch (without the stores: 0.49 ms; without stores and operand loads: 0.41 ms).
constexpr int SGF_STORE_GROUPS = 8; // 64 pending values leave in 8 groups of 8
//
// No vector-ALU instruction in the main loop: the f32 matrix instructions run on the vector ALU's own multipliers (profiles/r04_sine_matrix_pipe.txt), so every v_ instruction
// between them -- the 64-bit address arithmetic of a global_load / global_store was eight of them per four k pairs -- takes its time FROM the matrix pipe, with a bubble on
// either side (measured: the loop with plain pointer loads 0.57 ms, with neither loads nor stores 0.43 ms at the same 0.31 ms of matrix time).  Loads and stores are buffer
// instructions: a per-tile resource descriptor (scalar registers), a per-lane byte offset that never changes, and the k pair / output row as the scalar offset.
typedef float sgf_v2 __attribute__((ext_vector_type(2)));

struct sgf_pending_t {float v[64];}; // v[8 g + 2 r + j]: row 32 (g >> 2) + 8 (g & 3) + r (+ 4 for the upper half-wave), column half j

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sgf_rsrc(void const *p) {return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0xFFFFFFFFu, 0x00020000);}
__device__ __forceinline__ float sgf_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));}
__device__ __forceinline__ float sgf_min3(float a, float b, float c) {float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;}
__device__ __forceinline__ float sgf_max3(float a, float b, float c) {float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;}

// store group g of the pending tile (g is a compile-time constant where this is called from unrolled code).  vo[2 r + j]: the lane's byte offset of row r, column half j;
// the group's first row is the scalar offset -- one scalar register per group instead of one per row (32 of them, hoisted out of the loop and spilled, it was)
__device__ __forceinline__ void sgf_store_group(sgf_pending_t const &P, __amdgpu_buffer_rsrc_t ro, uint32_t const (&vo)[8], uint32_t row_bytes, int g) {
#define TERRA_SGF_CASE(G) case G: { \
		uint32_t const so = (uint32_t)(32*((G) >> 2) + 8*((G) & 3))*row_bytes; \
		_Pragma("unroll") for (int e = 0; e < 8; ++e) {__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(P.v[8*(G) + e]), ro, vo[e], so, 2);} \
	} break;
	switch (g) {TERRA_SGF_CASE(0) TERRA_SGF_CASE(1) TERRA_SGF_CASE(2) TERRA_SGF_CASE(3) TERRA_SGF_CASE(4) TERRA_SGF_CASE(5) TERRA_SGF_CASE(6) TERRA_SGF_CASE(7) default: break;}
#undef TERRA_SGF_CASE
}

// one k pair on the wave's 2 x 2 matrix tiles (C0 = acc, or a zero vector for a tile's first pair: the accumulators then need no clearing -- 64 vector moves per tile)
#define TERRA_SGF_MFMA4_(S, C00, C01, C10, C11) \
	acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S], b0[S], C00, 0, 0, 0); \
	acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S], b1[S], C01, 0, 0, 0); \
	acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S], b0[S], C10, 0, 0, 0); \
	acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S], b1[S], C11, 0, 0, 0);
#define TERRA_SGF_MFMA4(S) TERRA_SGF_MFMA4_(S, acc[0][0], acc[0][1], acc[1][0], acc[1][1])
#ifndef SGF_PROBE_NOLOAD
#define TERRA_SGF_LOAD4(S, PAIR) {uint32_t const q_ = (uint32_t)(PAIR); a0[S] = sgf_ld(ra, va0, q_*sa); a1[S] = sgf_ld(ra, va1, q_*sa); b0[S] = sgf_ld(rb, vb0, q_*sb); b1[S] = sgf_ld(rb, vb1, q_*sb);}
#else
#define TERRA_SGF_LOAD4(S, PAIR) {(void)(PAIR);}
#endif
// four k pairs; the reload of a register set stays right behind its four matrix instructions: SGF_STAGES - 1 stages (768 matrix-pipe cycles) until it is used
#define TERRA_SGF_ITERATION(FIRST) \
	_Pragma("unroll") for (int s = 0; s < SGF_STAGES; ++s) { \
		if ((FIRST) && s == 0) {TERRA_SGF_MFMA4_(0, zero16, zero16, zero16, zero16)} else {TERRA_SGF_MFMA4(s)} \
		TERRA_SGF_LOAD4(s, (p + s + SGF_STAGES < last) ? p + s + SGF_STAGES : last) \
		__builtin_amdgcn_sched_barrier(0); \
	}

// KIND: SGF_GRID (a heightmap, row-major), SGF_TILES (a tile batch's virtual grid, scattered), SGF_VOXELS (noise_gen_3d's field: "rows" are the (x, y) columns with the
// products xv*yv as their table, "columns" the z cells -- src/upsurface.cpp:60-70 is the same rank-k contraction with a different tail)
template<int KIND> __global__ __launch_bounds__(256, 2) void k_sine_grid_mx(sgf_job_t const J) {
	constexpr bool TILES = (KIND == SGF_TILES), VOX = (KIND == SGF_VOXELS);
	unsigned const lane = threadIdx.x & 63u, half = lane >> 5, c = lane & 31u;
	unsigned const w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform, and the compiler knows it
	// (no term at all, min_start_sin >= 90: the operand loads still run -- of the table's last pair, never multiplied)
	int const nk = (J.kstart < J.kend) ? J.kend - J.kstart : 0, kbase = (nk > 0) ? J.kstart - (nk & 1) : J.kend - 2, npairs = (nk + 1) >> 1, last = (npairs > 0) ? npairs - 1 : 0;
	uint32_t const sa = 8u*J.nyp, sb = 8u*J.nxp; // bytes per k pair
	uint32_t const va0 = (half*J.nyp + c)*4u, va1 = va0 + 128u, vb0 = (half*J.nxp + c)*4u, vb1 = vb0 + 128u; // the lane's byte offsets into a table slice: the same for every tile
	uint32_t const row_bytes = 4u*J.nx;
	uint32_t vo[8];                                                                                          // and into an output tile: rows r = 0 .. 3 of a group, both column halves
#pragma unroll
	for (int e = 0; e < 8; ++e) {vo[e] = (4u*half + (uint32_t)(e >> 1))*row_bytes + 4u*c + 128u*(uint32_t)(e & 1);}
	bool const gl = J.glaciate != 0, sm = J.sine_mag != 0;
	sgf_v2 const zme = {J.zmax_est, J.zmax_est}, inv = {J.zmax_est2_inv, J.zmax_est2_inv}, z2 = {J.zmax_est2, J.zmax_est2}, off = {J.sine_offset, J.sine_offset};
	// eval_index's tail for the common configuration (src/mesh_gen.cpp:358-386,781-790), contraction allowed; two cells (rows r, r + 1 of a column) per instruction
	sgf_v2 const zsc = {J.zscale, J.zscale};
	bool const norm = J.normalize != 0;
	auto const finish = [&](sgf_v2 z, sgf_v2 sx, sgf_v2 sy) -> sgf_v2 {
		if (VOX) { // val += z*zscale; CLIP_TO_pm1 (std::min / std::max: a NaN becomes 1); sx = (float)z of the column
			z = __builtin_elementwise_fma(sx, zsc, z);
			if (norm) {
#pragma unroll
				for (int e = 0; e < 2; ++e) {float const m = (z[e] < 1.0f) ? z[e] : 1.0f; z[e] = (-1.0f < m) ? m : -1.0f;}
			}
			return z;
		}
		if (gl) {sgf_v2 const rel = (z + zme)*inv; z = __builtin_elementwise_fma((rel*rel)*rel, z2, -zme);}
		if (sm) {z = z + __builtin_elementwise_fma(sx, sy, off);}
		return z;
	};
	float fmn = INFINITY, fmx = -INFINITY;
	sgf_v16 zero16;
#pragma unroll
	for (int v = 0; v < 16; ++v) {zero16[v] = 0.0f;}
	sgf_pending_t P;
	__amdgpu_buffer_rsrc_t ro = sgf_rsrc(J.out);
	bool have = false; // wave-uniform: P holds a finished tile
	unsigned const total = ((J.ntx*J.nty + 7u)/8u)*8u;
	// the tile walk: `t` is the tile whose operands are being loaded
	unsigned t = blockIdx.x, bxi = 0, byi = 0;
	auto const next_tile = [&]() -> bool {for (; t < total; t += gridDim.x) {if (sgf_tile_of_block(t, J.ntx, J.nty, J.rowgroup, bxi, byi)) {t += gridDim.x; return true;}} return false;};
	bool valid = next_tile();
	unsigned x0 = bxi*128u + (w & 1u)*64u, y0 = byi*128u + (w >> 1)*64u;
	__amdgpu_buffer_rsrc_t ra = sgf_rsrc(J.yt + (size_t)kbase*J.nyp + y0), rb = sgf_rsrc(J.xt + (size_t)kbase*J.nxp + x0);
	float a0[SGF_STAGES], a1[SGF_STAGES], b0[SGF_STAGES], b1[SGF_STAGES];
#pragma unroll
	for (int s = 0; s < SGF_STAGES; ++s) {a0[s] = a1[s] = b0[s] = b1[s] = 0.0f;}
	if (valid) {
#pragma unroll
		for (int s = 0; s < SGF_STAGES; ++s) {TERRA_SGF_LOAD4(s, (s < last) ? s : last)} // pairs beyond the last one re-read it (never multiplied)
	}
	while (valid) {
		unsigned const cx0 = x0, cy0 = y0;
		sgf_v16 acc[2][2];
		bool const fresh = have && npairs >= SGF_STAGES; // the unrolled iterations below start the sums themselves
		if (!fresh) {
#pragma unroll
			for (int i = 0; i < 2; ++i) {
#pragma unroll
				for (int j = 0; j < 2; ++j) {acc[i][j] = zero16;}
			}
		}
		if ((nk & 1) && half == 0) {a0[0] = 0.0f; a1[0] = 0.0f; b0[0] = 0.0f; b1[0] = 0.0f;} // the pair member in front of the first term
		int p = 0, g = 0;
		asm volatile("" : "+s"(p)); // (opaque: otherwise the 32 pair offsets of the unrolled iterations below are hoisted out of the tile loop as 64 scalar registers, spilled, and read back by vector instructions between the matrix ones)
		// the stores of the previous tile go FIRST in an iteration: the memory counter is one in-order count of loads and stores, so behind the loads of the iteration they
		// would have to be waited for with them; in front, they have a whole iteration (1024 matrix-pipe cycles) to be acknowledged before a load behind them is needed
		if (have) { // (unrolled: the group number is a compile-time constant, the pending values are named registers)
			constexpr int GPI = SGF_STAGES/4; // store groups per iteration: one per four k pairs
#pragma unroll
			for (int it = 0; it < SGF_STORE_GROUPS/GPI; ++it) {
				if (p + SGF_STAGES <= npairs) {
#pragma unroll
					for (int e = 0; e < GPI; ++e) {sgf_store_group(P, ro, vo, row_bytes, it*GPI + e);}
					__builtin_amdgcn_sched_barrier(0);
					TERRA_SGF_ITERATION(it == 0)
					p += SGF_STAGES; g = (it + 1)*GPI;
				}
			}
		}
		for (; p + SGF_STAGES <= npairs; p += SGF_STAGES) {TERRA_SGF_ITERATION(false)}
#pragma unroll
		for (int s = 0; s < SGF_STAGES - 1; ++s) { // the remaining npairs % SGF_STAGES pairs are in the first register sets
			if (p + s < npairs) {TERRA_SGF_MFMA4(s)}
		}
		if (have && g < SGF_STORE_GROUPS) { // (fewer than 32 k pairs: the rest leaves here)
#pragma unroll
			for (int gg = 0; gg < SGF_STORE_GROUPS; ++gg) {if (gg >= g) {sgf_store_group(P, ro, vo, row_bytes, gg);}}
		}
		have = false;
		// ---- the island terms of this tile, then the first operands of the NEXT tile: both are in flight while the vector ALU finishes this tile's cells
		// (all loads of a tile come before its first store: a load between the stores would wait for every store before it)
		float sxv[2] = {0.0f, 0.0f};
		float4 sy4[2][4];
#pragma unroll
		for (int i = 0; i < 2; ++i) {
#pragma unroll
			for (int q = 0; q < 4; ++q) {sy4[i][q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);}
		}
		if (VOX) {sxv[0] = (float)(cx0 + c); sxv[1] = (float)(cx0 + 32u + c);}
		else if (sm) { // the island terms of the wave's 2 x 32 columns and 64 rows (tables zero padded to the block grid)
			sxv[0] = J.smx[cx0 + c]; sxv[1] = J.smx[cx0 + 32u + c];
#pragma unroll
			for (int i = 0; i < 2; ++i) {
#pragma unroll
				for (int q = 0; q < 4; ++q) {sy4[i][q] = *(float4 const *)(J.smy + cy0 + 32u*i + 8u*q + 4u*half);}
			}
		}
		valid = next_tile();
		if (valid) {
			x0 = bxi*128u + (w & 1u)*64u; y0 = byi*128u + (w >> 1)*64u;
			ra = sgf_rsrc(J.yt + (size_t)kbase*J.nyp + y0); rb = sgf_rsrc(J.xt + (size_t)kbase*J.nxp + x0);
#pragma unroll
			for (int s = 0; s < SGF_STAGES; ++s) {TERRA_SGF_LOAD4(s, (s < last) ? s : last)}
		}
		bool const inside = (cx0 + 64u <= J.nx) && (cy0 + 64u <= J.ny); // wave-uniform: no per-cell bounds tests for a tile inside the grid
		if (!TILES && inside) { // finish into the pending registers; they leave during the next tile
#pragma unroll
			for (int i = 0; i < 2; ++i) {
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					sgf_v2 const sy01 = {sy4[i][q].x, sy4[i][q].y}, sy23 = {sy4[i][q].z, sy4[i][q].w};
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						sgf_v2 const sx = {sxv[j], sxv[j]};
						sgf_v2 const z01 = finish(sgf_v2{acc[i][j][4*q], acc[i][j][4*q + 1]}, sx, sy01), z23 = finish(sgf_v2{acc[i][j][4*q + 2], acc[i][j][4*q + 3]}, sx, sy23);
						fmn = sgf_min3(sgf_min3(fmn, z01.x, z01.y), z23.x, z23.y); fmx = sgf_max3(sgf_max3(fmx, z01.x, z01.y), z23.x, z23.y); // (NaNs are skipped, as min_eq / max_eq never let one win)
						int const b = 8*(4*i + q) + j;
						P.v[b] = z01.x; P.v[b + 2] = z01.y; P.v[b + 4] = z23.x; P.v[b + 6] = z23.y;
					}
				}
			}
			ro = sgf_rsrc(J.out + (size_t)cy0*J.nx + cx0);
			have = true;
#ifdef SGF_PROBE_NOSTORE
			have = (fmn == 123.456f);
#endif
		}
		else { // border tiles and the per-tile layout: stored at once
			unsigned t_ux[2] = {0, 0}, t_cx[2] = {0, 0};
			if (TILES) {
#pragma unroll
				for (int j = 0; j < 2; ++j) {unsigned const x = cx0 + 32u*j + c; t_ux[j] = x/J.tw; t_cx[j] = x - t_ux[j]*J.tw;}
			}
#pragma unroll
			for (int i = 0; i < 2; ++i) {
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					unsigned const yq = cy0 + 32u*i + 8u*q + 4u*half;
					sgf_v2 const sy01 = {sy4[i][q].x, sy4[i][q].y}, sy23 = {sy4[i][q].z, sy4[i][q].w};
					unsigned uy = 0, cy = 0;
					if (TILES) {uy = yq/J.tw; cy = yq - uy*J.tw;}
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						unsigned const x = cx0 + 32u*j + c;
						sgf_v2 const sx = {sxv[j], sxv[j]};
						sgf_v2 const z01 = finish(sgf_v2{acc[i][j][4*q], acc[i][j][4*q + 1]}, sx, sy01), z23 = finish(sgf_v2{acc[i][j][4*q + 2], acc[i][j][4*q + 3]}, sx, sy23);
						float const zr[4] = {z01.x, z01.y, z23.x, z23.y};
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							unsigned const y = yq + r;
							unsigned uyr = uy, cyr = cy + r;
							if (TILES && cyr >= J.tw) {cyr -= J.tw; ++uyr;}
							if (x < J.nx && y < J.ny) {
								fmn = fminf(fmn, zr[r]); fmx = fmaxf(fmx, zr[r]);
								if (TILES) {
									int const tl = J.tile_map[uyr*J.nux + t_ux[j]];
									if (tl >= 0) {J.out[(size_t)tl*J.tw*J.tw + cyr*J.tw + t_cx[j]] = zr[r];}
								}
								else {J.out[(size_t)y*J.nx + x] = zr[r];}
							}
						}
					}
				}
			}
		}
	}
	if (have) {
#pragma unroll
		for (int gg = 0; gg < SGF_STORE_GROUPS; ++gg) {sgf_store_group(P, ro, vo, row_bytes, gg);}
	}
	if (J.mm) {
		uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
		if (fmn <= fmx) {lo = sgf_f2ord(fmn); hi = ~sgf_f2ord(fmx);}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			uint32_t const l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
			lo = (l2 < lo) ? l2 : lo; hi = (h2 < hi) ? h2 : hi;
		}
		if (lane == 0 && lo != 0xFFFFFFFFu) {
			if (lo < __hip_atomic_load(&J.mm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {atomicMin(&J.mm[0], lo);}
			if (hi < __hip_atomic_load(&J.mm[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {atomicMin(&J.mm[1], hi);}
		}
	}
}
#undef TERRA_SGF_ITERATION
#undef TERRA_SGF_LOAD4
#undef TERRA_SGF_MFMA4
#undef TERRA_SGF_MFMA4_


// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------------
// k_sine_grid_h3 -- the same contraction on the HALF-precision matrix pipe (TERRA_GEN_FAST): 16 times the f32 pipe's rate, so the kernel is bound by writing its result.
//
// A table value t (scaled by a power of two so that the table's largest magnitude sits near 2^14) is split into two halves, t = h + l with h = half(t), l = half(t - h): 22
// significant bits, |t - h - l| <= 2^-22 |t| (or 2^-25 absolute in the scaled unit for tiny values: the half format's subnormal step).  A term x*y becomes the three products
// xh*yh + xh*yl + xl*yh (xl*yl <= 2^-22 |x y| is dropped): each is EXACT in fp32 (11 x 11 bits), the matrix instruction adds them into an fp32 accumulator.  So one k of the
// reference's sum is three k of v_mfma_f32_32x32x16_f16 -- [yh yl yh] on the row side against [xh xh xl] on the column side -- and 80 terms are 15 instructions per 32 x 32
// cells instead of 40 of the f32 form at a quarter of their rate each.  The error per term is <= 3 * 2^-22 |x y| plus the accumulator's roundings: the result is within
// 1e-5 * zmax_est of the reference (BASELINE's bar; the tests measure ~1e-6, the same as the fused-multiply-add chain of k_sine_grid_mx) but it is NOT the value of any simple
// restatement -- how the instruction rounds its 16-product sum is the hardware's business -- so this form is tested against the tolerance only.
//
// Tables (k_split_table below): chunk-major [chunk][row][16 halves], a chunk = 16 consecutive k of the tripled sum = 32 bytes per row: lane l of a wave reads the 16 bytes
// (row l & 31, halves 8 (l >> 5) .. + 7) -- a wave's operand load is ONE contiguous kilobyte.  Which k a lane half holds is the same on both sides, so the sum is complete
// whatever the instruction's internal k order.  4 operand loads (4 KB) per 4 instructions (128 matrix cycles) per wave: the L2 supplies that at about half of the matrix
// pipe's rate, which is still twice what the result's stores allow.  Blocks of 4 waves x (64 x 64) cells as in k_sine_grid_mx, one tile per block (the tiles of a CU's
// resident blocks are in different phases by themselves: three to four waves per SIMD fit).
typedef _Float16 sgh_h8 __attribute__((ext_vector_type(8)));
typedef uint32_t sgh_u4 __attribute__((ext_vector_type(4)));

// T: f32 table, k-major [k][np] (zero padded rows); H: [chunk][np][16] halves.  Tripled index q = 3 (k - kstart) + j; ROWSIDE: pieces (h, l, h), else (h, h, l).  One thread
// per (chunk, row): 32 bytes out.
template<bool ROWSIDE> __global__ __launch_bounds__(256) void k_split_table(float const *__restrict__ T, uint32_t np, int32_t kstart, int32_t kend, uint32_t nchunks, float scale, sgh_u4 *__restrict__ H) {
	size_t const i = (size_t)blockIdx.x*256 + threadIdx.x;
	if (i >= (size_t)nchunks*np) return;
	uint32_t const chunk = (uint32_t)(i / np), row = (uint32_t)(i % np);
	_Float16 v[16];
#pragma unroll
	for (int e = 0; e < 16; ++e) {
		int const q = (int)chunk*16 + e, k = kstart + q/3, j = q % 3;
		float t = 0.0f;
		if (k < kend) {t = T[(size_t)k*np + row]*scale;}
		_Float16 const h = (_Float16)t, l = (_Float16)(t - (float)h);
		v[e] = ((ROWSIDE ? (j == 1) : (j == 2)) ? l : h);
	}
	sgh_u4 o[2];
	__builtin_memcpy(o, v, 32);
	H[2*i] = o[0]; H[2*i + 1] = o[1];
}

// the voxel field's tables split straight from noise_gen_3d's [entry][NS] sine table (entries: nx x values, then ny y values, then nz z values; src/upsurface.cpp:41-57),
// no f32 copy in between.  ROWSIDE: a row is an (x, y) column, its value the product xv[x][k]*yv[y][k], rounded as the reference rounds it; else a row is a z cell.
// One thread per ROW: its NS values are read once (neighbouring threads read neighbouring table rows) and leave as ceil(3 NS / 16) 32-byte pieces, one per chunk plane --
// a wave writes 2 KB runs.  (One thread per (chunk, row) re-read the table rows through 240-byte strides: 88 us for the 512^2 columns of a field instead of 30.)
template<bool ROWSIDE, int NS> __global__ __launch_bounds__(256) void k_split_voxel_table(float const *__restrict__ tab, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t np, float scale, sgh_u4 *__restrict__ H) {
	size_t const row = (size_t)blockIdx.x*256 + threadIdx.x;
	if (row >= np) return;
	bool const live = ROWSIDE ? (row < (size_t)nx*ny) : (row < nz);
	float t[NS];
	if (live) {
		float const *pa = ROWSIDE ? tab + (row % nx)*NS : tab + ((size_t)nx + ny + row)*NS, *pb = ROWSIDE ? tab + ((size_t)nx + row / nx)*NS : pa;
#pragma unroll
		for (int k = 0; k < NS; ++k) {t[k] = ROWSIDE ? __fmul_rn(pa[k], pb[k])*scale : pa[k]*scale;}
	}
	else {
#pragma unroll
		for (int k = 0; k < NS; ++k) {t[k] = 0.0f;}
	}
	constexpr int NCHUNKS = (3*NS + 15)/16;
#pragma unroll
	for (int chunk = 0; chunk < NCHUNKS; ++chunk) {
		_Float16 v[16];
#pragma unroll
		for (int e = 0; e < 16; ++e) {
			int const q = chunk*16 + e, k = q/3, j = q % 3;
			float const tk = (k < NS) ? t[(k < NS) ? k : 0] : 0.0f;
			_Float16 const h = (_Float16)tk, l = (_Float16)(tk - (float)h);
			v[e] = ((ROWSIDE ? (j == 1) : (j == 2)) ? l : h);
		}
		sgh_u4 o[2];
		__builtin_memcpy(o, v, 32);
		size_t const i = (size_t)chunk*np + row;
		H[2*i] = o[0]; H[2*i + 1] = o[1];
	}
}

__device__ __forceinline__ sgh_h8 sgh_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {return __builtin_bit_cast(sgh_h8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));}

template<int KIND> __global__ __launch_bounds__(256, 2) void k_sine_grid_h3(sgf_job_t const J) {
	constexpr bool TILES = (KIND == SGF_TILES), VOX = (KIND == SGF_VOXELS);
	unsigned const lane = threadIdx.x & 63u, half = lane >> 5, c = lane & 31u;
	unsigned const w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	unsigned bxi = 0, byi = 0;
	if (!sgf_tile_of_block(blockIdx.x, J.ntx, J.nty, J.rowgroup, bxi, byi)) return; // (block-uniform)
	unsigned const x0 = J.narrow ? 0u : bxi*128u + (w & 1u)*64u, y0 = J.narrow ? byi*256u + w*64u : byi*128u + (w >> 1)*64u;
	__amdgpu_buffer_rsrc_t const ra = sgf_rsrc((char const *)J.yh + (size_t)y0*32u), rb = sgf_rsrc((char const *)J.xh + (size_t)x0*32u);
	uint32_t const v0 = c*32u + half*16u, v1 = v0 + 1024u; // the lane's bytes in a chunk: rows c and 32 + c
	uint32_t const sa = J.nyp*32u, sb = J.nxp*32u;         // bytes per chunk
	int const n = (int)J.nchunks;
	sgf_v16 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; ++i) {
#pragma unroll
		for (int j = 0; j < 2; ++j) {
#pragma unroll
			for (int v = 0; v < 16; ++v) {acc[i][j][v] = 0.0f;}
		}
	}
	sgh_h8 a0[2], a1[2], b0[2], b1[2];
#define TERRA_SGH_LOAD(S, Q) {uint32_t const q_ = (uint32_t)(Q); a0[S] = sgh_ld(ra, v0, q_*sa); a1[S] = sgh_ld(ra, v1, q_*sa); b0[S] = sgh_ld(rb, v0, q_*sb); b1[S] = sgh_ld(rb, v1, q_*sb);}
#define TERRA_SGH_MFMA(S) \
	acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[S], b0[S], acc[0][0], 0, 0, 0); \
	acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[S], b1[S], acc[0][1], 0, 0, 0); \
	acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S], b0[S], acc[1][0], 0, 0, 0); \
	acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S], b1[S], acc[1][1], 0, 0, 0);
	if (n > 0) {
		TERRA_SGH_LOAD(0, 0)
		int q = 0;
		for (; q + 2 <= n; q += 2) { // two register sets: a chunk's operands arrive while the previous chunk is multiplied (reads past the last chunk repeat it, never multiplied)
			TERRA_SGH_LOAD(1, q + 1)
			TERRA_SGH_MFMA(0)
			TERRA_SGH_LOAD(0, (q + 2 < n) ? q + 2 : n - 1)
			TERRA_SGH_MFMA(1)
		}
		if (q < n) {TERRA_SGH_MFMA(0)}
	}
#undef TERRA_SGH_MFMA
#undef TERRA_SGH_LOAD
	// ---- the tail of the cell (as k_sine_grid_mx) and the stores: accumulator register v of tile (i, j) is the cell (row 32 i + 8 (v >> 2) + 4 half + (v & 3), column 32 j + c)
	bool const gl = J.glaciate != 0, sm = J.sine_mag != 0, norm = J.normalize != 0;
	sgf_v2 const us = {J.unscale, J.unscale}, zme = {J.zmax_est, J.zmax_est}, inv = {J.zmax_est2_inv, J.zmax_est2_inv}, z2 = {J.zmax_est2, J.zmax_est2}, off = {J.sine_offset, J.sine_offset}, zsc = {J.zscale, J.zscale};
	// two cells (rows r, r + 1 of a column) per instruction; the tail itself is k_sine_grid_mx's (every a*b + c rounds once)
	auto const finish = [&](sgf_v2 z, sgf_v2 sx, sgf_v2 sy) -> sgf_v2 {
		z = z*us; // (a power of two: exact)
		if (VOX) {
			z = __builtin_elementwise_fma(sx, zsc, z);
			if (norm) {
#pragma unroll
				for (int e = 0; e < 2; ++e) {float const m = (z[e] < 1.0f) ? z[e] : 1.0f; z[e] = (-1.0f < m) ? m : -1.0f;}
			}
			return z;
		}
		if (gl) {sgf_v2 const rel = (z + zme)*inv; z = __builtin_elementwise_fma((rel*rel)*rel, z2, -zme);}
		if (sm) {z = z + __builtin_elementwise_fma(sx, sy, off);}
		return z;
	};
	float sxv[2] = {0.0f, 0.0f};
	if (VOX) {sxv[0] = (float)(x0 + c); sxv[1] = (float)(x0 + 32u + c);}
	else if (sm) {sxv[0] = J.smx[x0 + c]; sxv[1] = J.smx[x0 + 32u + c];}
	float fmn = INFINITY, fmx = -INFINITY;
	bool const inside = (x0 + 64u <= J.nx) && (y0 + 64u <= J.ny); // wave-uniform
	unsigned t_ux[2] = {0, 0}, t_cx[2] = {0, 0};
	if (TILES) {
#pragma unroll
		for (int j = 0; j < 2; ++j) {unsigned const x = x0 + 32u*j + c; t_ux[j] = x/J.tw; t_cx[j] = x - t_ux[j]*J.tw;}
	}
	// a tile inside a row-major grid: buffer stores -- the lane's byte offset never changes, the row is the scalar offset: no vector address arithmetic at all
	uint32_t const row_bytes = 4u*J.nx;
	__amdgpu_buffer_rsrc_t const ro = sgf_rsrc(J.out + (TILES ? (size_t)0 : (size_t)y0*J.nx + x0));
	uint32_t const vo[2] = {4u*half*row_bytes + 4u*c, 4u*half*row_bytes + 4u*c + 128u};
#pragma unroll
	for (int i = 0; i < 2; ++i) {
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			unsigned const yq = y0 + 32u*i + 8u*q + 4u*half;
			float4 sy4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (!VOX && sm) {sy4 = *(float4 const *)(J.smy + yq);}
			sgf_v2 const sy01 = {sy4.x, sy4.y}, sy23 = {sy4.z, sy4.w};
			unsigned uy = 0, cy = 0;
			if (TILES) {uy = yq/J.tw; cy = yq - uy*J.tw;}
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				sgf_v2 const sx = {sxv[j], sxv[j]};
				sgf_v2 const z01 = finish(sgf_v2{acc[i][j][4*q], acc[i][j][4*q + 1]}, sx, sy01), z23 = finish(sgf_v2{acc[i][j][4*q + 2], acc[i][j][4*q + 3]}, sx, sy23);
				float const zr[4] = {z01.x, z01.y, z23.x, z23.y};
				if (!TILES && inside) {
					fmn = sgf_min3(sgf_min3(fmn, zr[0], zr[1]), zr[2], zr[3]); fmx = sgf_max3(sgf_max3(fmx, zr[0], zr[1]), zr[2], zr[3]); // (NaNs are skipped, as min_eq / max_eq never let one win)
					uint32_t const so = (uint32_t)(32*i + 8*q)*row_bytes;
#pragma unroll
					for (int r = 0; r < 4; ++r) {__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(zr[r]), ro, vo[j], so + (uint32_t)r*row_bytes, 2);}
				}
				else { // border tiles and the per-tile layout
					unsigned const x = x0 + 32u*j + c;
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						unsigned const y = yq + (unsigned)r;
						if (x < J.nx && y < J.ny) {
							fmn = fminf(fmn, zr[r]); fmx = fmaxf(fmx, zr[r]);
							if (TILES) {
								unsigned uyr = uy, cyr = cy + (unsigned)r;
								if (cyr >= J.tw) {cyr -= J.tw; ++uyr;}
								int const tl = J.tile_map[uyr*J.nux + t_ux[j]];
								if (tl >= 0) {J.out[(size_t)tl*J.tw*J.tw + cyr*J.tw + t_cx[j]] = zr[r];}
							}
							else {J.out[(size_t)y*J.nx + x] = zr[r];}
						}
					}
				}
			}
		}
	}
	if (J.mm) {
		uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
		if (fmn <= fmx) {lo = sgf_f2ord(fmn); hi = ~sgf_f2ord(fmx);}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			uint32_t const l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
			lo = (l2 < lo) ? l2 : lo; hi = (h2 < hi) ? h2 : hi;
		}
		if (lane == 0 && lo != 0xFFFFFFFFu) {
			if (lo < __hip_atomic_load(&J.mm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {atomicMin(&J.mm[0], lo);}
			if (hi < __hip_atomic_load(&J.mm[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {atomicMin(&J.mm[1], hi);}
		}
	}
}

} // namespace terra
