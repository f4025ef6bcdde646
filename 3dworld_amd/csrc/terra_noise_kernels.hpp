// terra_noise_kernels.hpp -- the per-cell fBm kernels (K2/K3 of terra_kernels.hpp, and the 3-D lattice field of voxel_manager::create_procedural) in a header of their own:
// it is compiled TWICE.  terra_hip.hip builds it like everything else, without floating-point contraction (bit-identical to the reference's CPU path); terra_fz.hip builds
// the same source under another namespace name with contraction allowed -- the TOLERANCE mode of these kernels (TERRA_GEN_FUSED, include/terra.h): the reference's expression
// trees with a*b + c rounded once wherever the compiler finds one.  The lattice hashes are integer-valued fp32 arithmetic below 2^24 (terra_noise.hpp), exact with or without
// contraction, so a fused cell visits the same lattice points and gradients; what moves is the last bits of the interpolation.
#pragma once
#include "terra_driver.hpp"

namespace terra {

// fold a thread's (min,max) of order-preserving uints over its wave and publish with two atomics per wave
__device__ __forceinline__ void wave_minmax_publish(uint32_t lo, uint32_t hi, uint32_t *mm) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		uint32_t const l2 = __shfl_down(lo, off, 64), h2 = __shfl_down(hi, off, 64);
		lo = (l2 < lo) ? l2 : lo; hi = (h2 < hi) ? h2 : hi;
	}
	// thousands of waves fold into the same two words: look first (a stale value only costs a redundant atomic), so the atomics die out once the extrema have been seen
	if ((threadIdx.x & 63) == 0 && lo != 0xFFFFFFFFu) {
		if (lo < __hip_atomic_load(&mm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {atomicMin(&mm[0], lo);}
		if (hi < __hip_atomic_load(&mm[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {atomicMin(&mm[1], hi);}
	}
}
__device__ __forceinline__ float sg_min3(float a, float b, float c) {float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;}
__device__ __forceinline__ float sg_max3(float a, float b, float c) {float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;}
__device__ __forceinline__ void minmax_acc(float v, uint32_t &lo, uint32_t &hi) {if (v == v) {uint32_t const o = f2ord(v); lo = (o < lo) ? o : lo; hi = (~o < hi) ? ~o : hi;}}

// ------------------------------------------------------------------ K2/K3: fBm / domain-warp grid
// fp32-VALU bound lattice noise.  Two cells of a row per lane (packed arithmetic, terra_noise.hpp); everything that depends on the hashed lattice
// point only -- the second permute and the gradient / normalisation terms, ~25 of the ~60 instructions per lattice point -- comes from a table in
// LDS (noise_lut_fill: 10.4 KB for simplex, 5.8 KB for Perlin, staged per block with 16-byte loads), which works for any sample position, the
// domain-warped ones included.
template<int MODE> __device__ __forceinline__ noise_tab_t noise_stage_lut(uint32_t const *__restrict__ lut, uint32_t *s_lut) {
	constexpr bool SIMPLEX = (MODE != MGEN_PERLIN);
	constexpr unsigned first = SIMPLEX ? 0u : NOISE_LUT_S_DWORDS, count = SIMPLEX ? NOISE_LUT_S_DWORDS : NOISE_LUT_P_DWORDS;
	for (unsigned i = threadIdx.x; i < count/4; i += blockDim.x) {((uint4 *)s_lut)[i] = ((uint4 const *)(lut + first))[i];}
	__syncthreads();
	return noise_tab_t{(char const *)s_lut, (char const *)s_lut};
}
template<int MODE> constexpr unsigned noise_lut_dwords() {return (MODE != MGEN_PERLIN) ? NOISE_LUT_S_DWORDS : NOISE_LUT_P_DWORDS;}
constexpr unsigned NG_ROWS = 16; // rows per block of k_noise_grid

template<int MODE> __global__ __launch_bounds__(256) void k_noise_grid(grid_job_t job, noise_consts_t nc, sin_lut_t L, float const *__restrict__ smx, float const *__restrict__ smy, float *__restrict__ out, uint32_t *__restrict__ mm,
	uint32_t const *__restrict__ lut, noise_oct_t oc)
{
	// Perlin sums on a regular grid: every sample position is a grid position, so the lattice cells a block touches are known up front and their four gradients
	// can be gathered once per block (noise_blocktab_build).  Measured on MI355X at 16384^2, 8 octaves: Perlin 80.3 -> 87.4 Gcells/s.  The same for simplex
	// LOSES (68.3 -> 61.0): its three-corner look-up saves less than building the records of the skewed lattice's bounding box costs; the warped sums of the
	// domain warp have no regular footprint at all.  Both keep the per-cell table path.
	constexpr bool REGULAR = (MODE == MGEN_PERLIN);
	__shared__ __attribute__((aligned(16))) uint32_t s_lut[noise_lut_dwords<MODE>()];
	__shared__ float s_brec[REGULAR ? NOISE_BT_FLOATS : 1];
	__shared__ noise_bt_meta_t s_bmeta[NUM_FREQ_COMP];
	noise_tab_t const ns = noise_stage_lut<MODE>(lut, s_lut);
	// two neighbouring cells of a row per lane: the lattice noise runs on register pairs (v_pk_mul_f32 / v_pk_add_f32), see terra_noise.hpp.
	// A block walks NG_ROWS rows, 4 at a time (one per wave), so the table staging is paid once per NG_ROWS x 128 cells.
	unsigned const x = (blockIdx.x*64 + (threadIdx.x & 63))*2;
	bool bt_ok = false; // block-uniform
	if (REGULAR) { // the gradient terms of every lattice cell under this block's 128 x NG_ROWS cells, per octave (noise_blocktab_build): the cells then skip the hash chains
		float const xy_scale = 0.0007f*nc.mesh_scale;
		if (job.mdx >= 0.0f && job.mdy >= 0.0f && nc.DX_VAL_INV >= 0.0f && nc.DY_VAL_INV >= 0.0f && xy_scale >= 0.0f) { // positions do not decrease along x and y: the block's corners bound them
			unsigned const bx0 = blockIdx.x*128, by0 = blockIdx.y*NG_ROWS + job.row0;
			float const vx0 = xy_scale*(((float)bx0*job.mdx + job.mx0)*nc.DX_VAL_INV), vx1 = xy_scale*(((float)(bx0 + 127)*job.mdx + job.mx0)*nc.DX_VAL_INV);
			float const vy0 = xy_scale*(((float)by0*job.mdy + job.my0)*nc.DY_VAL_INV), vy1 = xy_scale*(((float)(by0 + NG_ROWS - 1)*job.mdy + job.my0)*nc.DY_VAL_INV);
			bt_ok = noise_blocktab_build<(MODE != MGEN_PERLIN)>(vx0, vy0, vx1, vy1, oc, (char const *)s_lut, s_brec, s_bmeta, threadIdx.x, 256);
		}
	}
	noise_btab_t const bt{s_brec, s_bmeta};
	uint32_t mm_lo = 0xFFFFFFFFu, mm_hi = 0xFFFFFFFFu;
	for (unsigned ry = 0; ry < NG_ROWS; ry += 4) {
		unsigned const y = blockIdx.y*NG_ROWS + ry + (threadIdx.x >> 6);
		if (x < job.nx && y < job.ny) {
			float const xv0 = ((float)x*job.mdx + job.mx0)*nc.DX_VAL_INV, xv1 = ((float)(x + 1)*job.mdx + job.mx0)*nc.DX_VAL_INV, yval = ((float)(y + job.row0)*job.mdy + job.my0)*nc.DY_VAL_INV;
			nv2 zz;
			if constexpr (REGULAR) {zz = bt_ok ? noise_zval_bt<MODE>(nv2{xv0, xv1}, nv2{yval, yval}, job.shape, nc, oc, bt) : noise_zval_tab<MODE>(nv2{xv0, xv1}, nv2{yval, yval}, job.shape, nc, oc, ns);}
			else {zz = noise_zval_tab<MODE>(nv2{xv0, xv1}, nv2{yval, yval}, job.shape, nc, oc, ns);}
			float const z0 = finish_cell(zz[0], job, nc, L, smx, smy, x, y);
			out[(size_t)y*job.nx + x] = z0;
			minmax_acc(z0, mm_lo, mm_hi);
			if (x + 1 < job.nx) {
				float const z1 = finish_cell(zz[1], job, nc, L, smx, smy, x + 1, y);
				out[(size_t)y*job.nx + x + 1] = z1;
				minmax_acc(z1, mm_lo, mm_hi);
			}
		}
	}
	if (mm) {wave_minmax_publish(mm_lo, mm_hi, mm);}
}

// fBm tiles: the same two-cells-per-lane evaluation for a batch of tw x tw tile fields (origins per distinct tile column / row in m0)
template<int MODE> __global__ __launch_bounds__(256) void k_noise_tiles(tile_ref_pod_t const *__restrict__ refs, uint32_t n, uint32_t nux, float const *__restrict__ d_sm, float const *__restrict__ m0,
	grid_job_t job, noise_consts_t nc, sin_lut_t L, float *__restrict__ out, uint32_t tw, uint32_t const *__restrict__ lut, noise_oct_t oc)
{
	__shared__ __attribute__((aligned(16))) uint32_t s_lut[noise_lut_dwords<MODE>()];
	noise_tab_t const ns = noise_stage_lut<MODE>(lut, s_lut);
	uint32_t const pairs = (tw + 1)/2, per_tile = tw*pairs;
	size_t const i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= (size_t)n*per_tile) return;
	uint32_t const t = (uint32_t)(i / per_tile), p = (uint32_t)(i % per_tile), y = p / pairs, x = (p % pairs)*2;
	tile_ref_pod_t const r = refs[t];
	job.mx0 = m0[r.xi]; job.my0 = m0[nux + r.yi];
	float const xv0 = ((float)x*job.mdx + job.mx0)*nc.DX_VAL_INV, xv1 = ((float)(x + 1)*job.mdx + job.mx0)*nc.DX_VAL_INV, yval = ((float)y*job.mdy + job.my0)*nc.DY_VAL_INV;
	nv2 const zz = noise_zval_tab<MODE>(nv2{xv0, xv1}, nv2{yval, yval}, job.shape, nc, oc, ns);
	float const *smx = d_sm + (size_t)r.xi*tw, *smy = d_sm + (size_t)(nux + r.yi)*tw;
	float *o = out + (size_t)t*tw*tw + (size_t)y*tw + x;
	o[0] = finish_cell(zz[0], job, nc, L, smx, smy, x, y);
	if (x + 1 < tw) {o[1] = finish_cell(zz[1], job, nc, L, smx, smy, x + 1, y);}
}

// ------------------------------------------------------------------ K9: 3-D lattice field (voxel_manager::create_procedural, src/voxels.cpp:312-345), z fastest
// Two voxels of a column per lane, (z, z + 1): the lattice hashes and gradients come from the 3-D table in LDS (noise3_lut_fill: 5.8 KB per block, staged once for
// VN_CHUNKS x 256 pairs), Perlin's x / y lattice work is shared by the pair and the interpolation runs on register pairs.  One voxel per lane with the direct code was
// 2114 (Perlin) / 1721 (simplex) vector instructions per voxel for five octaves (profiles/r06_pmc_voxel_noise_summary.txt).
constexpr unsigned VN_CHUNKS = 8;
template<bool PERLIN> __global__ __launch_bounds__(256) void k_voxel_noise(float *__restrict__ out, size_t npairs, uint32_t nzp, vox_noise_job_t J, uint32_t const *__restrict__ lut) {
	__shared__ __attribute__((aligned(16))) uint32_t s_lut[NOISE3_LUT_DWORDS];
	for (unsigned i = threadIdx.x; i < NOISE3_LUT_DWORDS/4; i += 256) {((uint4 *)s_lut)[i] = ((uint4 const *)lut)[i];}
	__syncthreads();
	char const *const tab = (char const *)s_lut;
#pragma unroll 1
	for (unsigned c = 0; c < VN_CHUNKS; ++c) {
		size_t const i = ((size_t)blockIdx.x*VN_CHUNKS + c)*256 + threadIdx.x;
		if (i >= npairs) break;
		size_t const col = i / nzp;
		unsigned const z = (unsigned)(i - col*nzp)*2u, x = (unsigned)(col % J.nx), y = (unsigned)(col / J.nx) + J.y0;
		nv2 const v = voxel_noise_pair<PERLIN>(x, y, z, J, tab);
		float *o = out + col*J.nz + z;
		if (z + 1 < J.nz) {
			if ((J.nz & 1u) == 0) {*(nv2 *)o = v;} // (even nz: every pair starts on an 8-byte boundary)
			else {o[0] = v[0]; o[1] = v[1];}
		}
		else {o[0] = v[0];}
	}
}

} // namespace terra
