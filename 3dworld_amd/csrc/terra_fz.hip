// terra_fz.hip -- the TOLERANCE-mode build of the per-cell noise kernels (TERRA_GEN_FUSED / terra_set_option "gen.fused", include/terra.h).
//
// The second translation unit of libterra_hip.so.  It compiles terra_noise_kernels.hpp -- the very source terra_hip.hip compiles without contraction -- under the namespace
// name terra_fz with floating-point contraction allowed: k_noise_grid / k_noise_tiles (get_noise_zval's simplex / Perlin fBm, src/mesh_gen.cpp:706-751) evaluate the reference's expression trees with every a*b + c the compiler finds rounded once.
// The exact kernels pay a multiply AND an add for each of those; BASELINE's bar for the z values is 1e-5 relative, and the reference's own GPU off-load of these functions
// (shaders/simplex_noise.part) is not bit-equal to its CPU path either.  Nothing here is reachable unless the caller asks for the mode.
//
// The two builds exchange nothing but plain data: the launchers below take the kernel arguments as untyped pointers to the (layout-identical, same source) structs.
#define terra terra_fz
#pragma clang fp contract(fast)
#include "terra_noise_kernels.hpp"
#include "terra_fz_api.hpp"

int terra_fz_noise_grid(int mode, void const *job_, void const *nc_, void const *L_, float const *smx, float const *smy, float *out, uint32_t *mm, uint32_t const *nlut, void *stream_) {
	terra::grid_job_t const job = *(terra::grid_job_t const *)job_; terra::noise_consts_t const nc = *(terra::noise_consts_t const *)nc_; terra::sin_lut_t const L = *(terra::sin_lut_t const *)L_;
	hipStream_t const stream = (hipStream_t)stream_;
	dim3 const grid((job.nx + 127)/128, (job.ny + terra::NG_ROWS - 1)/terra::NG_ROWS), block(256);
	terra::noise_oct_t const oc = terra::make_noise_oct(nc);
	switch (job.mode) {
	case terra::MGEN_PERLIN:      hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_PERLIN>,      grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
	case terra::MGEN_DWARP_GPU:   hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_DWARP_GPU>,   grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
	case terra::MGEN_SIMPLEX_GPU: hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_SIMPLEX_GPU>, grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
	default:                      hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_SIMPLEX>,     grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
	}
	(void)mode;
	return (int)hipGetLastError();
}

int terra_fz_noise_tiles(void const *refs_, uint32_t n, uint32_t nux, float const *d_sm, float const *d_m0, void const *job_, void const *nc_, void const *L_, float *zvals, uint32_t tw, uint32_t const *nlut, void *stream_) {
	terra::grid_job_t const job = *(terra::grid_job_t const *)job_; terra::noise_consts_t const nc = *(terra::noise_consts_t const *)nc_; terra::sin_lut_t const L = *(terra::sin_lut_t const *)L_;
	terra::tile_ref_pod_t const *refs = (terra::tile_ref_pod_t const *)refs_;
	hipStream_t const stream = (hipStream_t)stream_;
	terra::noise_oct_t const oc = terra::make_noise_oct(nc);
	size_t const threads = (size_t)n*tw*((tw + 1)/2);
	dim3 const grid((unsigned)((threads + 255)/256)), block(256);
	switch (job.mode) {
	case terra::MGEN_PERLIN:      hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_PERLIN>,      grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
	case terra::MGEN_DWARP_GPU:   hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_DWARP_GPU>,   grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
	case terra::MGEN_SIMPLEX_GPU: hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_SIMPLEX_GPU>, grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
	default:                      hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_SIMPLEX>,     grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
	}
	return (int)hipGetLastError();
}
