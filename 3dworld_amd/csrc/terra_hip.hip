// terra_hip.hip -- libterra_hip.so: HIP backend (gfx950 / MI355X) + the C ABI of include/terra.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off terra_hip.hip -o libterra_hip.so
// (see 3dworld_amd/build.py).  There is no host execution path in this library: every entry point needs a HIP device.
#include "terra_kernels.hpp"
#include "terra_simple_paths.hpp"
#include <stdlib.h>

#define TERRA_HIP_CHECK(expr) do {hipError_t const e_ = (expr); if (e_ != hipSuccess) {throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));}} while (0)

struct hip_backend_t : terra::simple_paths<hip_backend_t> {
	int device = -1;
	hipStream_t stream = nullptr, own_stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool simple_kernels = false; // TERRA_SIMPLE_KERNELS=1: run the one-thread-per-cell cross-check kernels instead of the LDS-tiled ones
	float *tile_pad = nullptr; size_t tile_pad_bytes = 0;

	static int device_count() {int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n;}
	void init(int dev) {
		device = dev;
		TERRA_HIP_CHECK(hipSetDevice(dev));
		TERRA_HIP_CHECK(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
		stream = own_stream;
		TERRA_HIP_CHECK(hipEventCreate(&ev0)); TERRA_HIP_CHECK(hipEventCreate(&ev1));
		char const *s = getenv("TERRA_SIMPLE_KERNELS");
		simple_kernels = (s && s[0] == '1');
		// LDS-tiled kernels use > 64 KiB of dynamic LDS (160 KiB per CU on gfx950)
		TERRA_HIP_CHECK(hipFuncSetAttribute((void const *)terra::k_tile_erosion, hipFuncAttributeMaxDynamicSharedMemorySize, 96*1024));
	}
	~hip_backend_t() {
		if (tile_pad) (void)hipFree(tile_pad);
		if (ev0) (void)hipEventDestroy(ev0);
		if (ev1) (void)hipEventDestroy(ev1);
		if (own_stream) (void)hipStreamDestroy(own_stream);
	}
	void use() {TERRA_HIP_CHECK(hipSetDevice(device));}
	void set_stream(void *s) {sync(); stream = s ? (hipStream_t)s : own_stream;}
	void sync() {use(); TERRA_HIP_CHECK(hipStreamSynchronize(stream));}
	void *alloc(size_t bytes) {use(); void *p = nullptr; TERRA_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1)); return p;}
	void free(void *p) {use(); (void)hipFree(p);}
	void fill32(void *p, uint32_t v, size_t count) {use(); if (count) TERRA_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)p, (int)v, count, stream));}
	// host buffers are ordinary pageable memory (often stack variables): copies are stream-ordered and then waited for
	void h2d(void *d, void const *h, size_t bytes) {use(); TERRA_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream)); TERRA_HIP_CHECK(hipStreamSynchronize(stream));}
	void d2h(void *h, void const *d, size_t bytes) {use(); TERRA_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream)); TERRA_HIP_CHECK(hipStreamSynchronize(stream));}
	void timer_start() {use(); TERRA_HIP_CHECK(hipEventRecord(ev0, stream));}
	float timer_stop() {use(); TERRA_HIP_CHECK(hipEventRecord(ev1, stream)); TERRA_HIP_CHECK(hipEventSynchronize(ev1)); float ms = 0; TERRA_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); return ms;}

	template<class F> void launch(size_t n, F f, int block = 256) {
		if (n == 0) return;
		use();
		size_t const nblocks = (n + block - 1)/block;
		if (nblocks > 0x7FFFFFFFull) throw std::invalid_argument("launch: grid too large");
		hipLaunchKernelGGL(terra::k_generic<F>, dim3((unsigned)nblocks), dim3(block), 0, stream, n, f);
		TERRA_HIP_CHECK(hipGetLastError());
	}

	template<class F> void launch_waves(size_t n, F f) {
		if (n == 0) return;
		use();
		if (n > 0x7FFFFFFFull) throw std::invalid_argument("launch_waves: grid too large");
		hipLaunchKernelGGL(terra::k_waves<F>, dim3((unsigned)n), dim3(64), 0, stream, f, 0u);
		TERRA_HIP_CHECK(hipGetLastError());
	}

	// mm (optional): device uint32[2] pre-set to 0xFFFFFFFF receiving min f2ord(z) / min ~f2ord(z); returns false when the caller must run minmax() itself
	bool sine_grid(terra::grid_job_t const &job, terra::noise_consts_t const &nc, terra::sin_lut_t const &L, float const *xt, float const *yt, float const *smx, float const *smy, float *out, uint32_t *mm) {
		if (simple_kernels) {sine_grid_simple(job, nc, L, xt, yt, smx, smy, out); return false;}
		use();
		unsigned const ntx = job.nxp/terra::SG_BX, nty = (job.ny + terra::SG_BY - 1)/terra::SG_BY;
		unsigned const nb = ntx*nty, grid = ((nb + 7)/8)*8;
		hipLaunchKernelGGL(terra::k_sine_grid, dim3(grid), dim3(terra::SG_THREADS), 0, stream, job, nc, L, xt, yt, smx, smy, out, ntx, nty, mm);
		TERRA_HIP_CHECK(hipGetLastError());
		return true;
	}
	bool noise_grid(terra::grid_job_t const &job, terra::noise_consts_t const &nc, terra::sin_lut_t const &L, float const *smx, float const *smy, float *out, uint32_t *mm) {
		if (simple_kernels) {noise_grid_simple(job, nc, L, smx, smy, out); return false;}
		use();
		dim3 const grid((job.nx + 63)/64, (job.ny + 3)/4), block(256);
		switch (job.mode) {
		case terra::MGEN_PERLIN:      hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_PERLIN>,      grid, block, 0, stream, job, nc, L, smx, smy, out, mm); break;
		case terra::MGEN_DWARP_GPU:   hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_DWARP_GPU>,   grid, block, 0, stream, job, nc, L, smx, smy, out, mm); break;
		case terra::MGEN_SIMPLEX_GPU: hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_SIMPLEX_GPU>, grid, block, 0, stream, job, nc, L, smx, smy, out, mm); break;
		default:                      hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_SIMPLEX>,     grid, block, 0, stream, job, nc, L, smx, smy, out, mm); break;
		}
		TERRA_HIP_CHECK(hipGetLastError());
		return true;
	}
	void tile_grid(uint32_t n, terra::tile_ref_pod_t const *refs, uint32_t nux, float const *d_tab, float const *d_sm, float const *d_m0, int md, int shp, int kstart, bool use_sm, float so,
		terra::noise_consts_t const &nc, terra::sin_lut_t const &L, float dxv, float dyv, float *zvals) {tile_grid_simple(n, refs, nux, d_tab, d_sm, d_m0, md, shp, kstart, use_sm, so, nc, L, dxv, dyv, zvals);}
	void tile_erosion(uint32_t n, float *zvals, terra::erosion_consts_t const &ec, uint32_t iters) {
		use();
		size_t const lds = (size_t)ec.NX*ec.NY*sizeof(float);
		if (simple_kernels || lds > 96*1024) { // cross-check path: padded scratch in HBM
			size_t const bytes = (size_t)n*lds;
			if (bytes > tile_pad_bytes) {if (tile_pad) {sync(); (void)hipFree(tile_pad);} TERRA_HIP_CHECK(hipMalloc((void **)&tile_pad, bytes)); tile_pad_bytes = bytes;}
			tile_erosion_simple(n, zvals, ec, iters, tile_pad);
			return;
		}
		hipLaunchKernelGGL(terra::k_tile_erosion, dim3(n), dim3(64), lds, stream, zvals, ec, iters);
		TERRA_HIP_CHECK(hipGetLastError());
	}
	void minmax(float const *vals, size_t n, uint32_t *d) {
		if (simple_kernels || ((uintptr_t)vals & 15)) {minmax_simple(vals, n, d); return;}
		use();
		unsigned const blocks = (unsigned)std::min<size_t>((n/4 + 255)/256 + 1, 256*8);
		hipLaunchKernelGGL(terra::k_minmax, dim3(blocks), dim3(256), 0, stream, vals, n, d);
		TERRA_HIP_CHECK(hipGetLastError());
	}
	void voxel_sines(float *out, uint32_t nx, uint32_t ny, uint32_t nz, float const *d_tab, float zscale, int normalize) {
		if (simple_kernels) {voxel_sines_simple(out, nx, ny, nz, d_tab, zscale, normalize); return;}
		use();
		unsigned const block = (nz >= 256) ? 256 : ((nz + 63)/64)*64;
		hipLaunchKernelGGL(terra::k_voxel_sines, dim3(nx, ny), dim3(block), 0, stream, out, nx, ny, nz, d_tab, zscale, normalize);
		TERRA_HIP_CHECK(hipGetLastError());
	}
};
typedef hip_backend_t terra_backend_t;
#include "terra_api_impl.hpp"
