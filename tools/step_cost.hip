// tools/step_cost.hip -- micro-benchmark: wall time per droplet step of the wave-cooperative erosion code on one wave (MI355X).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I3dworld_amd/csrc tools/step_cost.hip -o /tmp/step_cost && /tmp/step_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "terra_erosion.hpp"
using namespace terra;

template<int VARIANT> __global__ __launch_bounds__(64) void k_steps(float *grid, erosion_consts_t ec, unsigned budget, unsigned long long *out) {
	__shared__ __attribute__((aligned(16))) float win[2*EW*EW];
	__shared__ uint8_t dirty[2*EW*EW];
	extern __shared__ __attribute__((aligned(16))) float pad[];
	droplet_state_t d;
	unsigned long long t0 = 0, t1 = 0;
	unsigned steps = 0;
	if (VARIANT == 0) { // tile mode: whole grid in LDS
		for (int i = threadIdx.x; i < ec.NX*ec.NY; i += 64) pad[i] = grid[i];
		__syncthreads();
		wave_lds_mem_t m; m.pad = pad; m.NX = ec.NX; m.NY = ec.NY;
		droplet_start(7, m, ec, d);
		d.xi = ec.NX/2; d.zi = 8; d.xp = (float)d.xi; d.zp = (float)d.zi; float c[4]; m.corners(d.xi, d.zi, c); d.h = d.h00 = c[0]; d.h10 = c[1]; d.h01 = c[2]; d.h11 = c[3];
		t0 = wall_clock64();
		droplet_run_fast(d, m, ec, budget);
		t1 = wall_clock64();
	}
	else { // window mode, backing store = the grid in HBM
		grid_view_t g; g.interior = grid; g.border = nullptr; g.xsize = ec.NX; g.ysize = ec.NY; g.NX = ec.NX; g.NY = ec.NY;
		window_mem_t<grid_back_t> m; m.init(win, dirty, ec.NX, ec.NY); m.back.g = g; m.back.touched = nullptr; m.back.touched_count = nullptr; m.back.touched_cap = 0;
		droplet_start(7, m, ec, d);
		d.xi = ec.NX/2; d.zi = 8; d.xp = (float)d.xi; d.zp = (float)d.zi; m.begin_step(d.xi, d.zi); float c[4]; m.corners(d.xi, d.zi, c); d.h = d.h00 = c[0]; d.h10 = c[1]; d.h01 = c[2]; d.h11 = c[3];
		t0 = wall_clock64();
		droplet_run_fast(d, m, ec, budget);
		t1 = wall_clock64();
		m.finish();
	}
	steps = d.numMoves;
	if (threadIdx.x == 0) {out[0] = t1 - t0; out[1] = steps;}
}

int main() {
	int const N = 136;
	std::vector<float> h((size_t)N*N);
	for (int z = 0; z < N; ++z) for (int x = 0; x < N; ++x) h[(size_t)z*N + x] = 10.0f - 0.05f*z + 0.01f*sinf(0.7f*x) + 0.02f*cosf(0.9f*z); // a slope the droplet runs down
	float *dg; unsigned long long *dout; hipMalloc(&dg, h.size()*4); hipMalloc(&dout, 16);
	erosion_consts_t ec{};
	ec.xsize = N - 8; ec.ysize = N - 8; ec.NX = N; ec.NY = N; ec.max_path_len = 4u*N*N; ec.erode_amount = 1.0f; ec.water_thresh = -100.0f;
	ec.relh_adj_tex = 0; ec.zmin = 0; ec.zrange = 10; ec.clip_hd1 = 0.5f; ec.two_pi = 6.2831855f; ec.min_zval = -100;
	make_rock_threshold(ec);
	hipFuncSetAttribute((void const *)k_steps<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96*1024);
	for (int variant = 0; variant < 2; ++variant) for (int rep = 0; rep < 3; ++rep) {
		hipMemcpy(dg, h.data(), h.size()*4, hipMemcpyHostToDevice);
		if (variant == 0) hipLaunchKernelGGL(k_steps<0>, dim3(1), dim3(64), N*N*4, 0, dg, ec, 100u, dout);
		else hipLaunchKernelGGL(k_steps<1>, dim3(1), dim3(64), 0, 0, dg, ec, 100u, dout);
		unsigned long long o[2]; hipMemcpy(o, dout, 16, hipMemcpyDeviceToHost);
		printf("variant %d (%s): %llu steps, %.1f ns/step (wall clock 100 MHz ticks %llu)\n", variant, variant ? "32x32 LDS window over HBM grid" : "whole tile in LDS", o[1], o[1] ? 10.0*o[0]/o[1] : 0.0, o[0]);
	}
	return 0;
}
