// Issue-rate microbenchmark for the packed fp32 forms the kernels use: cycles per instruction per wave with W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/pk_rate.hip -o tools/_bin/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template<int V> __global__ __launch_bounds__(256) void k(float *out, float const *in, int iters, long long *cyc) {
	v2f acc = {in[threadIdx.x], in[threadIdx.x + 1]}, zv = {in[threadIdx.x + 2], in[threadIdx.x + 3]}, t, pv = {in[threadIdx.x + 4], in[threadIdx.x + 5]};
	v2f ps = {in[0], in[1]}; // uniform -> SGPR pair
	float a1 = acc.x, t1;
	long long const t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int j = 0; j < 32; ++j) {
			if (V == 0) asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(t) : "s"(ps), "v"(zv));
			if (V == 1) asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(t) : "v"(pv), "v"(zv));
			if (V == 2) asm volatile("v_mul_f32 %1, %2, %3\n\tv_add_f32 %0, %0, %1" : "+v"(a1), "=&v"(t1) : "s"(ps.x), "v"(zv.x));
			if (V == 3) asm volatile("v_pk_mul_f32 %1, %2, %3\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(t) : "v"(pv), "v"(zv));
			if (V == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(pv), "v"(zv));
			if (V == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(pv.x), "v"(zv.x));
		}
	}
	long long const t1c = __builtin_readcyclecounter();
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1c - t0;
	out[blockIdx.x*256 + threadIdx.x] = acc.x + acc.y + a1;
}
int main() {
	float *in, *out; long long *cyc, h;
	hipMalloc(&in, 4096); hipMemset(in, 0, 4096); hipMalloc(&out, 256*4096*4*8); hipMalloc(&cyc, 8);
	int const iters = 2000;
	for (int v = 0; v < 6; ++v) for (int wps = 1; wps <= 8; wps *= 2) { // wps waves per SIMD: blocks of 256 = 4 waves (1 per SIMD); 256 CUs
		int const blocks = 256*wps;
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			switch (v) {case 0: k<0><<<blocks, 256>>>(out, in, iters, cyc); break; case 1: k<1><<<blocks, 256>>>(out, in, iters, cyc); break; case 2: k<2><<<blocks, 256>>>(out, in, iters, cyc); break;
				case 3: k<3><<<blocks, 256>>>(out, in, iters, cyc); break; case 4: k<4><<<blocks, 256>>>(out, in, iters, cyc); break; default: k<5><<<blocks, 256>>>(out, in, iters, cyc);}
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
		int const ninstr = iters*32*((v >= 4) ? 1 : 2);
		printf("variant %d waves/SIMD %d: %.3f ms, wall ns per instr per wave-slot %.3f, shader-clock ticks/instr (wave 0) %.2f\n", v, wps, ms, ms*1e6/ninstr/wps, (double)h/ninstr);
	}
	return 0;
}
