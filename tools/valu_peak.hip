// microbenchmark: fp32 VALU issue rates on gfx950 for the instruction mix the sine kernel is allowed to use (no FMA)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do {hipError_t e = (x); if (e != hipSuccess) {printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;}} while (0)
template<int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 acc[16]; f2 x = {a, b}, y = {b, a};
	for (int i = 0; i < 16; ++i) {acc[i] = (f2){a + i + threadIdx.x, b};}
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			if (MODE == 0) {f2 p; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(acc[i])); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(p), "v"(y));}
			if (MODE == 1) {float p0, p1; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(x.x), "v"(acc[i].x)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(x.y), "v"(acc[i].y));
				asm volatile("v_add_f32 %0, %1, %2" : "=v"(acc[i].x) : "v"(p0), "v"(y.x)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(acc[i].y) : "v"(p1), "v"(y.y));}
			if (MODE == 2) {asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(acc[i]) : "v"(x), "v"(acc[i]), "v"(y));}
			if (MODE == 3) {asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(acc[i].x) : "v"(x.x), "v"(acc[i].x), "v"(y.x)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(acc[i].y) : "v"(x.y), "v"(acc[i].y), "v"(y.y));}
		}
	}
	float s = 0; for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}
template<int MODE> double run(float *d, int blocks, int iters, const char *name, double ops_per_iter_lane) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.9999f);
	hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double ops = (double)blocks*256*iters*ops_per_iter_lane;
	printf("%-28s %8.3f ms  %8.2f Tops/s (mul+add counted separately; fma = 2)\n", name, ms, ops/ms/1e9);
	return ops/ms/1e9;
}
int main() {
	float *d; CHECK(hipMalloc(&d, 256*8*256*4*8));
	for (int blocks : {256*4, 256*8}) {
		printf("blocks %d (x256 threads)\n", blocks);
		run<0>(d, blocks, 20000, "v_pk_mul_f32 + v_pk_add_f32", 16*4.0);
		run<1>(d, blocks, 20000, "v_mul_f32 + v_add_f32", 16*4.0);
		run<2>(d, blocks, 20000, "v_pk_fma_f32", 16*4.0);
		run<3>(d, blocks, 20000, "v_fma_f32", 16*4.0);
	}
	return 0;
}
