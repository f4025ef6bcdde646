// tools/lat_probe.hip -- latency probes of ONE wave on gfx950: what a dependent VALU op, an IEEE sqrt / division, a VALU->SALU->VALU round trip, an LDS round trip and a
// data-dependent scalar branch cost when nothing else runs on the SIMD (the situation of a droplet wave).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 512
__global__ __launch_bounds__(64) void k_probe(float *io, unsigned long long *out, int which) {
	__shared__ float lds[256];
	float x = io[threadIdx.x], a = io[64], b = io[65];
	lds[threadIdx.x] = x; __syncthreads();
	unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	switch (which) {
	case 0: for (int i = 0; i < N_IT; ++i) {x = x*a; x = x + b;} break; // 2 dependent VALU ops
	case 1: for (int i = 0; i < N_IT; ++i) {x = sqrtf(x) + b;} break;
	case 2: for (int i = 0; i < N_IT; ++i) {x = a/x + b;} break;
	case 3: for (int i = 0; i < N_IT; ++i) {int s = __builtin_amdgcn_readfirstlane(__float_as_int(x)); s += 3; x = __int_as_float(s) + b;} break; // v->s->v + 1 add
	case 4: for (int i = 0; i < N_IT; ++i) {lds[threadIdx.x] = x; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); x = lds[threadIdx.x ^ 1] + b;} break; // LDS store -> load
	case 5: for (int i = 0; i < N_IT; ++i) { // uniform data-dependent branch with different bodies
			int s = __builtin_amdgcn_readfirstlane(__float_as_int(x));
			if (s & 0x1000) {x = x*a; asm volatile("s_nop 0");} else {x = x + b; asm volatile("s_nop 1");}
		} break;
	case 6: for (int i = 0; i < N_IT; ++i) {int l = __builtin_amdgcn_readfirstlane(__float_as_int(x)) & 15; x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)) + b;} break; // readlane with data-dependent lane
	case 7: for (int i = 0; i < N_IT; ++i) {x = floorf(x*a) + b;} break;
	case 8: for (int i = 0; i < N_IT; ++i) {x = (x < a) ? x + b : x*a;} break; // cmp + cndmask chain
	case 9: if (threadIdx.x < 16) {for (int i = 0; i < N_IT; ++i) {x = floorf(x*a) + b;}} break; // the same chain as 7 with 16 of the 64 lanes active (does the SIMD skip empty quarter-waves?)
	case 10: if (threadIdx.x < 1) {for (int i = 0; i < N_IT; ++i) {x = floorf(x*a) + b;}} break; // ... one lane
	}
	unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	io[threadIdx.x] = x;
	if (threadIdx.x == 0) {out[0] = c1 - c0; out[1] = w1 - w0;}
}
int main() {
	float h[66]; for (int i = 0; i < 64; ++i) h[i] = 1.5f + 0.01f*i; h[64] = 1.0001f; h[65] = 0.25f;
	float *d; unsigned long long *o; (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, 16);
	char const *names[] = {"mul+add (2 dependent VALU)", "sqrtf + add", "a/x + add", "readfirstlane, s_add, v_add", "LDS store, fence, load, add", "readfirstlane + uniform branch", "readfirstlane, readlane, add", "mul, floor, add", "cmp, 2 ops, cndmask", "mul, floor, add, lanes 0-15 only", "mul, floor, add, lane 0 only"};
	for (int w = 0; w < 11; ++w) for (int rep = 0; rep < 2; ++rep) {
		(void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, o, w);
		unsigned long long r[2]; (void)hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
		if (rep) printf("%-34s %7.1f cycles/iter  %7.2f ns/iter  (%.0f MHz)\n", names[w], (double)r[0]/N_IT, 10.0*r[1]/N_IT, r[1] ? 100.0*r[0]/r[1] : 0.0);
	}
	return 0;
}
