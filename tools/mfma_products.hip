// What the matrix pipe can be for a sum that must round like mul-then-add (the k_sine_grid_mx experiment of round 4: profiles/r04_sine_matrix_pipe.txt):
//   A  exactness + layout: v_mfma_f32_32x32x1_2b_f32 with C = 0 against the host's float multiply, bit for bit, over random operand sets that include subnormal
//      inputs / products, zeros, infinities and NaNs; the register -> (row, column) map the kernel's epilogue assumes
//   B  rates: one matrix instruction (2048 products) beside 16 v_pk_add_f32 / 32 v_add_f32, against 16 v_pk_mul_f32 + 16 v_pk_add_f32 on the vector ALU alone,
//      each pipe alone, with 1 and 2 waves per SIMD -- do the two pipes run side by side, and is the packed add slowed down beside a matrix instruction?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_products.hip -o tools/_bin/mfma_products
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef float v32f __attribute__((ext_vector_type(32)));

__global__ __launch_bounds__(64) void k_products(float const *a, float const *b, float *d) { // one wave per operand set: a[64] rows, b[32] columns
	unsigned const lane = threadIdx.x, set = blockIdx.x;
	v32f zero; for (int v = 0; v < 32; ++v) zero[v] = 0.0f;
	v32f const r = __builtin_amdgcn_mfma_f32_32x32x1f32(a[set*64 + lane], b[set*32 + (lane & 31)], zero, 0, 0, 0);
	for (int v = 0; v < 32; ++v) d[((size_t)set*64 + lane)*32 + v] = r[v];
}

#define STR2(x) #x
#define STR(x) STR2(x)
#define PKADD1(ACC, P, O) "v_pk_add_f32 v[" STR(ACC) "+" STR(O) ":" STR(ACC) "+" STR(O) "+1], v[" STR(ACC) "+" STR(O) ":" STR(ACC) "+" STR(O) "+1], v[" STR(P) "+" STR(O) ":" STR(P) "+" STR(O) "+1]\n\t"
#define PKADD4(ACC, P, O) PKADD1(ACC, P, O) PKADD1(ACC, P, O+2) PKADD1(ACC, P, O+4) PKADD1(ACC, P, O+6)
#define PKADD16(ACC, P) PKADD4(ACC, P, 0) PKADD4(ACC, P, 8) PKADD4(ACC, P, 16) PKADD4(ACC, P, 24)
#define PKMUL1(P, O) "v_pk_mul_f32 v[" STR(P) "+" STR(O) ":" STR(P) "+" STR(O) "+1], %[x], %[y] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
#define PKMUL4(P, O) PKMUL1(P, O) PKMUL1(P, O+2) PKMUL1(P, O+4) PKMUL1(P, O+6)
#define PKMUL16(P) PKMUL4(P, 0) PKMUL4(P, 8) PKMUL4(P, 16) PKMUL4(P, 24)
#define ADD1(ACC, P, O) "v_add_f32 v[" STR(ACC) "+" STR(O) "], v[" STR(ACC) "+" STR(O) "], v[" STR(P) "+" STR(O) "]\n\t"
#define ADD8(ACC, P, O) ADD1(ACC, P, O) ADD1(ACC, P, O+1) ADD1(ACC, P, O+2) ADD1(ACC, P, O+3) ADD1(ACC, P, O+4) ADD1(ACC, P, O+5) ADD1(ACC, P, O+6) ADD1(ACC, P, O+7)
#define ADD32(ACC, P) ADD8(ACC, P, 0) ADD8(ACC, P, 8) ADD8(ACC, P, 16) ADD8(ACC, P, 24)
#define MUL(P) "v_mfma_f32_32x32x1_2b_f32 v[" STR(P) ":" STR(P) "+31], %[a], %[b], 0\n\t"
#define INIT MUL(128) MUL(160) MUL(192) MUL(224) "s_nop 15\n\ts_nop 15\n\t"
#define TAIL "s_nop 15\n\ts_nop 15\n\t"
#define CLOB "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
// a step = 4096 cell-terms (what one k step of a 64 x 64 wave tile is): sums in v[64:127], product sets v[128:255]
// V 0: 2 matrix instructions + 32 v_pk_add (the kernel's loop)   1: 2 matrix + 64 v_add_f32   2: 32 v_pk_mul + 32 v_pk_add (vector ALU alone, today's kernel)
//   3: 2 matrix instructions alone   4: 32 v_pk_add alone   5: 64 v_add_f32 alone
template<int V> __global__ __launch_bounds__(256) void k_rate(float *out, float const *in, int steps, long long *cyc) {
	float a = in[threadIdx.x], b = in[threadIdx.x + 64];
	typedef float v2f __attribute__((ext_vector_type(2)));
	v2f x = {in[threadIdx.x + 1], in[threadIdx.x + 2]}, y = {in[threadIdx.x + 3], in[threadIdx.x + 4]};
	v32f acc0, acc1; for (int v = 0; v < 32; ++v) {acc0[v] = 0.0f; acc1[v] = 0.0f;}
	int left = steps/2;
	long long const t0 = __builtin_readcyclecounter();
	if (V == 0) asm volatile(INIT ".Lv0_%=:\n\t" MUL(192) PKADD16(64, 128) MUL(224) PKADD16(96, 160) MUL(128) PKADD16(64, 192) MUL(160) PKADD16(96, 224)
		"s_sub_i32 %[left], %[left], 1\n\ts_cmp_gt_i32 %[left], 0\n\ts_cbranch_scc1 .Lv0_%=\n\t" TAIL : "+{v[64:95]}"(acc0), "+{v[96:127]}"(acc1), [left] "+s"(left) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y) : "scc", "memory", CLOB);
	if (V == 1) asm volatile(INIT ".Lv1_%=:\n\t" MUL(192) ADD32(64, 128) MUL(224) ADD32(96, 160) MUL(128) ADD32(64, 192) MUL(160) ADD32(96, 224)
		"s_sub_i32 %[left], %[left], 1\n\ts_cmp_gt_i32 %[left], 0\n\ts_cbranch_scc1 .Lv1_%=\n\t" TAIL : "+{v[64:95]}"(acc0), "+{v[96:127]}"(acc1), [left] "+s"(left) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y) : "scc", "memory", CLOB);
	if (V == 2) asm volatile(INIT ".Lv2_%=:\n\t" PKMUL16(192) PKADD16(64, 128) PKMUL16(224) PKADD16(96, 160) PKMUL16(128) PKADD16(64, 192) PKMUL16(160) PKADD16(96, 224)
		"s_sub_i32 %[left], %[left], 1\n\ts_cmp_gt_i32 %[left], 0\n\ts_cbranch_scc1 .Lv2_%=\n\t" TAIL : "+{v[64:95]}"(acc0), "+{v[96:127]}"(acc1), [left] "+s"(left) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y) : "scc", "memory", CLOB);
	if (V == 3) asm volatile(INIT ".Lv3_%=:\n\t" MUL(192) MUL(224) MUL(128) MUL(160)
		"s_sub_i32 %[left], %[left], 1\n\ts_cmp_gt_i32 %[left], 0\n\ts_cbranch_scc1 .Lv3_%=\n\t" TAIL : "+{v[64:95]}"(acc0), "+{v[96:127]}"(acc1), [left] "+s"(left) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y) : "scc", "memory", CLOB);
	if (V == 4) asm volatile(INIT ".Lv4_%=:\n\t" PKADD16(64, 128) PKADD16(96, 160) PKADD16(64, 192) PKADD16(96, 224)
		"s_sub_i32 %[left], %[left], 1\n\ts_cmp_gt_i32 %[left], 0\n\ts_cbranch_scc1 .Lv4_%=\n\t" TAIL : "+{v[64:95]}"(acc0), "+{v[96:127]}"(acc1), [left] "+s"(left) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y) : "scc", "memory", CLOB);
	if (V == 5) asm volatile(INIT ".Lv5_%=:\n\t" ADD32(64, 128) ADD32(96, 160) ADD32(64, 192) ADD32(96, 224)
		"s_sub_i32 %[left], %[left], 1\n\ts_cmp_gt_i32 %[left], 0\n\ts_cbranch_scc1 .Lv5_%=\n\t" TAIL : "+{v[64:95]}"(acc0), "+{v[96:127]}"(acc1), [left] "+s"(left) : [a] "v"(a), [b] "v"(b), [x] "v"(x), [y] "v"(y) : "scc", "memory", CLOB);
	long long const t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
	float s = 0.0f; for (int v = 0; v < 32; ++v) s += acc0[v] + acc1[v];
	out[blockIdx.x*256 + threadIdx.x] = s;
}

static uint32_t bits(float f) {uint32_t u; memcpy(&u, &f, 4); return u;}
static float from_bits(uint32_t u) {float f; memcpy(&f, &u, 4); return f;}

int main() {
	// ---- A: exactness and layout
	int const sets = 4096;
	std::mt19937 rng(12345);
	std::vector<float> ha(sets*64), hb(sets*32);
	auto gen = [&](int set) -> float {
		uint32_t const r = rng();
		int const kind = (set < 64) ? 0 : (int)(rng() % 16); // the first sets: plain values (a layout error shows there already)
		switch (kind) {
		case 1: return from_bits((r & 0x807FFFFFu));                              // subnormal (or zero)
		case 2: return from_bits((r & 0x807FFFFFu) | ((uint32_t)(1 + rng() % 40) << 23));  // tiny normal: products underflow
		case 3: return (r & 1) ? 0.0f : -0.0f;
		case 4: return from_bits((r & 0x80000000u) | 0x7F800000u);                // infinity
		case 5: return from_bits(r | 0x7FC00000u);                                // NaN
		case 6: return from_bits((r & 0x807FFFFFu) | ((uint32_t)(215 + rng() % 39) << 23)); // huge: products overflow
		default: return from_bits((r & 0x807FFFFFu) | ((uint32_t)(100 + rng() % 56) << 23)); // exponents -27 .. 28
		}
	};
	for (int s = 0; s < sets; ++s) {for (int i = 0; i < 64; ++i) ha[s*64 + i] = gen(s); for (int j = 0; j < 32; ++j) hb[s*32 + j] = gen(s);}
	float *da, *db, *dd; hipMalloc(&da, ha.size()*4); hipMalloc(&db, hb.size()*4); hipMalloc(&dd, (size_t)sets*64*32*4);
	hipMemcpy(da, ha.data(), ha.size()*4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size()*4, hipMemcpyHostToDevice);
	k_products<<<sets, 64>>>(da, db, dd);
	std::vector<float> hd((size_t)sets*64*32); hipMemcpy(hd.data(), dd, hd.size()*4, hipMemcpyDeviceToHost);
	uint64_t n = 0, bad = 0, bad_plain = 0, zero_sign = 0, subn = 0, nan_payload = 0;
	for (int s = 0; s < sets; ++s) for (int l = 0; l < 64; ++l) for (int v = 0; v < 32; ++v) {
		int const row = 32*(v >> 4) + 8*((v & 15) >> 2) + 4*(l >> 5) + (v & 3), col = l & 31; // the map k_sine_grid_mx's epilogue assumes
		volatile float const p = ha[s*64 + row]*hb[s*32 + col]; // one IEEE rounding (x86 SSE, subnormals kept)
		float const g = hd[((size_t)s*64 + l)*32 + v];
		++n;
		if (p != 0.0f && std::fabs(p) < 1.17549435e-38f) ++subn;
		if (bits(p) == bits(g)) continue;
		if (p == 0.0f && g == 0.0f) {++zero_sign; continue;}   // fma(a, b, +0): -0 products come out as +0 (no accumulator can tell, see the kernel's comment)
		if (p != p && g != g) {++nan_payload; continue;}
		++bad; if (s < 64) ++bad_plain;
		if (bad <= 8) printf("  mismatch set %d lane %d reg %d: a %08x b %08x host %08x device %08x\n", s, l, v, bits(ha[s*64 + row]), bits(hb[s*32 + col]), bits(p), bits(g));
	}
	printf("A products: %llu compared, %llu mismatches (%llu in the plain sets = layout), %llu zero-sign (+0 for -0), %llu NaN payload only, %llu subnormal products among them all -> %s\n",
		(unsigned long long)n, (unsigned long long)bad, (unsigned long long)bad_plain, (unsigned long long)zero_sign, (unsigned long long)nan_payload, (unsigned long long)subn, bad ? "MISMATCH" : "EXACT, LAYOUT OK");
	// ---- B: rates
	float *in, *out; long long *cyc, h;
	hipMalloc(&in, 4096); {std::vector<float> hin(1024); for (int i = 0; i < 1024; ++i) hin[i] = 0.37f + 0.0011f*(float)i; hipMemcpy(in, hin.data(), 4096, hipMemcpyHostToDevice);} hipMalloc(&out, 256*2*256*4); hipMalloc(&cyc, 8);
	int const steps = 20000;
	char const *names[6] = {"2 matrix + 32 v_pk_add (kernel loop)", "2 matrix + 64 v_add_f32", "32 v_pk_mul + 32 v_pk_add (vector ALU alone)", "2 matrix alone", "32 v_pk_add alone", "64 v_add_f32 alone"};
	for (int v = 0; v < 6; ++v) for (int wps = 1; wps <= 2; ++wps) {
		int const blocks = 256*wps;
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			switch (v) {case 0: k_rate<0><<<blocks, 256>>>(out, in, steps, cyc); break; case 1: k_rate<1><<<blocks, 256>>>(out, in, steps, cyc); break; case 2: k_rate<2><<<blocks, 256>>>(out, in, steps, cyc); break;
				case 3: k_rate<3><<<blocks, 256>>>(out, in, steps, cyc); break; case 4: k_rate<4><<<blocks, 256>>>(out, in, steps, cyc); break; default: k_rate<5><<<blocks, 256>>>(out, in, steps, cyc);}
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
		printf("B %-46s waves/SIMD %d: %8.3f ms  %7.1f ns per step per SIMD  %7.1f s_memtime ticks per step (wave 0)  -> %6.1f G cell-terms/s chip\n", names[v], wps, ms, ms*1e6/steps/wps, (double)h/steps,
			(double)steps*4096.0*1024.0*wps/(ms*1e-3)/1e9);
	}
	return 0;
}
