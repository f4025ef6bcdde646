// terra_common.hpp -- shared host/device helpers for libterra_hip (MI355X / gfx950).
//
// Everything here is arithmetic that must agree bit-for-bit with the reference CPU path
// (g++ -O3, x86-64 SSE2 scalar float, NO fused multiply-add): the whole library is compiled with
// -ffp-contract=off and the few places where the C++ source promotes to double are written out explicitly.
// Citations are relative to the 3DWorld reference tree.
#pragma once
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <limits.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TERRA_HD __host__ __device__ __forceinline__
#define TERRA_D  __device__ __forceinline__
#define TERRA_HD_COLD __host__ __device__ __attribute__((noinline)) // rarely executed and large: a real call keeps its registers out of the caller's hot loop
#define TERRA_LAMBDA __host__ __device__
#else
#define TERRA_HD inline
#define TERRA_HD_COLD inline
#define TERRA_D  inline
#define TERRA_LAMBDA
#endif

// branch-probability hints: they only steer basic-block placement, so that the step loop of a droplet trace stays one compact run of
// instructions (the rarely taken window shifts, multi-version look-ups and libm restatements are laid out behind it)
#define TERRA_LIKELY(x)   __builtin_expect(!!(x), 1)
#define TERRA_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace terra {

constexpr int   F_TABLE_SIZE   = 90;     // NUM_FREQ_COMP(9)*N_RAND_SIN2(10), src/mesh_gen.cpp:14,16,30
constexpr int   NUM_FREQ_COMP  = 9;
constexpr int   N_RAND_SIN2    = 10;
constexpr int   TSIZE          = 1 << 15; // src/sinf.h:8
constexpr float PI_F           = 3.141592654f; // src/3DWorld.h:43
constexpr int   EROSION_PAD    = 4;      // src/erosion.cpp:25

enum {MGEN_SINE = 0, MGEN_SIMPLEX, MGEN_PERLIN, MGEN_SIMPLEX_GPU, MGEN_DWARP_GPU}; // src/3DWorld.h:1399

// std::min / std::max semantics (NaN-propagation exactly as "(b<a)?b:a" / "(a<b)?b:a"), NOT fminf/fmaxf
TERRA_HD float min_std(float a, float b) {return (b < a) ? b : a;}
TERRA_HD float max_std(float a, float b) {return (a < b) ? b : a;}
TERRA_HD int   imin(int a, int b) {return (b < a) ? b : a;}
TERRA_HD int   imax(int a, int b) {return (a < b) ? b : a;}
TERRA_HD float clip01(float x)  {return max_std(0.0f, min_std(1.0f, x));}    // CLIP_TO_01  src/3DWorld.h:148
TERRA_HD float clip_pm1(float x) {return max_std(-1.0f, min_std(1.0f, x));}  // CLIP_TO_pm1 src/3DWorld.h:149
// float -> int with x86 cvttss2si semantics (NaN / out of range -> INT_MIN): what the reference binary does
TERRA_HD int f2i_x86(float f) {return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;}

// ------------------------------------------------------------------ RNG (src/rand_gen.h:20-35,63-79; src/gen_object.cpp:377-381)
// L'Ecuyer combined LCG; the reference holds the state in `long`, values always fit in int32 after the first step.
struct rand_gen_t {
	int64_t rseed1, rseed2;
	TERRA_HD void set_state(int64_t s1, int64_t s2) {rseed1 = s1; rseed2 = s2;}
	TERRA_HD void advance() {
		if ((rseed1 = 40014*(rseed1%53668) - 12211*(rseed1/53668)) < 0) rseed1 += 2147483563;
		if ((rseed2 = 40692*(rseed2%52774) - 3791 *(rseed2/52774)) < 0) rseed2 += 2147483399;
	}
	TERRA_HD int rand() {
		advance();
		int v = (int)rseed1 - (int)rseed2;
		if (v < 1) v += 2147483562;
		return v;
	}
	TERRA_HD double randd() {
		advance();
		double v = (double)rseed1 - (double)rseed2;
		if (v < 1) v += 2147483562;
		return v/2147483563.;
	}
	TERRA_HD float rand_float() {return (float)(0.000001*(rand()%1000000));}                  // src/rand_gen.h:87
	TERRA_HD float rand_uniform(float a, float b) {return a + (b - a)*(float)randd();}        // src/rand_gen.h:91
};

// ------------------------------------------------------------------ SINF/COSF table lookup (src/sinf.h:8-21)
// The 2*TSIZE table itself is filled on the host with libm sinf/cosf exactly as create_sin_table() does
// (src/mesh_gen.cpp:72-81) and uploaded once per context.
struct sin_lut_t {
	float const *tab; // [2*TSIZE]: sin then cos
	float sscale;     // float(TSIZE)/TWO_PI
	// int(sscale*val) overflows for |val| > ~4e5 (estimate_zminmax samples at a 4000-cell spacing): the reference binary then gets
	// cvttss2si's 0x80000000 -> table index 0, so the x86 conversion is part of the function's definition
	TERRA_HD int st_scale(float v) const {return f2i_x86(sscale*v) & (TSIZE-1);}
	TERRA_HD float SINF(float v) const {return (v < 0) ? -tab[st_scale(-v)] : tab[st_scale(v)];}
	TERRA_HD float COSF(float v) const {return tab[TSIZE + st_scale(fabsf(v))];}
};

// hmap_params_t (src/mesh.h:84-88)
struct hmap_params_t {
	float plat_bot, plat_h, plat_s, plat_max, crat_h, crat_s, crack_lo, crack_hi, crack_d, sine_mag, sine_freq, sine_bias, volcano_width, volcano_height;
};

// Everything the per-cell evaluators read; mirrors the reference's process globals (SURVEY section 5 "Config / flags").
struct noise_consts_t {
	hmap_params_t hp;
	float mesh_scale, mesh_scale_z_inv, DX_VAL_INV, DY_VAL_INV, MESH_HEIGHT, mesh_height_scale;
	float zmax_est, zmax_est2, zmax_est2_inv, custom_glaciate_exp;
	float rx, ry;           // gen_rx_ry()
	int   start_eval_sin;   // compute_scale()
	int   glaciate;         // GLACIATE
};

} // namespace terra
