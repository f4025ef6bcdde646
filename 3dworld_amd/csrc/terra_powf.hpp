// terra_powf.hpp -- device/host powf with the EXACT results of glibc's libm (2.28+: sysdeps/ieee754/flt-32/e_powf.c,
// e_powf_log2_data.c, e_exp2f_data.c = ARM optimized-routines' powf: log2 by a 16-entry table + degree-5 polynomial in
// fp64, exp2 by a 32-entry table + cubic in fp64, result rounded to float).
//
// Why: glaciate with a custom exponent evaluates pow(relh, custom_glaciate_exp) per cell (src/mesh_gen.cpp:358-362) through libm's
// powf, which is not correctly rounded; the device's ocml powf differs from it by one ulp on ~13% of the cells of a test grid.
// The restatement below is evaluated with separate fp64 multiplies and adds and is checked against libm over 3*10^7 arguments on the
// build host (tests/test_oracle.py) and against the oracle on the GPU (tests/test_gpu_parity.py).
#pragma once
#include "terra_common.hpp"

namespace terra {

TERRA_HD uint32_t powf_asuint(float f) {uint32_t u; memcpy(&u, &f, 4); return u;}
TERRA_HD float powf_asfloat(uint32_t u) {float f; memcpy(&f, &u, 4); return f;}
TERRA_HD uint64_t powf_asuint64(double d) {uint64_t u; memcpy(&u, &d, 8); return u;}
TERRA_HD double powf_asdouble(uint64_t u) {double d; memcpy(&d, &u, 8); return d;}

// 0: y is not an integer, 1: odd integer, 2: even integer
TERRA_HD int powf_checkint(uint32_t iy) {
	int const e = (int)((iy >> 23) & 0xff);
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
TERRA_HD bool powf_zeroinfnan(uint32_t ix) {return 2*ix - 1 >= 2u*0x7f800000u - 1;}
TERRA_HD bool powf_issignaling(uint32_t ix) {return 2*(ix ^ 0x00400000u) > 2u*0x7fc00000u;}

TERRA_HD double powf_log2_inline(uint32_t ix) {
	// __powf_log2_data: {invc, logc} for 16 sub-intervals of [0x1.66p-1, 0x1.66p0), then the polynomial
	double const invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
		0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1, 0x1.b2036576afce6p-1,
		0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
	double const logc[16] = {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3,
		-0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5, 0x0p+0, 0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3, 0x1.e840b4ac4e4d2p-3,
		0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2, 0x1.ce0a44eb17bccp-2};
	double const A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
	uint32_t const tmp = ix - 0x3f330000u;
	int const i = (int)((tmp >> (23 - 4)) % 16u);
	uint32_t const top = tmp & 0xff800000u;
	uint32_t const iz = ix - top;
	int const k = (int32_t)top >> 23;
	double const z = (double)powf_asfloat(iz);
	double const r = z*invc[i] - 1.0;
	double const y0 = logc[i] + (double)k;
	double const r2 = r*r;
	double y = A0*r + A1;
	double const p = A2*r + A3;
	double const r4 = r2*r2;
	double q = A4*r + y0;
	q = p*r2 + q;
	y = y*r4 + q;
	return y;
}

TERRA_HD float powf_exp2_inline(double xd, uint32_t sign_bias) {
	uint64_t const T[32] = {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
		0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull,
		0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
		0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull,
		0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
	double const C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
	double const SHIFT = 0x1.8p+52/32.0;
	double kd = xd + SHIFT;
	uint64_t const ki = powf_asuint64(kd);
	kd -= SHIFT; // k/N
	double const r = xd - kd;
	uint64_t t = T[ki % 32u];
	uint64_t const ski = ki + sign_bias;
	t += ski << (52 - 5);
	double const s = powf_asdouble(t);
	double const z = C0*r + C1;
	double const r2 = r*r;
	double y = C2*r + 1.0;
	y = z*r2 + y;
	y = y*s;
	return (float)y;
}

TERRA_HD float glibc_powf(float x, float y) {
	uint32_t sign_bias = 0;
	uint32_t ix = powf_asuint(x);
	uint32_t const iy = powf_asuint(y);
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
		if (powf_zeroinfnan(iy)) {
			if (2*iy == 0) return powf_issignaling(ix) ? x + y : 1.0f;
			if (ix == 0x3f800000u) return powf_issignaling(iy) ? x + y : 1.0f;
			if (2*ix > 2u*0x7f800000u || 2*iy > 2u*0x7f800000u) return x + y;
			if (2*ix == 2*0x3f800000u) return 1.0f;
			if ((2*ix < 2*0x3f800000u) == !(iy & 0x80000000u)) return 0.0f; // |x| < 1 && y == inf or |x| > 1 && y == -inf
			return y*y;
		}
		if (powf_zeroinfnan(ix)) {
			float x2 = x*x;
			if ((ix & 0x80000000u) && powf_checkint(iy) == 1) {x2 = -x2;}
			return (iy & 0x80000000u) ? 1.0f/x2 : x2;
		}
		if (ix & 0x80000000u) { // x is negative, finite, non-zero
			int const yint = powf_checkint(iy);
			if (yint == 0) return (x - x)/(x - x); // invalid: NaN
			if (yint == 1) sign_bias = 1u << (5 + 11);
			ix &= 0x7fffffffu;
		}
		if (ix < 0x00800000u) { // subnormal x: normalise
			ix = powf_asuint(x*0x1p23f);
			ix &= 0x7fffffffu;
			ix -= 23u << 23;
		}
	}
	double const logx = powf_log2_inline(ix);
	double const ylogx = (double)y*logx; // cannot overflow, y is single precision
	if (((powf_asuint64(ylogx) >> 47) & 0xffff) >= (powf_asuint64(126.0) >> 47)) { // |y*log(x)| >= 126
		if (ylogx > 0x1.fffffffd1d571p+6) {float const big = sign_bias ? -0x1p97f : 0x1p97f; return big*0x1p97f;} // overflow: +-inf
		if (ylogx <= -150.0) {float const tiny = sign_bias ? -0x1p-95f : 0x1p-95f; return tiny*0x1p-95f;}     // underflow: +-0
	}
	return powf_exp2_inline(ylogx, sign_bias);
}

} // namespace terra
