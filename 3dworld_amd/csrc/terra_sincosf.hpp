// terra_sincosf.hpp -- device/host sinf/cosf with the EXACT results of glibc's libm (2.28+: sysdeps/ieee754/flt-32/
// s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c = ARM optimized-routines' sincosf, fp64 polynomial, result rounded to float).
//
// Why: the droplet's "pick a random direction" branch calls libm cosf(a)/sinf(a) (src/erosion.cpp:80-83).  glibc's
// sinf/cosf are NOT correctly rounded: on the 10^6 possible arguments a = rand_float()*TWO_PI they differ from
// (float)cos((double)a) in 2.6% of the cases, so neither the device's ocml sinf/cosf nor a correctly-rounded evaluation
// reproduces the reference there.  This restatement (evaluated in fp64 with separate multiply/add) matches libm on all
// 10^6 arguments (checked exhaustively on the build host; GPU check in tests/test_gpu_parity.py).
// Only the argument range the path can produce is covered: 0 <= y < 120.
#pragma once
#include "terra_common.hpp"

namespace terra {

struct sincosf_tab_t {double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;};

TERRA_HD float sincosf_poly(double x, double x2, bool neg_tab, int n) {
	// __sincosf_table[0] / [1]: the second table is the first with c0..c4 negated
	double const sg = neg_tab ? -1.0 : 1.0;
	double const c0 = sg*0x1p0, c1 = sg*-0x1.ffffffd0c621cp-2, c2 = sg*0x1.55553e1068f19p-5, c3 = sg*-0x1.6c087e89a359dp-10, c4 = sg*0x1.99343027bf8c3p-16;
	double const s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
	if ((n & 1) == 0) {
		double const x3 = x*x2, t1 = s2 + x2*s3, x5 = x3*x2, s = x + x3*s1;
		return (float)(s + x5*t1);
	}
	double const x4 = x2*x2, t2 = c3 + x2*c4, t1 = c0 + x2*c1, x6 = x4*x2, c = t1 + x4*c2;
	return (float)(c + x6*t2);
}
TERRA_HD uint32_t sincosf_abstop12(float x) {uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff;}

// is_cos = false: sinf(y), true: cosf(y); valid for 0 <= y < 120
TERRA_HD float glibc_sincosf(float y, bool is_cos) {
	double x = (double)y;
	if (sincosf_abstop12(y) < sincosf_abstop12(0x1.921FB6p-1f)) { // |y| < pi/4
		double const x2 = x*x;
		if (sincosf_abstop12(y) < sincosf_abstop12(0x1p-12f)) {return is_cos ? 1.0f : y;}
		return sincosf_poly(x, x2, false, is_cos ? 1 : 0);
	}
	// reduce_fast: n = round(x * 2/pi), x -= n*pi/2
	double const r = x*0x1.45F306DC9C883p+23;
	int const n = ((int32_t)r + 0x800000) >> 24;
	x = x - (double)n*0x1.921FB54442D18p0;
	double const sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; // sign[] = {1,-1,-1,1}
	return sincosf_poly(x*sign, x*x, (n & 2) != 0, is_cos ? (n ^ 1) : n);
}
TERRA_HD float glibc_sinf(float y) {return glibc_sincosf(y, false);}
TERRA_HD float glibc_cosf(float y) {return glibc_sincosf(y, true);}

} // namespace terra
