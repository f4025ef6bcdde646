// terra_noise.hpp -- per-cell evaluators of the heightmap generator (host+device).
//
// Replaces the CPU inner loops of mesh_xy_grid_cache_t::eval_index (src/mesh_gen.cpp:754-792):
//   * 2-D / 3-D simplex and classic Perlin noise as vendored by the reference
//     (dependencies/glm/glm/gtc/noise.inl, GLM 0.9.9.1) -- same operation order, fp32, no FMA contraction;
//   * fBm accumulation and domain warp (src/mesh_gen.cpp:706-751);
//   * noise shaping / plateau-crater-crack post-process (src/mesh_gen.cpp:555-571);
//   * glaciate cubic remap, sine-mag islands, volcano (src/mesh_gen.cpp:358-385,782-790).
// One thread evaluates one cell; there is no cross-lane traffic in these functions.
#pragma once
#include "terra_common.hpp"
#include "terra_powf.hpp"

namespace terra {

// ---- two cells per lane.  The 2-D lattice noise below is written once, as templates over the arithmetic type T: float (one cell; what the
// host emulator, the 3-D noise and the point queries use) or nv2 = two floats in a 64-bit register pair (two neighbouring cells of a row).
// With nv2 every multiply / add / subtract becomes one v_pk_mul_f32 / v_pk_add_f32 for both cells -- the same IEEE operations in the same
// order, so the results are bit-identical to the one-cell instantiation (tests/emul: terra_emul_noise_x2_mismatches; GPU parity tests).
typedef float nv2 __attribute__((vector_size(8)));
typedef int   ni2 __attribute__((vector_size(8)));
template<class T> struct nt_traits;
template<> struct nt_traits<float> {typedef bool mask_t;};
template<> struct nt_traits<nv2>   {typedef ni2  mask_t;};
TERRA_HD float nt_floor(float a) {return floorf(a);}
TERRA_HD nv2   nt_floor(nv2 a)   {return nv2{floorf(a[0]), floorf(a[1])};}
TERRA_HD float nt_abs(float a) {return fabsf(a);}
TERRA_HD nv2   nt_abs(nv2 a)   {return nv2{fabsf(a[0]), fabsf(a[1])};}
TERRA_HD float nt_sel(bool m, float a, float b) {return m ? a : b;}
TERRA_HD nv2   nt_sel(ni2 m, nv2 a, nv2 b) {return nv2{m[0] ? a[0] : b[0], m[1] ? a[1] : b[1]};}
TERRA_HD float nt_max_std(float a, float b) {return (a < b) ? b : a;} // std::max
TERRA_HD nv2   nt_max_std(nv2 a, nv2 b) {return nt_sel(a < b, b, a);}
TERRA_HD float nt_fma(float a, float b, float c) {return fmaf(a, b, c);}
TERRA_HD nv2   nt_fma(nv2 a, nv2 b, nv2 c) {return nv2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};}
template<class T> TERRA_HD T nt_bc(float v);
template<> TERRA_HD float nt_bc<float>(float v) {return v;}
template<> TERRA_HD nv2   nt_bc<nv2>(float v)   {return nv2{v, v};}
TERRA_HD bool nt_all_below(float a, float lim) {return fabsf(a) < lim;}
TERRA_HD bool nt_all_below(nv2 a, float lim) {return fabsf(a[0]) < lim && fabsf(a[1]) < lim;}

// ---- GLM scalar helpers (detail/_noise.hpp:14-84, detail/func_common.inl:123-131,211-218,386-398,543-546)
template<class T> TERRA_HD T gl_mod289(T x)  {return x - nt_floor(x*(1.0f/289.0f))*289.0f;}
template<class T> TERRA_HD T gl_permute(T x) {return gl_mod289(((x*34.0f) + 1.0f)*x);}
TERRA_HD float gl_tinvsqrt(float r) {return 1.79284291400159f - 0.85373472095314f*r;}
template<class T> TERRA_HD T gl_fade(T t)    {return (t*t*t)*(t*(t*6.0f - 15.0f) + 10.0f);}
TERRA_HD float gl_mod(float a, float b) {return a - b*floorf(a/b);}
// glm::mod(a, 289.0f) for the integer-valued arguments the lattice code passes (floor() results): for |a| < 2^23 every step below is exact,
// so the IEEE division of a - b*floor(a/b) can be replaced by a reciprocal estimate of the quotient and one range correction; anything
// else (huge, inf, NaN) takes the division.  Checked against the division on 5.2e6 integers incl. negatives and on the GPU by the parity tests.
template<class T> TERRA_HD T gl_mod289_int_fast(T a) {
	T const q = nt_floor(a*(1.0f/289.0f));
	T r = a - q*289.0f;
	r = nt_sel(r < 0.0f, r + 289.0f, r);
	r = nt_sel(r >= 289.0f, r - 289.0f, r);
	return r;
}
TERRA_HD float gl_mod289_int(float a) {return nt_all_below(a, 8388608.0f) ? gl_mod289_int_fast(a) : gl_mod(a, 289.0f);}
TERRA_HD nv2   gl_mod289_int(nv2 a)   {return nt_all_below(a, 8388608.0f) ? gl_mod289_int_fast(a) : nv2{gl_mod289_int(a[0]), gl_mod289_int(a[1])};}
// h/41.0f for h = permute(...) in {0, ..., 288}: the correctly rounded quotient by one FMA correction of h*RN(1/41) (every integer in
// [-400, 700] checked: bit-identical to the IEEE division; a plain multiplication differs on 132 of the 289 values)
template<class T> TERRA_HD T gl_div41(T h) {
	T const r = nt_bc<T>(1.0f/41.0f), q0 = h*r;
	return nt_fma(nt_fma(-q0, nt_bc<T>(41.0f), h), r, q0);
}
template<class T> TERRA_HD T gl_fract(T x)   {return x - nt_floor(x);}
template<class T> TERRA_HD T gl_mix(T x, T y, T a) {return x + a*(y - x);}
TERRA_HD float gl_step(float edge, float x) {return (x < edge) ? 0.0f : 1.0f;}

// glm::simplex(vec2)  (gtc/noise.inl:592-646)
template<class T> TERRA_HD T simplex2_t(T vx, T vy) {
	float const C0 = 0.211324865405187f, C1 = 0.366025403784439f, C2 = -0.577350269189626f, C3 = 0.024390243902439f;
	T const one = nt_bc<T>(1.0f), zero = nt_bc<T>(0.0f);
	T const skew = vx*C1 + vy*C1;                           // dot(v, C.yy)
	T cx = nt_floor(vx + skew), cy = nt_floor(vy + skew);   // first corner (skewed cell)
	T const unskew = cx*C0 + cy*C0;                         // dot(i, C.xx)
	T const ax = vx - cx + unskew, ay = vy - cy + unskew;   // x0
	typename nt_traits<T>::mask_t const lower = (ax > ay);
	T const ox = nt_sel(lower, one, zero), oy = nt_sel(lower, zero, one); // i1
	T const bx = (ax + C0) - ox, by = (ay + C0) - oy;       // x12.xy
	T const ex = ax + C2, ey = ay + C2;                     // x12.zw
	cx = gl_mod289_int(cx); cy = gl_mod289_int(cy);
	T const pa = gl_permute(gl_permute(cy + 0.0f) + cx + 0.0f);
	T const pb = gl_permute(gl_permute(cy + oy  ) + cx + ox  );
	T const pc = gl_permute(gl_permute(cy + 1.0f) + cx + 1.0f);
	T ma = nt_max_std(0.5f - (ax*ax + ay*ay), zero);
	T mb = nt_max_std(0.5f - (bx*bx + by*by), zero);
	T mc = nt_max_std(0.5f - (ex*ex + ey*ey), zero);
	ma = ma*ma; mb = mb*mb; mc = mc*mc;
	ma = ma*ma; mb = mb*mb; mc = mc*mc;
	// gradients: 41 points on a line mapped onto a diamond
	T const ga = 2.0f*gl_fract(pa*C3) - 1.0f, gb = 2.0f*gl_fract(pb*C3) - 1.0f, gc = 2.0f*gl_fract(pc*C3) - 1.0f;
	T const ha = nt_abs(ga) - 0.5f, hb = nt_abs(gb) - 0.5f, hc = nt_abs(gc) - 0.5f;
	T const a0a = ga - nt_floor(ga + 0.5f), a0b = gb - nt_floor(gb + 0.5f), a0c = gc - nt_floor(gc + 0.5f);
	ma *= 1.79284291400159f - 0.85373472095314f*(a0a*a0a + ha*ha);
	mb *= 1.79284291400159f - 0.85373472095314f*(a0b*a0b + hb*hb);
	mc *= 1.79284291400159f - 0.85373472095314f*(a0c*a0c + hc*hc);
	T const da = a0a*ax + ha*ay;
	T const db = a0b*bx + hb*by;
	T const dc = a0c*ex + hc*ey;
	return 130.0f*(ma*da + mb*db + mc*dc);
}
TERRA_HD float simplex2(float vx, float vy) {return simplex2_t<float>(vx, vy);}

// glm::perlin(vec2)  (gtc/noise.inl:25-62)
template<class T> TERRA_HD T perlin2_t(T px, T py) {
	T const flx = nt_floor(px), fly = nt_floor(py);
	T const frx = px - flx, fry = py - fly;                    // fract
	T const cx0 = gl_mod289_int(flx + 0.0f), cy0 = gl_mod289_int(fly + 0.0f);
	T const cx1 = gl_mod289_int(flx + 1.0f), cy1 = gl_mod289_int(fly + 1.0f);
	T const fx0 = frx - 0.0f, fy0 = fry - 0.0f, fx1 = frx - 1.0f, fy1 = fry - 1.0f;
	// corner order of the vec4 lanes: (x0,y0) (x1,y0) (x0,y1) (x1,y1)
	T gx[4], gy[4];
	T const cxs[4] = {cx0, cx1, cx0, cx1}, cys[4] = {cy0, cy0, cy1, cy1};
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		T const h = gl_permute(gl_permute(cxs[c]) + cys[c]);
		T const g = 2.0f*gl_fract(gl_div41(h)) - 1.0f; // h/41.0f
		gy[c] = nt_abs(g) - 0.5f;
		gx[c] = g - nt_floor(g + 0.5f);
	}
	// norm = taylorInvSqrt(dot(g00), dot(g01), dot(g10), dot(g11)); lanes: g00=0 g10=1 g01=2 g11=3
	T const n00 = 1.79284291400159f - 0.85373472095314f*(gx[0]*gx[0] + gy[0]*gy[0]), n01 = 1.79284291400159f - 0.85373472095314f*(gx[2]*gx[2] + gy[2]*gy[2]);
	T const n10 = 1.79284291400159f - 0.85373472095314f*(gx[1]*gx[1] + gy[1]*gy[1]), n11 = 1.79284291400159f - 0.85373472095314f*(gx[3]*gx[3] + gy[3]*gy[3]);
	T const d00 = (gx[0]*n00)*fx0 + (gy[0]*n00)*fy0;
	T const d10 = (gx[1]*n10)*fx1 + (gy[1]*n10)*fy0;
	T const d01 = (gx[2]*n01)*fx0 + (gy[2]*n01)*fy1;
	T const d11 = (gx[3]*n11)*fx1 + (gy[3]*n11)*fy1;
	T const ux = gl_fade(fx0), uy = gl_fade(fy0);
	T const lo = gl_mix(d00, d10, ux), hi = gl_mix(d01, d11, ux);
	return 2.3f*gl_mix(lo, hi, uy);
}
TERRA_HD float perlin2(float px, float py) {return perlin2_t<float>(px, py);}

// glm::perlin(vec3)  (gtc/noise.inl:66-133)
TERRA_HD float perlin3(float px, float py, float pz) {
	float const p[3] = {px, py, pz};
	float c0[3], c1[3], f0[3], f1[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		float const fl = floorf(p[d]);
		c0[d] = gl_mod289(fl); c1[d] = gl_mod289(fl + 1.0f);
		f0[d] = p[d] - fl; f1[d] = f0[d] - 1.0f;
	}
	float const seventh = (float)(1.0/7.0);
	float nlo[4], nhi[4]; // dot products at z0 / z1 for lanes (x0,y0) (x1,y0) (x0,y1) (x1,y1)
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		float const hx = (c & 1) ? c1[0] : c0[0], hy = (c & 2) ? c1[1] : c0[1];
		float const hxy = gl_permute(gl_permute(hx) + hy);
		float const fxc = (c & 1) ? f1[0] : f0[0], fyc = (c & 2) ? f1[1] : f0[1];
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			float const hz = gl_permute(hxy + (s ? c1[2] : c0[2]));
			float gx = hz*seventh;
			float gy = gl_fract(floorf(gx)*seventh) - 0.5f;
			gx = gl_fract(gx);
			float const gz = 0.5f - fabsf(gx) - fabsf(gy);
			float const sz = gl_step(gz, 0.0f);
			gx -= sz*(gl_step(0.0f, gx) - 0.5f);
			gy -= sz*(gl_step(0.0f, gy) - 0.5f);
			float const nrm = gl_tinvsqrt(gx*gx + gy*gy + gz*gz);
			float const d = (gx*nrm)*fxc + (gy*nrm)*fyc + (gz*nrm)*(s ? f1[2] : f0[2]);
			if (s) {nhi[c] = d;} else {nlo[c] = d;}
		}
	}
	float const ux = gl_fade(f0[0]), uy = gl_fade(f0[1]), uz = gl_fade(f0[2]);
	float const z0 = gl_mix(nlo[0], nhi[0], uz), z1 = gl_mix(nlo[1], nhi[1], uz), z2 = gl_mix(nlo[2], nhi[2], uz), z3 = gl_mix(nlo[3], nhi[3], uz);
	float const y0 = gl_mix(z0, z2, uy), y1 = gl_mix(z1, z3, uy);
	return 2.2f*gl_mix(y0, y1, ux);
}

// glm::simplex(vec3)  (gtc/noise.inl:649-722)
TERRA_HD float simplex3(float vx, float vy, float vz) {
	float const G = (float)(1.0/6.0), F = (float)(1.0/3.0);
	float const skew = vx*F + vy*F + vz*F;
	float ci[3] = {floorf(vx + skew), floorf(vy + skew), floorf(vz + skew)};
	float const unskew = ci[0]*G + ci[1]*G + ci[2]*G;
	float const a[3] = {vx - ci[0] + unskew, vy - ci[1] + unskew, vz - ci[2] + unskew}; // x0
	float const g[3] = {gl_step(a[1], a[0]), gl_step(a[2], a[1]), gl_step(a[0], a[2])};  // step(x0.yzx, x0)
	float const l[3] = {1.0f - g[0], 1.0f - g[1], 1.0f - g[2]};
	float const o1[3] = {min_std(g[0], l[2]), min_std(g[1], l[0]), min_std(g[2], l[1])}; // i1 = min(g, l.zxy)
	float const o2[3] = {max_std(g[0], l[2]), max_std(g[1], l[0]), max_std(g[2], l[1])}; // i2 = max(g, l.zxy)
	float b[3], c[3], e[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {b[d] = a[d] - o1[d] + G; c[d] = a[d] - o2[d] + F; e[d] = a[d] - 0.5f;}
#pragma unroll
	for (int d = 0; d < 3; ++d) {ci[d] = gl_mod289(ci[d]);}
	float const n_ = 0.142857142857f;
	float const ns0 = n_*2.0f - 0.0f, ns1 = n_*0.5f - 1.0f, ns2 = n_*1.0f - 0.0f;
	float const offz[4] = {0.0f, o1[2], o2[2], 1.0f}, offy[4] = {0.0f, o1[1], o2[1], 1.0f}, offx[4] = {0.0f, o1[0], o2[0], 1.0f};
	float qx[4], qy[4], qh[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		float const pm = gl_permute(gl_permute(gl_permute(ci[2] + offz[k]) + ci[1] + offy[k]) + ci[0] + offx[k]);
		float const j  = pm - 49.0f*floorf(pm*ns2*ns2);
		float const x_ = floorf(j*ns2);
		float const y_ = floorf(j - 7.0f*x_);
		qx[k] = x_*ns0 + ns1;
		qy[k] = y_*ns0 + ns1;
		qh[k] = 1.0f - fabsf(qx[k]) - fabsf(qy[k]);
	}
	// b0 = (x.xy, y.xy), b1 = (x.zw, y.zw); s = floor(b)*2+1; sh = -step(h,0); a0 = b0.xzyw + s0.xzyw*sh.xxyy; a1 = b1.xzyw + s1.xzyw*sh.zzww
	float gx[4], gy[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		float const sh = -gl_step(qh[k], 0.0f);
		gx[k] = qx[k] + (floorf(qx[k])*2.0f + 1.0f)*sh;
		gy[k] = qy[k] + (floorf(qy[k])*2.0f + 1.0f)*sh;
	}
	float const *xs[4] = {a, b, c, e};
	float w[4], dp[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		float const nrm = gl_tinvsqrt(gx[k]*gx[k] + gy[k]*gy[k] + qh[k]*qh[k]);
		float const *xk = xs[k];
		dp[k] = (gx[k]*nrm)*xk[0] + (gy[k]*nrm)*xk[1] + (qh[k]*nrm)*xk[2];
		float m = max_std(0.6f - (xk[0]*xk[0] + xk[1]*xk[1] + xk[2]*xk[2]), 0.0f);
		m = m*m;
		w[k] = m*m;
	}
	return 42.0f*((w[0]*dp[0] + w[1]*dp[1]) + (w[2]*dp[2] + w[3]*dp[3]));
}

// ---- lattice tables.  Everything glm's 2-D noises compute from the hashed lattice point alone is a function of one small integer:
//   simplex: p = permute(permute(cy + oy) + cx + ox) with cx, cy in {0..288} after mod 289 and ox, oy in {0, 1}: the inner permute has 290 possible
//            arguments, the outer one 579 (0 .. 289 + 288 + 1), and the gradient terms {a0, h, 1.79 - 0.85*(a0^2 + h^2)} depend on p only;
//   Perlin:  h = permute(permute(cx) + cy), cx / cy in {0..288}: 289 inner arguments, 578 outer ones, {gx*norm, gy*norm} depend on h only.
// The tables hold exactly what the per-cell code computes for each argument (filled by that code, noise_lut_fill), so a look-up returns the same
// bits -- including the cases where permute() leaves 289 instead of 0, which is why simplex's inner table is NOT wrapped (cy + 1 = 289 is a
// different argument from 0) while Perlin's is indexed by mod289(flx + 1).  Per lattice point ~25 instructions become a few integer adds + LDS reads.
// The look-up index is the residue r = a - floor(a*(1/289))*289 WITHOUT range fix-up where the table can absorb it: for an integer |a| < 2^22 the
// quotient estimate is exact unless a is a multiple of 289, where it may come out one too small (RN(1/289) > 1/289 makes that the negative multiples), so
// r is the true residue or 289 standing for 0, never negative (every integer of the range is checked by tests/emul).  Entries [289] repeat entries [0].
// Layout (dwords; all stored offsets are BYTE offsets from the start of the part, so a look-up is base + offset + (residue << 2)):
//   simplex part  S_I: 290 x {A, A + 4, C, 0},  A = &a0[permute(r' + 0)], C = &a0[permute(r' + 1)], r' = r mod 289
//                 S_G: three arrays a0[579], h[579], norm[579] -- structure of arrays on purpose: the two cells of a lane fetch the same field with one
//                      4-byte read each, into a register PAIR that the packed arithmetic uses as is (an array of {a0, h, norm} records is read with wide
//                      loads per cell and then needs a register move per value to build the pairs); the array stride (2316 bytes) is neither below
//                      1 KiB nor a multiple of 256, so the compiler cannot fuse reads of one cell into ds_read2
//   Perlin part   P_I: 292 x (&gxn[permute(v mod 289)]),  P_G: gxn[580], gyn[580]
// A kernel stages the part it needs in LDS.
constexpr unsigned NOISE_LUT_S_G = 290*4, NOISE_LUT_S_N = 579, NOISE_LUT_S_DWORDS = ((NOISE_LUT_S_G + 3*NOISE_LUT_S_N + 3)/4)*4;
constexpr unsigned NOISE_LUT_P_G = 292, NOISE_LUT_P_N = 580, NOISE_LUT_P_DWORDS = ((NOISE_LUT_P_G + 2*NOISE_LUT_P_N + 3)/4)*4;
constexpr unsigned NOISE_LUT_DWORDS = NOISE_LUT_S_DWORDS + NOISE_LUT_P_DWORDS; // simplex part first, then the Perlin part
TERRA_HD uint32_t nt_bits(float f) {uint32_t u; memcpy(&u, &f, 4); return u;}
TERRA_HD uint32_t noise_lut_fill(unsigned i) { // dword i of the table
	if (i < NOISE_LUT_S_DWORDS) {
		if (i < NOISE_LUT_S_G) {
			unsigned const r = (i >> 2) % 289u, c = i & 3u;
			uint32_t const A = NOISE_LUT_S_G*4 + ((uint32_t)(int)gl_permute<float>((float)r + 0.0f) << 2), C = NOISE_LUT_S_G*4 + ((uint32_t)(int)gl_permute<float>((float)r + 1.0f) << 2);
			return (c == 0) ? A : ((c == 1) ? A + 4 : ((c == 2) ? C : 0u));
		}
		unsigned const k = i - NOISE_LUT_S_G, c = k / NOISE_LUT_S_N, v = k % NOISE_LUT_S_N;
		if (c >= 3) return 0u;
		float const C3 = 0.024390243902439f;
		float const p = gl_permute<float>((float)v);
		float const g = 2.0f*gl_fract(p*C3) - 1.0f, h = nt_abs(g) - 0.5f, a0 = g - nt_floor(g + 0.5f);
		float const n = 1.79284291400159f - 0.85373472095314f*(a0*a0 + h*h);
		return (c == 0) ? nt_bits(a0) : ((c == 1) ? nt_bits(h) : nt_bits(n));
	}
	i -= NOISE_LUT_S_DWORDS;
	if (i < NOISE_LUT_P_G) {return NOISE_LUT_P_G*4 + ((uint32_t)(int)gl_permute<float>((float)(i % 289u)) << 2);} // [r] -> permute(mod289(fl)), [r + 1] -> permute(mod289(fl + 1)), r in 0..289
	unsigned const k = i - NOISE_LUT_P_G, c = k / NOISE_LUT_P_N, v = k % NOISE_LUT_P_N;
	if (c >= 2) return 0u;
	float const h = gl_permute<float>((float)v);
	float const g = 2.0f*gl_fract(gl_div41(h)) - 1.0f, gy = nt_abs(g) - 0.5f, gx = g - nt_floor(g + 0.5f);
	float const n = 1.79284291400159f - 0.85373472095314f*(gx*gx + gy*gy);
	return nt_bits(c ? gy*n : gx*n);
}
// residue of an integer |a| < 2^22 in {0, ..., 288} or 289 (= 0) -- see the note above; _small: the same folded into 0 .. 288
template<class T> TERRA_HD T gl_mod289_raw(T a) {return a - nt_floor(a*(1.0f/289.0f))*289.0f;}
template<class T> TERRA_HD T gl_mod289_small(T a) {T const r = gl_mod289_raw(a); return nt_sel(r >= 289.0f, r - 289.0f, r);}
TERRA_HD float nt_max0(float a) {return fmaxf(a, 0.0f);}                 // std::max(a, 0.0f) for a that is not a NaN: one v_max_f32 instead of compare + select
TERRA_HD nv2   nt_max0(nv2 a)   {return nv2{fmaxf(a[0], 0.0f), fmaxf(a[1], 0.0f)};}
struct alignas(16) nt_i4 {int x, y, z, w;};
TERRA_HD float nt_ldf(char const *p) {return *(float const *)p;}
// glm::simplex(vec2) for two cells, lattice part from the table (`tab` = start of the simplex part, 16-byte aligned).  CHECK: fall back to simplex2_t
// when a lattice coordinate is not an exactly representable small integer (|c| >= 2^22, inf, NaN); without it the caller guarantees that it is.
template<bool CHECK> TERRA_HD nv2 simplex2_lut(nv2 vx, nv2 vy, char const *tab) {
	float const C0 = 0.211324865405187f, C1 = 0.366025403784439f, C2 = -0.577350269189626f;
	nv2 const one = nt_bc<nv2>(1.0f), zero = nt_bc<nv2>(0.0f);
	nv2 const skew = vx*C1 + vy*C1;
	nv2 const cx = nt_floor(vx + skew), cy = nt_floor(vy + skew);
	if (CHECK && TERRA_UNLIKELY(!(nt_all_below(cx, 4194304.0f) && nt_all_below(cy, 4194304.0f)))) {return simplex2_t<nv2>(vx, vy);}
	// from here on everything is finite (cx, cy small => vx, vy small), so std::max(m, 0) is fmaxf(m, 0)
	nv2 const unskew = cx*C0 + cy*C0;
	nv2 const ax = vx - cx + unskew, ay = vy - cy + unskew;
	ni2 const lower = (ax > ay);
	nv2 const ox = nt_sel(lower, one, zero), oy = one - ox; // (1, 0) or (0, 1)
	nv2 const bx = (ax + C0) - ox, by = (ay + C0) - oy;
	nv2 const ex = ax + C2, ey = ay + C2;
	nv2 const mx4 = gl_mod289_small(cx)*4.0f, ry16 = gl_mod289_raw(cy)*16.0f; // byte offsets of the residues: exact small integers
	nv2 ma = nt_max0(0.5f - (ax*ax + ay*ay));
	nv2 mb = nt_max0(0.5f - (bx*bx + by*by));
	nv2 mc = nt_max0(0.5f - (ex*ex + ey*ey));
	ma = ma*ma; mb = mb*mb; mc = mc*mc;
	ma = ma*ma; mb = mb*mb; mc = mc*mc;
	char const *pa[2], *pb[2], *pc[2];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		nt_i4 const si = *(nt_i4 const *)(tab + (int)ry16[e]);   // {&a0[permute(cy + 0)], the same + 4, &a0[permute(cy + 1)]}
		char const *g = tab + (int)mx4[e];
		pa[e] = g + si.x;                                        // permute(cy + 0) + cx + 0
		pb[e] = g + (lower[e] ? si.y : si.z);                    // (ox, oy) = (1, 0): permute(cy) + cx + 1, or (0, 1): permute(cy + 1) + cx
		pc[e] = g + si.z + 4;                                    // permute(cy + 1) + cx + 1
	}
	constexpr int SH = NOISE_LUT_S_N*4, SN = 2*NOISE_LUT_S_N*4;
	nv2 const a0a = {nt_ldf(pa[0]), nt_ldf(pa[1])}, ha = {nt_ldf(pa[0] + SH), nt_ldf(pa[1] + SH)}, na = {nt_ldf(pa[0] + SN), nt_ldf(pa[1] + SN)};
	nv2 const a0b = {nt_ldf(pb[0]), nt_ldf(pb[1])}, hb = {nt_ldf(pb[0] + SH), nt_ldf(pb[1] + SH)}, nb = {nt_ldf(pb[0] + SN), nt_ldf(pb[1] + SN)};
	nv2 const a0c = {nt_ldf(pc[0]), nt_ldf(pc[1])}, hc = {nt_ldf(pc[0] + SH), nt_ldf(pc[1] + SH)}, ncc = {nt_ldf(pc[0] + SN), nt_ldf(pc[1] + SN)};
	ma *= na; mb *= nb; mc *= ncc;
	nv2 const da = a0a*ax + ha*ay;
	nv2 const db = a0b*bx + hb*by;
	nv2 const dc = a0c*ex + hc*ey;
	return 130.0f*(ma*da + mb*db + mc*dc);
}
// glm::perlin(vec2) for two cells (`tab` = start of the Perlin part)
template<bool CHECK> TERRA_HD nv2 perlin2_lut(nv2 px, nv2 py, char const *tab) {
	nv2 const flx = nt_floor(px), fly = nt_floor(py);
	if (CHECK && TERRA_UNLIKELY(!(nt_all_below(flx, 4194304.0f) && nt_all_below(fly, 4194304.0f)))) {return perlin2_t<nv2>(px, py);}
	nv2 const frx = px - flx, fry = py - fly;
	nv2 const rx4 = gl_mod289_raw(flx)*4.0f, my = gl_mod289_small(fly); // = mod289(fl + 0) (x: 289 stands for 0, the table is indexed that way); mod289(fl + 1) is the next residue
	nv2 const fx0 = frx - 0.0f, fy0 = fry - 0.0f, fx1 = frx - 1.0f, fy1 = fry - 1.0f;
	char const *p00[2], *p10[2], *p01[2], *p11[2];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		int const *pip = (int const *)(tab + (int)rx4[e]);
		struct {int x, y;} const pi = {pip[0], pip[1]};          // &gxn[permute(cx0)], &gxn[permute(cx1)] (the table wraps 289 -> 0)
		int const iy0 = (int)my[e];
		int iy1 = iy0 + 1; iy1 = (iy1 == 289) ? 0 : iy1;
		char const *g0 = tab + (iy0 << 2), *g1 = tab + (iy1 << 2);
		p00[e] = g0 + pi.x; p10[e] = g0 + pi.y; p01[e] = g1 + pi.x; p11[e] = g1 + pi.y;
	}
	constexpr int SY = NOISE_LUT_P_N*4;
	nv2 const gx00 = {nt_ldf(p00[0]), nt_ldf(p00[1])}, gy00 = {nt_ldf(p00[0] + SY), nt_ldf(p00[1] + SY)}, gx10 = {nt_ldf(p10[0]), nt_ldf(p10[1])}, gy10 = {nt_ldf(p10[0] + SY), nt_ldf(p10[1] + SY)};
	nv2 const gx01 = {nt_ldf(p01[0]), nt_ldf(p01[1])}, gy01 = {nt_ldf(p01[0] + SY), nt_ldf(p01[1] + SY)}, gx11 = {nt_ldf(p11[0]), nt_ldf(p11[1])}, gy11 = {nt_ldf(p11[0] + SY), nt_ldf(p11[1] + SY)};
	nv2 const d00 = gx00*fx0 + gy00*fy0;
	nv2 const d10 = gx10*fx1 + gy10*fy0;
	nv2 const d01 = gx01*fx0 + gy01*fy1;
	nv2 const d11 = gx11*fx1 + gy11*fy1;
	nv2 const ux = gl_fade(fx0), uy = gl_fade(fy0);
	nv2 const lo = gl_mix(d00, d10, ux), hi = gl_mix(d01, d11, ux);
	return 2.3f*gl_mix(lo, hi, uy);
}
// evaluator for fbm2_t / noise_zval_t: stab / ptab = the simplex / Perlin part of the table (only the one the mode uses needs to be valid)
struct noise_tab_t {
	char const *stab, *ptab;
	TERRA_HD nv2 simplex(nv2 x, nv2 y) const {return simplex2_lut<true>(x, y, stab);}
	TERRA_HD nv2 perlin(nv2 x, nv2 y) const {return perlin2_lut<true>(x, y, ptab);}
};
struct noise_tab_nocheck_t { // for sample positions the caller has bounded (fbm2_tab)
	char const *stab, *ptab;
	TERRA_HD nv2 simplex(nv2 x, nv2 y) const {return simplex2_lut<false>(x, y, stab);}
	TERRA_HD nv2 perlin(nv2 x, nv2 y) const {return perlin2_lut<false>(x, y, ptab);}
};

// ---- 3-D lattice tables (the voxel fields of voxel_manager::create_procedural, src/voxels.cpp:312-345).  The same observation as in 2-D: glm's perlin(vec3) / simplex(vec3)
// reach the lattice through permute() of small integers only -- permute(permute(permute(a) + b) + c) with every argument an exact integer in 0 .. 580 -- and everything they
// compute from the hashed point (the gradient and its taylorInvSqrt norm, ~45 of the ~55 instructions per corner) depends on that integer alone.  One table of permute()
// stored as BYTE offsets (value*4: the next look-up's index, or the gradient's) and three arrays {gx*norm, gy*norm, gz*norm}, all filled by the per-voxel code itself
// (noise3_lut_fill): 5.8 KB of LDS.  Two voxels per lane (z, z + 1): Perlin's x / y lattice work is shared by the pair, the rest runs on register pairs.
//   layout (dwords): PERMB[584] | GX[292] GY[292] GZ[292]
constexpr unsigned NOISE3_PERM_N = 584, NOISE3_G_N = 292, NOISE3_LUT_DWORDS = NOISE3_PERM_N + 3*NOISE3_G_N; // 1460 = 365 x 4
// the gradient of perlin(vec3) at a corner whose hash is hz, times its norm (gtc/noise.inl:91-118 for one lane of the vec4s)
TERRA_HD void perlin3_gradn(float hz, float g[3]) {
	float const seventh = (float)(1.0/7.0);
	float gx = hz*seventh;
	float gy = gl_fract(floorf(gx)*seventh) - 0.5f;
	gx = gl_fract(gx);
	float const gz = 0.5f - fabsf(gx) - fabsf(gy);
	float const sz = gl_step(gz, 0.0f);
	gx -= sz*(gl_step(0.0f, gx) - 0.5f);
	gy -= sz*(gl_step(0.0f, gy) - 0.5f);
	float const nrm = gl_tinvsqrt(gx*gx + gy*gy + gz*gz);
	g[0] = gx*nrm; g[1] = gy*nrm; g[2] = gz*nrm;
}
// the gradient of simplex(vec3) at a corner whose hash is pm, times its norm (gtc/noise.inl:675-708)
TERRA_HD void simplex3_gradn(float pm, float g[3]) {
	float const n_ = 0.142857142857f;
	float const ns0 = n_*2.0f - 0.0f, ns1 = n_*0.5f - 1.0f, ns2 = n_*1.0f - 0.0f;
	float const j  = pm - 49.0f*floorf(pm*ns2*ns2);
	float const x_ = floorf(j*ns2);
	float const y_ = floorf(j - 7.0f*x_);
	float const qx = x_*ns0 + ns1, qy = y_*ns0 + ns1;
	float const qh = 1.0f - fabsf(qx) - fabsf(qy);
	float const sh = -gl_step(qh, 0.0f);
	float const gx = qx + (floorf(qx)*2.0f + 1.0f)*sh, gy = qy + (floorf(qy)*2.0f + 1.0f)*sh;
	float const nrm = gl_tinvsqrt(gx*gx + gy*gy + qh*qh);
	g[0] = gx*nrm; g[1] = gy*nrm; g[2] = qh*nrm;
}
TERRA_HD uint32_t noise3_lut_fill(unsigned i, bool perlin) { // dword i of the table
	if (i < NOISE3_PERM_N) {return (uint32_t)(int)gl_permute<float>((float)i) << 2;}
	unsigned const k = i - NOISE3_PERM_N, c = k / NOISE3_G_N, v = k % NOISE3_G_N;
	float g[3];
	if (perlin) {perlin3_gradn((float)v, g);} else {simplex3_gradn((float)v, g);}
	return nt_bits(g[c]);
}
TERRA_HD int nt_ldi(char const *p) {return *(int const *)p;}
constexpr int NOISE3_GX = NOISE3_PERM_N*4, NOISE3_GS = NOISE3_G_N*4; // byte offset of GX, byte stride between the gradient arrays
// glm::perlin(vec3) at (px, py, pz[0]) and (px, py, pz[1]), lattice part from the table.  CHECK: the direct code when a lattice coordinate is not a small integer
template<bool CHECK> TERRA_HD nv2 perlin3_lut_z2(float px, float py, nv2 pz, char const *tab) {
	float const flx = floorf(px), fly = floorf(py);
	nv2 const flz = nt_floor(pz);
	if (CHECK && TERRA_UNLIKELY(!(fabsf(flx) < 4194304.0f && fabsf(fly) < 4194304.0f && nt_all_below(flz, 4194304.0f)))) {return nv2{perlin3(px, py, pz[0]), perlin3(px, py, pz[1])};}
	float const f0x = px - flx, f0y = py - fly, f1x = f0x - 1.0f, f1y = f0y - 1.0f;
	nv2 const f0z = pz - flz, f1z = f0z - 1.0f;
	// residues (0 .. 289, exact integers) as byte offsets into PERMB
	int const ix0 = (int)gl_mod289<float>(flx) << 2, ix1 = (int)gl_mod289<float>(flx + 1.0f) << 2, iy0 = (int)gl_mod289<float>(fly) << 2, iy1 = (int)gl_mod289<float>(fly + 1.0f) << 2;
	nv2 const cz0 = gl_mod289<nv2>(flz), cz1 = gl_mod289<nv2>(flz + 1.0f);
	char const *const tx0 = tab + nt_ldi(tab + ix0), *const tx1 = tab + nt_ldi(tab + ix1);                 // permute(hx) + ...
	char const *const t[4] = {tab + nt_ldi(tx0 + iy0), tab + nt_ldi(tx1 + iy0), tab + nt_ldi(tx0 + iy1), tab + nt_ldi(tx1 + iy1)}; // hxy + ... for (x0,y0) (x1,y0) (x0,y1) (x1,y1)
	int const iz0[2] = {(int)cz0[0] << 2, (int)cz0[1] << 2}, iz1[2] = {(int)cz1[0] << 2, (int)cz1[1] << 2};
	nv2 nlo[4], nhi[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		float const fxc = (c & 1) ? f1x : f0x, fyc = (c & 2) ? f1y : f0y;
		char const *const g0a = tab + NOISE3_GX + nt_ldi(t[c] + iz0[0]), *const g0b = tab + NOISE3_GX + nt_ldi(t[c] + iz0[1]);
		char const *const g1a = tab + NOISE3_GX + nt_ldi(t[c] + iz1[0]), *const g1b = tab + NOISE3_GX + nt_ldi(t[c] + iz1[1]);
		nv2 const gx0 = {nt_ldf(g0a), nt_ldf(g0b)}, gy0 = {nt_ldf(g0a + NOISE3_GS), nt_ldf(g0b + NOISE3_GS)}, gz0 = {nt_ldf(g0a + 2*NOISE3_GS), nt_ldf(g0b + 2*NOISE3_GS)};
		nv2 const gx1 = {nt_ldf(g1a), nt_ldf(g1b)}, gy1 = {nt_ldf(g1a + NOISE3_GS), nt_ldf(g1b + NOISE3_GS)}, gz1 = {nt_ldf(g1a + 2*NOISE3_GS), nt_ldf(g1b + 2*NOISE3_GS)};
		nlo[c] = gx0*fxc + gy0*fyc + gz0*f0z;
		nhi[c] = gx1*fxc + gy1*fyc + gz1*f1z;
	}
	float const ux = gl_fade(f0x), uy = gl_fade(f0y);
	nv2 const uz = gl_fade(f0z);
	nv2 const z0 = gl_mix(nlo[0], nhi[0], uz), z1 = gl_mix(nlo[1], nhi[1], uz), z2 = gl_mix(nlo[2], nhi[2], uz), z3 = gl_mix(nlo[3], nhi[3], uz);
	nv2 const y0 = gl_mix(z0, z2, nt_bc<nv2>(uy)), y1 = gl_mix(z1, z3, nt_bc<nv2>(uy));
	return 2.2f*gl_mix(y0, y1, nt_bc<nv2>(ux));
}
// glm::simplex(vec3) for two positions
template<bool CHECK> TERRA_HD nv2 simplex3_lut(nv2 vx, nv2 vy, nv2 vz, char const *tab) {
	float const G = (float)(1.0/6.0), F = (float)(1.0/3.0);
	nv2 const one = nt_bc<nv2>(1.0f), zero = nt_bc<nv2>(0.0f);
	nv2 const skew = vx*F + vy*F + vz*F;
	nv2 const c0 = nt_floor(vx + skew), c1 = nt_floor(vy + skew), c2 = nt_floor(vz + skew);
	if (CHECK && TERRA_UNLIKELY(!(nt_all_below(c0, 4194304.0f) && nt_all_below(c1, 4194304.0f) && nt_all_below(c2, 4194304.0f)))) {
		return nv2{simplex3(vx[0], vy[0], vz[0]), simplex3(vx[1], vy[1], vz[1])};
	}
	nv2 const unskew = c0*G + c1*G + c2*G;
	nv2 const a0 = vx - c0 + unskew, a1 = vy - c1 + unskew, a2 = vz - c2 + unskew; // x0
	// g = step(x0.yzx, x0.xyz): 1 where x0 >= x0.yzx; l = 1 - g; i1 = min(g, l.zxy), i2 = max(g, l.zxy) -- on {0, 1} that is "and" / "or"
	ni2 const g0 = ~(a0 < a1), g1 = ~(a1 < a2), g2 = ~(a2 < a0);
	ni2 const p0 = g0 & ~g2, p1 = g1 & ~g0, p2 = g2 & ~g1; // i1
	ni2 const q0 = g0 | ~g2, q1 = g1 | ~g0, q2 = g2 | ~g1; // i2
	nv2 const o10 = nt_sel(p0, one, zero), o11 = nt_sel(p1, one, zero), o12 = nt_sel(p2, one, zero);
	nv2 const o20 = nt_sel(q0, one, zero), o21 = nt_sel(q1, one, zero), o22 = nt_sel(q2, one, zero);
	nv2 const b0 = a0 - o10 + G, b1 = a1 - o11 + G, b2 = a2 - o12 + G;
	nv2 const d0 = a0 - o20 + F, d1 = a1 - o21 + F, d2 = a2 - o22 + F;
	nv2 const e0 = a0 - 0.5f, e1 = a1 - 0.5f, e2 = a2 - 0.5f;
	nv2 const r0 = gl_mod289<nv2>(c0), r1 = gl_mod289<nv2>(c1), r2 = gl_mod289<nv2>(c2);
	nv2 dp[4];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		int const i0 = (int)r0[e] << 2, i1 = (int)r1[e] << 2, i2 = (int)r2[e] << 2;
		int const x1 = p0[e] ? 4 : 0, y1 = p1[e] ? 4 : 0, z1 = p2[e] ? 4 : 0, x2 = q0[e] ? 4 : 0, y2 = q1[e] ? 4 : 0, z2 = q2[e] ? 4 : 0;
		// permute(permute(permute(ci.z + off.z) + ci.y + off.y) + ci.x + off.x) for off = 0, i1, i2, 1
		char const *const gA = tab + NOISE3_GX + nt_ldi(tab + nt_ldi(tab + nt_ldi(tab + i2)      + i1)      + i0);
		char const *const gB = tab + NOISE3_GX + nt_ldi(tab + nt_ldi(tab + nt_ldi(tab + i2 + z1) + i1 + y1) + i0 + x1);
		char const *const gC = tab + NOISE3_GX + nt_ldi(tab + nt_ldi(tab + nt_ldi(tab + i2 + z2) + i1 + y2) + i0 + x2);
		char const *const gD = tab + NOISE3_GX + nt_ldi(tab + nt_ldi(tab + nt_ldi(tab + i2 + 4)  + i1 + 4)  + i0 + 4);
		dp[0][e] = nt_ldf(gA)*a0[e] + nt_ldf(gA + NOISE3_GS)*a1[e] + nt_ldf(gA + 2*NOISE3_GS)*a2[e];
		dp[1][e] = nt_ldf(gB)*b0[e] + nt_ldf(gB + NOISE3_GS)*b1[e] + nt_ldf(gB + 2*NOISE3_GS)*b2[e];
		dp[2][e] = nt_ldf(gC)*d0[e] + nt_ldf(gC + NOISE3_GS)*d1[e] + nt_ldf(gC + 2*NOISE3_GS)*d2[e];
		dp[3][e] = nt_ldf(gD)*e0[e] + nt_ldf(gD + NOISE3_GS)*e1[e] + nt_ldf(gD + 2*NOISE3_GS)*e2[e];
	}
	// (everything is finite here: std::max(m, 0) is fmaxf)
	nv2 wa = nt_max0(0.6f - (a0*a0 + a1*a1 + a2*a2)), wb = nt_max0(0.6f - (b0*b0 + b1*b1 + b2*b2)), wc = nt_max0(0.6f - (d0*d0 + d1*d1 + d2*d2)), wd = nt_max0(0.6f - (e0*e0 + e1*e1 + e2*e2));
	wa = wa*wa; wb = wb*wb; wc = wc*wc; wd = wd*wd;
	wa = wa*wa; wb = wb*wb; wc = wc*wc; wd = wd*wd;
	return 42.0f*((wa*dp[0] + wb*dp[1]) + (wc*dp[2] + wd*dp[3]));
}

// ---- noise shaping (src/mesh_gen.cpp:555-571)
TERRA_HD float postproc_noise_zval(float z, hmap_params_t const &h) {
	if (z > h.plat_bot) {z = h.plat_bot + h.plat_h*(z - h.plat_bot) + min_std(h.plat_max, h.plat_s*(z - h.plat_bot));}
	if (z > h.crat_h  ) {z = h.crat_h - h.crat_s*(z - h.crat_h);}
	if (z > h.crack_lo && z < h.crack_hi) {z -= h.crack_d*min_std(z - h.crack_lo, h.crack_hi - z);}
	return z;
}
TERRA_HD float apply_noise_shape_final(float n, int shape, hmap_params_t const &h) {
	if      (shape == 1) {n = (float)((double)fabsf(n) - 2.0);}
	else if (shape == 2) {n = (float)(3.5 - (double)fabsf(n));}
	return postproc_noise_zval(n, h);
}

// ---- fBm (gen_noise, src/mesh_gen.cpp:706-730). SIMPLEX selects glm::simplex vs glm::perlin.
TERRA_HD float nt_octave_shape(float n, int shape) { // billowy / ridged octave: the constant is a double in the reference
	if      (shape == 1) {n = (float)((double)fabsf(n) - 0.40);}
	else if (shape == 2) {n = (float)(0.45 - (double)fabsf(n));}
	return n;
}
TERRA_HD nv2 nt_octave_shape(nv2 n, int shape) {return nv2{nt_octave_shape(n[0], shape), nt_octave_shape(n[1], shape)};}
// NS = how one lattice-noise sample is evaluated: noise_direct_t computes every hash and gradient (below); the grid kernels pass an evaluator that
// looks the lattice-point part up in LDS tables (terra_kernels.hpp: noise_lds_t) -- same values, fewer instructions
struct noise_direct_t {
	template<class T> TERRA_HD T simplex(T x, T y) const {return simplex2_t<T>(x, y);}
	template<class T> TERRA_HD T perlin(T x, T y) const {return perlin2_t<T>(x, y);}
};
template<bool SIMPLEX, class T, class NS = noise_direct_t> TERRA_HD T fbm2_t(T xv, T yv, int shape, unsigned end_octave, float rx, float ry, NS const &ns = NS()) {
	T zval = nt_bc<T>(0.0f);
	float mag = 1.0f, freq = 1.0f;
	for (unsigned i = 0; i < end_octave; ++i) {
		T const qx = freq*xv + rx, qy = freq*yv + ry;
		T n = SIMPLEX ? ns.simplex(qx, qy) : ns.perlin(qx, qy);
		if (shape != 0) {n = nt_octave_shape(n, shape);}
		zval += mag*n;
		mag  *= 0.5f;
		freq *= 1.92f;
		rx   *= 1.5f; // (float)((double)rx*1.5): one correctly rounded product either way
		ry   *= 1.5f;
	}
	return zval;
}
template<bool SIMPLEX> TERRA_HD float fbm2(float xv, float yv, int shape, unsigned end_octave, float rx, float ry) {return fbm2_t<SIMPLEX, float>(xv, yv, shape, end_octave, rx, ry);}

// get_hmap_scale (src/mesh_gen.cpp:550-553)
TERRA_HD float hmap_scale(int mode, noise_consts_t const &nc) {
	float const scale = (mode == MGEN_SIMPLEX || mode == MGEN_SIMPLEX_GPU || mode == MGEN_DWARP_GPU) ? 16.0f : 32.0f;
	return scale*nc.MESH_HEIGHT*nc.mesh_height_scale*nc.mesh_scale_z_inv;
}

TERRA_HD float nt_add_d(float v, double c) {return (float)((double)v + c);} // vec2(xv, yv) + vec2(5.2, 1.3): double literals in the reference
TERRA_HD nv2   nt_add_d(nv2 v, double c)   {return nv2{nt_add_d(v[0], c), nt_add_d(v[1], c)};}
TERRA_HD float nt_postproc(float z, hmap_params_t const &h) {return postproc_noise_zval(z, h);}
TERRA_HD nv2   nt_postproc(nv2 z, hmap_params_t const &h) {return nv2{postproc_noise_zval(z[0], h), postproc_noise_zval(z[1], h)};}

// get_noise_zval (src/mesh_gen.cpp:734-751). MODE is MGEN_SIMPLEX / MGEN_PERLIN / MGEN_SIMPLEX_GPU / MGEN_DWARP_GPU.
template<int MODE, class T, class NS = noise_direct_t> TERRA_HD T noise_zval_t(T xval, T yval, int shape, noise_consts_t const &nc, NS const &ns = NS()) {
	constexpr bool SIMPLEX = (MODE != MGEN_PERLIN);
	float const xy_scale = 0.0007f*nc.mesh_scale; // MESH_SCALE_FACTOR
	T xv = xy_scale*xval, yv = xy_scale*yval;
	unsigned const end_octave = NUM_FREQ_COMP - nc.start_eval_sin/N_RAND_SIN2;
	if (MODE == MGEN_DWARP_GPU) {
		float const scale = 0.2f;
		T const dx1 = fbm2_t<SIMPLEX, T, NS>(nt_add_d(xv, 0.0), nt_add_d(yv, 0.0), shape, end_octave, nc.rx, nc.ry, ns);
		T const dy1 = fbm2_t<SIMPLEX, T, NS>(nt_add_d(xv, 5.2), nt_add_d(yv, 1.3), shape, end_octave, nc.rx, nc.ry, ns);
		T const wx = xv + scale*dx1, wy = yv + scale*dy1;
		T const dx2 = fbm2_t<SIMPLEX, T, NS>(nt_add_d(wx, 1.7), nt_add_d(wy, 9.2), shape, end_octave, nc.rx, nc.ry, ns);
		T const dy2 = fbm2_t<SIMPLEX, T, NS>(nt_add_d(wx, 8.3), nt_add_d(wy, 2.8), shape, end_octave, nc.rx, nc.ry, ns);
		xv += scale*dx2; yv += scale*dy2;
	}
	T z = fbm2_t<SIMPLEX, T, NS>(xv, yv, shape, end_octave, nc.rx, nc.ry, ns);
	z = nt_postproc(z, nc.hp);
	return z*hmap_scale(MODE, nc);
}
template<int MODE> TERRA_HD float noise_zval(float xval, float yval, int shape, noise_consts_t const &nc) {return noise_zval_t<MODE, float>(xval, yval, shape, nc);}

// ---- the grid kernels' form of gen_noise / get_noise_zval: the per-octave scalars (freq *= 1.92, mag *= 0.5, rx *= 1.5, ry *= 1.5: the same float products
// in the same order, made once on the host) arrive as kernel arguments = scalar registers, and the "is every lattice coordinate a small integer" test of
// the table look-ups is made once per fBm sum from a bound on the sample positions instead of once per octave.
struct noise_oct_t {float freq[NUM_FREQ_COMP], mag[NUM_FREQ_COMP], rx[NUM_FREQ_COMP], ry[NUM_FREQ_COMP]; float freq_max, r_max; unsigned end_octave;};
TERRA_HD noise_oct_t make_noise_oct(noise_consts_t const &nc) {
	noise_oct_t o;
	o.end_octave = NUM_FREQ_COMP - nc.start_eval_sin/N_RAND_SIN2;
	float mag = 1.0f, freq = 1.0f, rx = nc.rx, ry = nc.ry;
	o.freq_max = 0.0f; o.r_max = 0.0f;
	for (unsigned i = 0; i < (unsigned)NUM_FREQ_COMP; ++i) {
		o.freq[i] = freq; o.mag[i] = mag; o.rx[i] = rx; o.ry[i] = ry;
		if (i < o.end_octave) {o.freq_max = max_std(o.freq_max, fabsf(freq)); o.r_max = max_std(o.r_max, max_std(fabsf(rx), fabsf(ry)));}
		mag *= 0.5f; freq *= 1.92f; rx *= 1.5f; ry *= 1.5f;
	}
	return o;
}
template<bool SIMPLEX> TERRA_HD nv2 fbm2_tab(nv2 xv, nv2 yv, int shape, noise_oct_t const &oc, noise_tab_t const &ns) {
	nv2 zval = nt_bc<nv2>(0.0f);
	// |q| <= freq_max*|v| + r_max for every octave; simplex's skewed coordinate is below 1.74*max|q|: under 2^20 here, every lattice coordinate is an
	// integer far below 2^22 (a NaN fails the comparison and takes the checked loop)
	float const vmax = fmaxf(fmaxf(fabsf(xv[0]), fabsf(xv[1])), fmaxf(fabsf(yv[0]), fabsf(yv[1])));
	if (TERRA_LIKELY(vmax*oc.freq_max + oc.r_max < 1048576.0f)) {
		noise_tab_nocheck_t const nn{ns.stab, ns.ptab};
		for (unsigned i = 0; i < oc.end_octave; ++i) {
			nv2 const qx = oc.freq[i]*xv + oc.rx[i], qy = oc.freq[i]*yv + oc.ry[i];
			nv2 n = SIMPLEX ? nn.simplex(qx, qy) : nn.perlin(qx, qy);
			if (shape != 0) {n = nt_octave_shape(n, shape);}
			zval += oc.mag[i]*n;
		}
		return zval;
	}
	for (unsigned i = 0; i < oc.end_octave; ++i) {
		nv2 const qx = oc.freq[i]*xv + oc.rx[i], qy = oc.freq[i]*yv + oc.ry[i];
		nv2 n = SIMPLEX ? ns.simplex(qx, qy) : ns.perlin(qx, qy);
		if (shape != 0) {n = nt_octave_shape(n, shape);}
		zval += oc.mag[i]*n;
	}
	return zval;
}
// ---- block tables for REGULAR fBm sums.  When the sample positions of a block of cells are a grid (every fBm sum except the domain-warped ones), the lattice cells it touches in
// octave i form a small rectangle [cxmin, cxmax] x [cymin, cymax] of (skewed, for simplex) lattice coordinates: the floating-point expressions that lead from a cell index to its
// lattice coordinate are monotone in x and in y (positive steps and scales, rounding is monotone), so the rectangle's corners are the lattice cells of the block's first and last cell.
// The block then builds, once per sum, one record per lattice cell and octave with the gradient terms of its four lattice points -- fetched through the very look-up chain of
// simplex2_lut / perlin2_lut, so the same bits -- and a cell only converts its lattice coordinate into a record index: no mod 289, no permute tables, no per-point index sums.
// Record: simplex 12 floats {A(a0,h,n), B1 = (1,0), B2 = (0,1), C}; Perlin 8 floats {g00(gxn,gyn), g10, g01, g11}.
#if defined(__HIP_DEVICE_COMPILE__)
#define TERRA_BLOCK_SYNC() __syncthreads()
#else
#define TERRA_BLOCK_SYNC() do {} while (0)
#endif
constexpr unsigned NOISE_BT_FLOATS = 6144; // 24 KB of records per block
struct noise_bt_meta_t {int cxmin, cymin, nxl, off;}; // per octave: lattice origin, cells per record row, first float of the octave's records
struct noise_btab_t {float const *rec; noise_bt_meta_t const *meta;};
// lattice coordinate of a sample position in octave i (the expressions of simplex2_lut / perlin2_lut)
template<bool SIMPLEX> TERRA_HD void noise_lattice_of(float vx, float vy, noise_oct_t const &oc, unsigned i, float &cx, float &cy) {
	float const qx = oc.freq[i]*vx + oc.rx[i], qy = oc.freq[i]*vy + oc.ry[i];
	if (SIMPLEX) {float const C1 = 0.366025403784439f, skew = qx*C1 + qy*C1; cx = floorf(qx + skew); cy = floorf(qy + skew);}
	else {cx = floorf(qx); cy = floorf(qy);}
}
// (vx0, vy0) / (vx1, vy1): sample positions of the block's first / last cell (the caller guarantees the monotone mapping).  Called by every thread of the block;
// returns false (for all of them alike) when the records do not fit or a lattice coordinate is not a small integer: the caller then uses the per-cell tables.
template<bool SIMPLEX> TERRA_HD bool noise_bt_octave(float vx0, float vy0, float vx1, float vy1, noise_oct_t const &oc, unsigned i, unsigned &total, noise_bt_meta_t &m) {
	constexpr int RF = SIMPLEX ? 12 : 8;
	float ax, ay, bx, by;
	noise_lattice_of<SIMPLEX>(vx0, vy0, oc, i, ax, ay); noise_lattice_of<SIMPLEX>(vx1, vy1, oc, i, bx, by);
	if (!(fabsf(ax) < 4194304.0f && fabsf(ay) < 4194304.0f && fabsf(bx) < 4194304.0f && fabsf(by) < 4194304.0f && ax <= bx && ay <= by)) return false;
	int const nxl = (int)(bx - ax) + 1, nyl = (int)(by - ay) + 1;
	if (nxl > 4096 || nyl > 4096) return false;
	m.cxmin = (int)ax; m.cymin = (int)ay; m.nxl = nxl; m.off = (int)total;
	total += (unsigned)(nxl*nyl*RF);
	return total <= NOISE_BT_FLOATS;
}
template<bool SIMPLEX> TERRA_HD bool noise_blocktab_build(float vx0, float vy0, float vx1, float vy1, noise_oct_t const &oc, char const *lut, float *rec, noise_bt_meta_t *meta, unsigned tid, unsigned nthreads) {
	constexpr int RF = SIMPLEX ? 12 : 8;
	unsigned total = 0;
	noise_bt_meta_t m;
	for (unsigned i = 0; i < oc.end_octave; ++i) {if (!noise_bt_octave<SIMPLEX>(vx0, vy0, vx1, vy1, oc, i, total, m)) return false;} // the same answer in every thread
	TERRA_BLOCK_SYNC(); // the previous sum's records and metadata are no longer read
	if (tid == 0) {unsigned t2 = 0; for (unsigned i = 0; i < oc.end_octave; ++i) {noise_bt_octave<SIMPLEX>(vx0, vy0, vx1, vy1, oc, i, t2, m); meta[i] = m;}}
	TERRA_BLOCK_SYNC();
	unsigned const ncorner = SIMPLEX ? total/3u : total/2u; // one entry per (cell, lattice point): 3 floats (simplex) / 2 floats (Perlin)
	for (unsigned e = tid; e < ncorner; e += nthreads) {
		unsigned const f = e*(SIMPLEX ? 3u : 2u); // first float of the entry
		unsigned i = 0;
		while (i + 1 < oc.end_octave && (unsigned)meta[i + 1].off <= f) {++i;}
		noise_bt_meta_t const mi = meta[i];
		unsigned const rel = f - (unsigned)mi.off, cell = rel/(unsigned)RF, k = (rel % (unsigned)RF)/(SIMPLEX ? 3u : 2u);
		int const lx = (int)(cell % (unsigned)mi.nxl), ly = (int)(cell/(unsigned)mi.nxl);
		float const X = (float)(mi.cxmin + lx), Y = (float)(mi.cymin + ly);
		if (SIMPLEX) { // the look-ups of simplex2_lut for cx = X, cy = Y
			int const mx4 = (int)(gl_mod289_small(X)*4.0f), ry16 = (int)(gl_mod289_raw(Y)*16.0f);
			nt_i4 const si = *(nt_i4 const *)(lut + ry16);
			char const *g = lut + mx4;
			char const *p = (k == 0) ? g + si.x : ((k == 1) ? g + si.y : ((k == 2) ? g + si.z : g + si.z + 4));
			rec[f] = nt_ldf(p); rec[f + 1] = nt_ldf(p + NOISE_LUT_S_N*4); rec[f + 2] = nt_ldf(p + 2*NOISE_LUT_S_N*4);
		}
		else { // the look-ups of perlin2_lut for flx = X, fly = Y: k = 0..3 = (x0,y0) (x1,y0) (x0,y1) (x1,y1)
			int const rx4 = (int)(gl_mod289_raw(X)*4.0f), iy0 = (int)gl_mod289_small(Y);
			int iy1 = iy0 + 1; iy1 = (iy1 == 289) ? 0 : iy1;
			int const *pip = (int const *)(lut + rx4);
			char const *p = lut + (((k >> 1) ? iy1 : iy0) << 2) + ((k & 1u) ? pip[1] : pip[0]);
			rec[f] = nt_ldf(p); rec[f + 1] = nt_ldf(p + NOISE_LUT_P_N*4);
		}
	}
	TERRA_BLOCK_SYNC();
	return true;
}
// glm::simplex(vec2) for two cells of a regular sum, octave i, gradient terms from the block's records
TERRA_HD nv2 simplex2_bt(nv2 vx, nv2 vy, noise_bt_meta_t const &m, float const *rec) {
	float const C0 = 0.211324865405187f, C1 = 0.366025403784439f, C2 = -0.577350269189626f;
	nv2 const one = nt_bc<nv2>(1.0f), zero = nt_bc<nv2>(0.0f);
	nv2 const skew = vx*C1 + vy*C1;
	nv2 const cx = nt_floor(vx + skew), cy = nt_floor(vy + skew);
	nv2 const unskew = cx*C0 + cy*C0;
	nv2 const ax = vx - cx + unskew, ay = vy - cy + unskew;
	ni2 const lower = (ax > ay);
	nv2 const ox = nt_sel(lower, one, zero), oy = one - ox;
	nv2 const bx = (ax + C0) - ox, by = (ay + C0) - oy;
	nv2 const ex = ax + C2, ey = ay + C2;
	nv2 ma = nt_max0(0.5f - (ax*ax + ay*ay));
	nv2 mb = nt_max0(0.5f - (bx*bx + by*by));
	nv2 mc = nt_max0(0.5f - (ex*ex + ey*ey));
	ma = ma*ma; mb = mb*mb; mc = mc*mc;
	ma = ma*ma; mb = mb*mb; mc = mc*mc;
	float const *r[2]; int bo[2];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		int const lx = (int)cx[e] - m.cxmin, ly = (int)cy[e] - m.cymin;
		r[e] = rec + m.off + (ly*m.nxl + lx)*12;
		bo[e] = lower[e] ? 3 : 6;
	}
	nv2 const a0a = {r[0][0], r[1][0]}, ha = {r[0][1], r[1][1]}, na = {r[0][2], r[1][2]};
	nv2 const a0b = {r[0][bo[0]], r[1][bo[1]]}, hb = {r[0][bo[0] + 1], r[1][bo[1] + 1]}, nb = {r[0][bo[0] + 2], r[1][bo[1] + 2]};
	nv2 const a0c = {r[0][9], r[1][9]}, hc = {r[0][10], r[1][10]}, ncc = {r[0][11], r[1][11]};
	ma *= na; mb *= nb; mc *= ncc;
	nv2 const da = a0a*ax + ha*ay;
	nv2 const db = a0b*bx + hb*by;
	nv2 const dc = a0c*ex + hc*ey;
	return 130.0f*(ma*da + mb*db + mc*dc);
}
TERRA_HD nv2 perlin2_bt(nv2 px, nv2 py, noise_bt_meta_t const &m, float const *rec) {
	nv2 const flx = nt_floor(px), fly = nt_floor(py);
	nv2 const frx = px - flx, fry = py - fly;
	nv2 const fx0 = frx - 0.0f, fy0 = fry - 0.0f, fx1 = frx - 1.0f, fy1 = fry - 1.0f;
	float const *r[2];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		int const lx = (int)flx[e] - m.cxmin, ly = (int)fly[e] - m.cymin;
		r[e] = rec + m.off + (ly*m.nxl + lx)*8;
	}
	nv2 const gx00 = {r[0][0], r[1][0]}, gy00 = {r[0][1], r[1][1]}, gx10 = {r[0][2], r[1][2]}, gy10 = {r[0][3], r[1][3]};
	nv2 const gx01 = {r[0][4], r[1][4]}, gy01 = {r[0][5], r[1][5]}, gx11 = {r[0][6], r[1][6]}, gy11 = {r[0][7], r[1][7]};
	nv2 const d00 = gx00*fx0 + gy00*fy0;
	nv2 const d10 = gx10*fx1 + gy10*fy0;
	nv2 const d01 = gx01*fx0 + gy01*fy1;
	nv2 const d11 = gx11*fx1 + gy11*fy1;
	nv2 const ux = gl_fade(fx0), uy = gl_fade(fy0);
	nv2 const lo = gl_mix(d00, d10, ux), hi = gl_mix(d01, d11, ux);
	return 2.3f*gl_mix(lo, hi, uy);
}
// the fBm sum of two cells inside the block the records were built for
template<bool SIMPLEX> TERRA_HD nv2 fbm2_bt(nv2 xv, nv2 yv, int shape, noise_oct_t const &oc, noise_btab_t const &bt) {
	nv2 zval = nt_bc<nv2>(0.0f);
	for (unsigned i = 0; i < oc.end_octave; ++i) {
		nv2 const qx = oc.freq[i]*xv + oc.rx[i], qy = oc.freq[i]*yv + oc.ry[i];
		nv2 n = SIMPLEX ? simplex2_bt(qx, qy, bt.meta[i], bt.rec) : perlin2_bt(qx, qy, bt.meta[i], bt.rec);
		if (shape != 0) {n = nt_octave_shape(n, shape);}
		zval += oc.mag[i]*n;
	}
	return zval;
}

template<int MODE> TERRA_HD nv2 noise_zval_tab(nv2 xval, nv2 yval, int shape, noise_consts_t const &nc, noise_oct_t const &oc, noise_tab_t const &ns) {
	constexpr bool SIMPLEX = (MODE != MGEN_PERLIN);
	float const xy_scale = 0.0007f*nc.mesh_scale; // MESH_SCALE_FACTOR
	nv2 xv = xy_scale*xval, yv = xy_scale*yval;
	if (MODE == MGEN_DWARP_GPU) {
		float const scale = 0.2f;
		nv2 const dx1 = fbm2_tab<SIMPLEX>(nt_add_d(xv, 0.0), nt_add_d(yv, 0.0), shape, oc, ns);
		nv2 const dy1 = fbm2_tab<SIMPLEX>(nt_add_d(xv, 5.2), nt_add_d(yv, 1.3), shape, oc, ns);
		nv2 const wx = xv + scale*dx1, wy = yv + scale*dy1;
		nv2 const dx2 = fbm2_tab<SIMPLEX>(nt_add_d(wx, 1.7), nt_add_d(wy, 9.2), shape, oc, ns);
		nv2 const dy2 = fbm2_tab<SIMPLEX>(nt_add_d(wx, 8.3), nt_add_d(wy, 2.8), shape, oc, ns);
		xv += scale*dx2; yv += scale*dy2;
	}
	nv2 z = fbm2_tab<SIMPLEX>(xv, yv, shape, oc, ns);
	z = nt_postproc(z, nc.hp);
	return z*hmap_scale(MODE, nc);
}

// the same for a cell pair inside a block whose records are built (simplex / Perlin sums only: the warped sums of the domain warp have no regular lattice footprint)
template<int MODE> TERRA_HD nv2 noise_zval_bt(nv2 xval, nv2 yval, int shape, noise_consts_t const &nc, noise_oct_t const &oc, noise_btab_t const &bt) {
	static_assert(MODE != MGEN_DWARP_GPU, "regular sums only");
	float const xy_scale = 0.0007f*nc.mesh_scale;
	nv2 z = fbm2_bt<(MODE != MGEN_PERLIN)>(xy_scale*xval, xy_scale*yval, shape, oc, bt);
	z = nt_postproc(z, nc.hp);
	return z*hmap_scale(MODE, nc);
}

// ---- glaciate + islands + volcano epilogue of eval_index (src/mesh_gen.cpp:358-385,782-790)
// pow(val, custom_glaciate_exp) is libm's powf in the reference (float arguments): reproduced bit for bit by terra_powf.hpp
TERRA_HD float glaciate_exp_fn(float v, float custom_exp) {return (custom_exp == 0.0f) ? v*v*v : glibc_powf(v, custom_exp);}

TERRA_HD float volcano_height(float xi, float yi, noise_consts_t const &nc, sin_lut_t const &lut) { // src/mesh_gen.cpp:364-371
	float const freq = nc.mesh_scale/nc.hp.volcano_width, x = freq*xi, y = freq*yi, dist = sqrtf(x*x + y*y);
	if ((double)dist > 2.0) return 0.0f;
	float const val = lut.COSF(x)*lut.COSF(y);
	double const hole_d = 400.0*((double)val - 0.999);
	float const hole = (float)((0.0 < hole_d) ? hole_d : 0.0);
	float const peak = (float)(0.08*(double)val/(double)max_std(0.04f, dist));
	return nc.hp.volcano_height*max_std(0.0f, (peak - hole))*nc.mesh_scale_z_inv;
}

// smx = sine_mag_terms[x], smy = sine_mag_terms[nx+y] (enable_glaciate, src/mesh_gen.cpp:640-650); xg/yg = eval_index's (x*mdx+mx0)*DX_VAL_INV
TERRA_HD float glaciate_epilogue(float z, float smx, float smy, float sine_offset, float xg, float yg, noise_consts_t const &nc, sin_lut_t const &lut) {
	if (nc.glaciate) {
		float const relh = (z + nc.zmax_est)*nc.zmax_est2_inv;
		z = glaciate_exp_fn(relh, nc.custom_glaciate_exp)*nc.zmax_est2 - nc.zmax_est;
	}
	if (nc.hp.sine_mag > 0.0f) {
		z += smx*smy + sine_offset;
		if (nc.hp.volcano_width > 0.0f && nc.hp.volcano_height > 0.0f) {z += volcano_height(xg, yg, nc, lut);}
	}
	return z;
}

} // namespace terra
