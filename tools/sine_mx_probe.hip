// Probe of the fused sine-grid kernel (3dworld_amd/csrc/terra_fused.hpp: k_sine_grid_mx) on its own: semantics of v_mfma_f32_32x32x2_f32's k order, parity with the
// fmaf chain, rate.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/sine_mx_probe.hip -o tools/_bin/sine_mx_probe
//   sine_mx_probe [n=16384] [kstart=10] [reps=20]
#include "../3dworld_amd/csrc/terra_fused.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

#define CK(e) do {hipError_t const e_ = (e); if (e_ != hipSuccess) {printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1);}} while (0)
static uint32_t bits(float f) {uint32_t u; memcpy(&u, &f, 4); return u;}

int main(int argc, char **argv) {
	unsigned const n = (argc > 1) ? atoi(argv[1]) : 16384;
	int const kstart = (argc > 2) ? atoi(argv[2]) : 10;
	int const reps = (argc > 3) ? atoi(argv[3]) : 20;
	unsigned const np = (n + 127)/128*128;
	std::mt19937 rng(7);
	std::uniform_real_distribution<float> U(-1.0f, 1.0f);
	std::vector<float> hx((size_t)90*np, 0.0f), hy((size_t)90*np, 0.0f), smx(np, 0.0f), smy(np, 0.0f);
	for (int k = 0; k < 90; ++k) {float const amp = 1.0f/(1.0f + 0.1f*k); for (unsigned i = 0; i < n; ++i) {hx[(size_t)k*np + i] = amp*U(rng); hy[(size_t)k*np + i] = U(rng);}}
	for (unsigned i = 0; i < n; ++i) {smx[i] = U(rng); smy[i] = U(rng);}
	float *dx, *dy, *dsx, *dsy, *out; uint32_t *mm;
	CK(hipMalloc(&dx, hx.size()*4)); CK(hipMalloc(&dy, hy.size()*4)); CK(hipMalloc(&dsx, np*4)); CK(hipMalloc(&dsy, np*4)); CK(hipMalloc(&out, (size_t)n*n*4)); CK(hipMalloc(&mm, 8));
	CK(hipMemcpy(dx, hx.data(), hx.size()*4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hy.data(), hy.size()*4, hipMemcpyHostToDevice));
	CK(hipMemcpy(dsx, smx.data(), np*4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsy, smy.data(), np*4, hipMemcpyHostToDevice));
	terra::sgf_job_t J; memset(&J, 0, sizeof(J));
	J.xt = dx; J.yt = dy; J.smx = dsx; J.smy = dsy; J.out = out; J.mm = mm; J.nx = J.ny = n; J.nxp = J.nyp = np; J.ntx = np/128; J.nty = (n + 127)/128; J.rowgroup = getenv("SGF_RG") ? atoi(getenv("SGF_RG")) : 4;
	J.kstart = kstart; J.glaciate = 1; J.sine_mag = 1; J.zmax_est = 3.1f; J.zmax_est2 = 6.2f; J.zmax_est2_inv = 1.0f/6.2f; J.sine_offset = -0.25f;
	unsigned const nb = J.ntx*J.nty; unsigned grid = (nb + 7)/8*8; if (getenv("SGF_GRID")) {unsigned const g = atoi(getenv("SGF_GRID")); if (g < grid) grid = g;}
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int variant = 0; variant < 2; ++variant) { // 0: bare sum (semantics), 1: with epilogue
		J.glaciate = J.sine_mag = variant;
		CK(hipMemset(mm, 0xFF, 8));
		for (int r = 0; r < 3; ++r) {hipLaunchKernelGGL(terra::k_sine_grid_mx<false>, dim3(grid), dim3(256), 0, 0, J);}
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0));
		for (int r = 0; r < reps; ++r) {hipLaunchKernelGGL(terra::k_sine_grid_mx<false>, dim3(grid), dim3(256), 0, 0, J);}
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
		double const flop = (double)n*n*(90 - kstart)*2;
		printf("variant %d: %u^2, %d terms: %.4f ms per launch, %.1f TFLOP/s, %.2f TB/s written, %.1f Gcells/s\n", variant, n, 90 - kstart, ms, flop/ms*1e-9, (double)n*n*4/ms*1e-9, (double)n*n/ms*1e-6);
		// sampled rows against the candidates
		std::vector<float> row(n);
		unsigned long long cnt = 0, bad_chain = 0, bad_rev = 0, bad_pairsum = 0; double maxrel = 0;
		float hmn = INFINITY, hmx = -INFINITY;
		for (unsigned yi = 0; yi < 48; ++yi) {
			unsigned const y = (yi < 16) ? yi*9 % n : (unsigned)(rng() % n);
			CK(hipMemcpy(row.data(), out + (size_t)y*n, n*4, hipMemcpyDeviceToHost));
			for (unsigned x = 0; x < n; ++x) {
				float zc = 0.0f, zr = 0.0f, zp = 0.0f; double zd = 0.0, za = 0.0;
				int const nk = 90 - kstart;
				for (int k = kstart; k < 90; ++k) {zc = fmaf(hx[(size_t)k*np + x], hy[(size_t)k*np + y], zc);}
				{ // pair members in the other order
					int k = kstart; if (nk & 1) {zr = fmaf(hx[(size_t)k*np + x], hy[(size_t)k*np + y], zr); ++k;}
					for (; k < 90; k += 2) {zr = fmaf(hx[(size_t)(k + 1)*np + x], hy[(size_t)(k + 1)*np + y], zr); zr = fmaf(hx[(size_t)k*np + x], hy[(size_t)k*np + y], zr);}
				}
				{ // a pair summed exactly, rounded once
					int k = kstart; if (nk & 1) {zp = fmaf(hx[(size_t)k*np + x], hy[(size_t)k*np + y], zp); ++k;}
					for (; k < 90; k += 2) {zp = (float)((double)hx[(size_t)k*np + x]*(double)hy[(size_t)k*np + y] + (double)hx[(size_t)(k + 1)*np + x]*(double)hy[(size_t)(k + 1)*np + y] + (double)zp);}
				}
				for (int k = kstart; k < 90; ++k) {double const t = (double)hx[(size_t)k*np + x]*(double)hy[(size_t)k*np + y]; zd += t; za += fabs(t);}
				if (variant) {
					float const rel = (zc + J.zmax_est)*J.zmax_est2_inv; zc = fmaf((rel*rel)*rel, J.zmax_est2, -J.zmax_est); zc = zc + fmaf(smx[x], smy[y], J.sine_offset);
					zr = zp = zc;
				}
				else {double const e = fabs((double)row[x] - zd)/za; if (e > maxrel) maxrel = e;}
				++cnt; bad_chain += bits(zc) != bits(row[x]); bad_rev += bits(zr) != bits(row[x]); bad_pairsum += bits(zp) != bits(row[x]);
				hmn = fminf(hmn, row[x]); hmx = fmaxf(hmx, row[x]);
			}
		}
		uint32_t hmm[2]; CK(hipMemcpy(hmm, mm, 8, hipMemcpyDeviceToHost));
		printf("  %llu cells: mismatches vs fmaf chain in k order %llu, vs pair-reversed chain %llu, vs exact pair sums %llu; max |z - exact| / sum|terms| %.3g\n", cnt, bad_chain, bad_rev, bad_pairsum, maxrel);
		auto ord2f = [](uint32_t o) {uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; memcpy(&f, &u, 4); return f;};
		printf("  device min %g max %g (sampled rows: %g %g)\n", ord2f(hmm[0]), ord2f(~hmm[1]), hmn, hmx);
	}
	return 0;
}
