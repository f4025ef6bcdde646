// terra_simple_paths.hpp -- "one logical thread per output element" versions of the hot kernels, expressed through
// BACKEND::launch().  They are (1) what the test-only CPU emulator runs and (2) the on-device cross-check of the
// LDS-tiled fast kernels in terra_hip.hip (select with TERRA_SIMPLE_KERNELS=1).  CRTP: DERIVED provides launch().
#pragma once
#include "terra_driver.hpp"

namespace terra {

template<class DERIVED> struct simple_paths {
	DERIVED &self() {return *static_cast<DERIVED *>(this);}

	void sine_grid_simple(grid_job_t const &job, noise_consts_t const &nc, sin_lut_t const &L, float const *xt, float const *yt, float const *smx, float const *smy, float *out) {
		self().launch((size_t)job.nx*job.ny, [=] TERRA_LAMBDA (size_t i) {
			unsigned const x = (unsigned)(i % job.nx), y = (unsigned)(i / job.nx);
			out[i] = finish_cell(sine_cell(job, xt, yt, x, y), job, nc, L, smx, smy, x, y);
		});
	}
	void noise_grid_simple(grid_job_t const &job, noise_consts_t const &nc, sin_lut_t const &L, float const *smx, float const *smy, float *out) {
		self().launch((size_t)job.nx*job.ny, [=] TERRA_LAMBDA (size_t i) {
			unsigned const x = (unsigned)(i % job.nx), y = (unsigned)(i / job.nx);
			out[i] = finish_cell(noise_cell(job, nc, x, y), job, nc, L, smx, smy, x, y);
		});
	}
	// tiles: d_tab = [nux + nuy][90][130] tables, d_sm = [nux + nuy][130] sine-mag terms, d_m0 = per distinct tx / ty grid origin (mx0 / my0)
	void tile_grid_simple(uint32_t n, tile_ref_pod_t const *refs, uint32_t nux, float const *d_tab, float const *d_sm, float const *d_m0,
		int md, int shp, int kstart, bool use_sm, float sine_offset, noise_consts_t const &nc, sin_lut_t const &L, float dxv, float dyv, float *zvals)
	{
		unsigned const zv = 130;
		self().launch((size_t)n*zv*zv, [=] TERRA_LAMBDA (size_t i) {
			unsigned const t = (unsigned)(i / (zv*zv)), p = (unsigned)(i % (zv*zv)), y = p / zv, x = p % zv;
			tile_ref_pod_t const r = refs[t];
			grid_job_t job;
			job.mx0 = d_m0[r.xi]; job.my0 = d_m0[nux + r.yi]; job.mdx = dxv; job.mdy = dyv; job.nx = job.ny = zv; job.nxp = job.nyp = zv;
			job.mode = md; job.shape = shp; job.kstart = kstart; job.glaciate = 1; job.use_sine_mag = use_sm ? 1 : 0; job.sine_offset = sine_offset;
			float z;
			if (md == MGEN_SINE) {z = sine_cell(job, d_tab + (size_t)r.xi*F_TABLE_SIZE*zv, d_tab + (size_t)(nux + r.yi)*F_TABLE_SIZE*zv, x, y);}
			else {z = noise_cell(job, nc, x, y);}
			zvals[i] = finish_cell(z, job, nc, L, d_sm + (size_t)r.xi*zv, d_sm + (size_t)(nux + r.yi)*zv, x, y);
		});
	}
	// tile erosion on a global-memory padded scratch: one logical thread per tile, droplets in order
	void tile_erosion_simple(uint32_t n, float *zvals, erosion_consts_t const &ec, uint32_t iters, float *padded /* n*NX*NY */) {
		int const NX = ec.NX, NY = ec.NY, xs = ec.xsize, ys = ec.ysize;
		self().launch((size_t)n*NX*NY, [=] TERRA_LAMBDA (size_t i) { // clamp-padded copy (src/erosion.cpp:31-37)
			unsigned const t = (unsigned)(i / ((size_t)NX*NY)), p = (unsigned)(i % ((size_t)NX*NY));
			int const X = (int)(p % NX), Z = (int)(p / NX);
			padded[i] = zvals[(size_t)t*xs*ys + (size_t)imax(imin(Z - EROSION_PAD, ys-1), 0)*xs + imax(imin(X - EROSION_PAD, xs-1), 0)];
		});
		self().launch(n, [=] TERRA_LAMBDA (size_t t) {
			grid_view_t g; g.interior = padded + t*(size_t)NX*NY; g.border = nullptr; g.xsize = xs; g.ysize = ys; g.NX = NX; g.NY = NY;
			direct_mem_t m{g};
			for (uint32_t it = 0; it < iters; ++it) {simulate_droplet((int)it, m, ec);}
		}, 64);
		self().launch((size_t)n*xs*ys, [=] TERRA_LAMBDA (size_t i) { // unpad + clamp (src/erosion.cpp:158-162)
			unsigned const t = (unsigned)(i / ((size_t)xs*ys)), p = (unsigned)(i % ((size_t)xs*ys));
			int const x = (int)(p % xs), y = (int)(p / xs);
			zvals[i] = max_std(ec.min_zval, padded[(size_t)t*NX*NY + (size_t)(y + EROSION_PAD)*NX + (x + EROSION_PAD)]);
		});
	}
	// min / max of a float array as order-preserving uints (NaNs skipped): one logical thread per 2048-element chunk
	void minmax_simple(float const *vals, size_t n, uint32_t *d) {
		size_t const chunk = 2048, nchunks = (n + chunk - 1)/chunk;
		self().launch(nchunks, [=] TERRA_LAMBDA (size_t c) {
			size_t const b = c*chunk, e = (b + chunk < n) ? b + chunk : n;
			bool have = false; float lo = 0, hi = 0;
			for (size_t i = b; i < e; ++i) {
				float const v = vals[i];
				if (v != v) continue;
				if (!have) {lo = hi = v; have = true;}
				lo = min_std(lo, v); hi = max_std(hi, v);
			}
			if (have) {TERRA_ATOMIC_MIN(&d[0], f2ord(lo)); TERRA_ATOMIC_MIN(&d[1], ~f2ord(hi));} // max as min of the complement
		});
	}
	// voxel sine field: val = sum_k xv[k]*yv[k]*zv[k] (src/upsurface.cpp:60-70); d_tab = [nx + ny + nz][60]
	void voxel_sines_simple(float *out, uint32_t nx, uint32_t ny, uint32_t nz, float const *d_tab, float zscale, int normalize) {
		self().launch((size_t)nx*ny*nz, [=] TERRA_LAMBDA (size_t i) {
			unsigned const z = (unsigned)(i % nz), x = (unsigned)((i / nz) % nx), y = (unsigned)(i / ((size_t)nz*nx));
			float const *xv = d_tab + (size_t)x*VOX_SINES, *yv = d_tab + ((size_t)nx + y)*VOX_SINES, *zvp = d_tab + ((size_t)nx + ny + z)*VOX_SINES;
			float val = 0.0f;
			for (unsigned k = 0; k < VOX_SINES; ++k) {val += xv[k]*yv[k]*zvp[k];}
			val += (float)z*zscale;
			if (normalize) {val = clip_pm1(val);}
			out[i] = val;
		});
	}
};

} // namespace terra
