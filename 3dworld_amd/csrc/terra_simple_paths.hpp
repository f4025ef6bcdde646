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
			out[i] = job.fused ? finish_cell_fused(sine_cell_fused(job, xt, yt, x, y), job, nc, smx, smy, x, y) : finish_cell(sine_cell(job, xt, yt, x, y), job, nc, L, smx, smy, x, y);
		});
	}
	void noise_grid_simple(grid_job_t const &job, noise_consts_t const &nc, sin_lut_t const &L, float const *smx, float const *smy, float *out) {
		self().launch((size_t)job.nx*job.ny, [=] TERRA_LAMBDA (size_t i) {
			unsigned const x = (unsigned)(i % job.nx), y = (unsigned)(i / job.nx);
			out[i] = finish_cell(noise_cell(job, nc, x, y), job, nc, L, smx, smy, x, y);
		});
	}
	// tiles: xt / yt = k-major tables of all distinct tile columns / rows side by side (row lengths nxpv / nypv), d_sm = [nux + nuy][zv] sine-mag terms,
	// d_m0 = per distinct tx / ty grid origin (mx0 / my0)
	void tile_grid_simple(uint32_t n, tile_ref_pod_t const *refs, uint32_t nux, uint32_t /*nuy*/, float const *xt, float const *yt, uint32_t nxpv, uint32_t nypv,
		float const *d_sm, float const *d_m0, int md, int shp, int kstart, bool use_sm, float sine_offset, noise_consts_t const &nc, sin_lut_t const &L, float dxv, float dyv, float *zvals, uint32_t zv, bool glaciate = true, int fused = 0)
	{
		self().launch((size_t)n*zv*zv, [=] TERRA_LAMBDA (size_t i) {
			unsigned const t = (unsigned)(i / (zv*zv)), p = (unsigned)(i % (zv*zv)), y = p / zv, x = p % zv;
			tile_ref_pod_t const r = refs[t];
			grid_job_t job;
			job.mx0 = d_m0[r.xi]; job.my0 = d_m0[nux + r.yi]; job.mdx = dxv; job.mdy = dyv; job.nx = job.ny = zv; job.nxp = nxpv; job.nyp = nypv;
			job.mode = md; job.shape = shp; job.kstart = kstart; job.glaciate = glaciate ? 1 : 0; job.use_sine_mag = use_sm ? 1 : 0; job.sine_offset = sine_offset;
			float z;
			if (md == MGEN_SINE && fused) {zvals[i] = finish_cell_fused(sine_cell_fused(job, xt, yt, r.xi*zv + x, r.yi*zv + y), job, nc, d_sm + (size_t)r.xi*zv, d_sm + (size_t)(nux + r.yi)*zv, x, y); return;}
			if (md == MGEN_SINE) {z = sine_cell(job, xt, yt, r.xi*zv + x, r.yi*zv + y);}
			else {z = noise_cell(job, nc, x, y);}
			zvals[i] = finish_cell(z, job, nc, L, d_sm + (size_t)r.xi*zv, d_sm + (size_t)(nux + r.yi)*zv, x, y);
		});
	}
	// mesh shadows of one dependency level: one logical thread per (tile, sweep); smask bits by atomic OR on the containing word, outgoing edge
	// heights by atomic max of (sequential order << 32 | value bits) so that the writer the single-threaded reference would see last wins
	struct shadow_out_t {
		uint8_t *sm; unsigned long long *ox, *oy; int xsize;
		TERRA_HD void shadow(int x, int y) {
			size_t const o = (size_t)y*xsize + x;
#if defined(__HIP_DEVICE_COMPILE__)
			atomicOr((unsigned int *)(sm + (o & ~(size_t)3)), 0x02u << (8u*(unsigned)(o & 3))); // tile bases are multiples of 16900 bytes: word-aligned
#else
			sm[o] |= 0x02;
#endif
		}
		TERRA_HD static unsigned long long pack(uint32_t order, float v) {uint32_t b; memcpy(&b, &v, 4); return ((unsigned long long)order << 32) | b;}
		TERRA_HD void out_x(int ix, uint32_t order, float v) {TERRA_ATOMIC_MAX(&ox[ix], pack(order, v));}
		TERRA_HD void out_y(int iy, uint32_t order, float v) {TERRA_ATOMIC_MAX(&oy[iy], pack(order, v));}
	};
	// incoming edge heights = what the neighbours toward the light left in their ordered out-arrays (0 = never written = MESH_MIN_Z)
	struct shadow_in_t {
		unsigned long long const *ix, *iy; // the y-neighbour's out_x / the x-neighbour's out_y, or nullptr (src/tiled_mesh.cpp:676-687)
		TERRA_HD static float decode(unsigned long long v) {if (v == 0) return -1.0E6f; uint32_t const b = (uint32_t)(v & 0xFFFFFFFFull); float f; memcpy(&f, &b, 4); return f;}
		TERRA_HD float x(int i) const {return ix ? decode(ix[i]) : -1.0E6f;}
		TERRA_HD float y(int i) const {return iy ? decode(iy[i]) : -1.0E6f;}
	};
	void tile_shadows_simple(shadow_consts_t const &c, uint32_t cnt, uint32_t const *d_order, int32_t const *d_adj, uint32_t n, float const *d_zvals,
		unsigned long long *d_out, uint8_t *d_smask, uint32_t npaths)
	{
		unsigned const zv = 130;
		self().launch((size_t)cnt*npaths, [=] TERRA_LAMBDA (size_t i) {
			uint32_t const k = (uint32_t)(i / npaths), p = (uint32_t)(i % npaths), t = d_order[k];
			int32_t const ax = d_adj[2*t], ay = d_adj[2*t + 1];
			shadow_in_t const in{(ay >= 0) ? d_out + ((size_t)0*n + ay)*zv : nullptr, (ax >= 0) ? d_out + ((size_t)1*n + ax)*zv : nullptr};
			shadow_out_t out{d_smask + (size_t)t*zv*zv, d_out + ((size_t)0*n + t)*zv, d_out + ((size_t)1*n + t)*zv, (int)zv};
			shadow_trace_path(c, d_zvals + (size_t)t*zv*zv, in, p, out);
		});
	}
	// AO lighting, simple form: one logical thread per texel, context read from global memory
	// own: inside the tile the context is the tile's own zvals (src/tiled_mesh.cpp:622) -- read from d_zvals, whatever d_ctx holds there
	void tile_ao_simple(uint32_t n, float const *d_zvals, float const *d_ctx, uint8_t *d_ao, float dz, bool own) {
		unsigned const stride = 129, zv = 130, cs = 201, rl = 36;
		self().launch((size_t)n*stride*stride, [=] TERRA_LAMBDA (size_t i) {
			unsigned const t = (unsigned)(i / (stride*stride)), p = (unsigned)(i % (stride*stride)), y = p / stride, x = p % stride;
			float const *c = d_ctx + (size_t)t*cs*cs, *z = d_zvals + (size_t)t*zv*zv;
			d_ao[i] = tile_ao_texel(z[y*zv + x], (int)x, (int)y, dz, [=] TERRA_LAMBDA (int cx, int cy) {
				bool const in = own && (unsigned)(cx - (int)rl) < zv && (unsigned)(cy - (int)rl) < zv;
				return in ? z[(cy - (int)rl)*(int)zv + (cx - (int)rl)] : c[cy*(int)cs + cx];
			});
		});
	}
	// tile post-pass, simple form: sub-block ranges + water bbox (one logical thread per (tile, sub-block) then per tile), normals (one per texel)
	void tile_post_simple(uint32_t n, tile_ref_pod_t const *d_refs, float const *d_zvals, terra_tile_stats *d_stats, uint8_t *d_normals, float *d_min_nz,
		float wpz_max, float rad_c, float dxv, float dyv, float dxy)
	{
		unsigned const size = 128, stride = 129, zv = 130;
		if (d_stats) {
			self().launch((size_t)n*16, [=] TERRA_LAMBDA (size_t i) {
				unsigned const t = (unsigned)(i >> 4), sbk = (unsigned)(i & 15), yy = sbk >> 2, xx = sbk & 3, bs = zv/4;
				float const *z = d_zvals + (size_t)t*zv*zv;
				float szmin = 100.0f, szmax = -100.0f; // FAR_DISTANCE (src/3DWorld.h:116)
				for (unsigned y = yy*bs; y <= (yy+1)*bs; ++y) {
					for (unsigned x = xx*bs; x <= (xx+1)*bs; ++x) {float const v = z[y*zv + x]; szmin = min_std(szmin, v); szmax = max_std(szmax, v);}
				}
				d_stats[t].sub_zmin[sbk] = szmin; d_stats[t].sub_zmax[sbk] = szmax;
			});
			self().launch(n, [=] TERRA_LAMBDA (size_t t) {
				tile_ref_pod_t const r = d_refs[t];
				int const x1 = r.tx*(int)size, y1 = r.ty*(int)size;
				float const *z = d_zvals + (size_t)t*zv*zv;
				terra_tile_stats &st = d_stats[t];
				float mzmin = 100.0f, mzmax = -100.0f;
				for (int sbk = 0; sbk < 16; ++sbk) {mzmin = min_std(mzmin, st.sub_zmin[sbk]); mzmax = max_std(mzmax, st.sub_zmax[sbk]);}
				int wx1 = x1 + (int)size, wy1 = y1 + (int)size, wx2 = x1, wy2 = y1; // start denormalized (src/tiled_mesh.cpp:308)
				unsigned const lim = 4*(zv/4); // cells 0..128 are visited by the 4x4 blocks; row/column 129 is skipped
				for (unsigned y = 0; y <= lim; ++y) {
					for (unsigned x = 0; x <= lim; ++x) {
						if (z[y*zv + x] < wpz_max) {wx1 = imin(wx1, x1+(int)x); wy1 = imin(wy1, y1+(int)y); wx2 = imax(wx2, x1+(int)x); wy2 = imax(wy2, y1+(int)y);}
					}
				}
				st.mzmin = mzmin; st.mzmax = mzmax;
				st.radius = (float)(0.5*sqrt((double)(rad_c + (mzmax - mzmin)*(mzmax - mzmin))));
				st.wx1 = wx1; st.wy1 = wy1; st.wx2 = wx2; st.wy2 = wy2;
			});
		}
		if (d_normals) {
			uint32_t *d_mnz = (uint32_t *)d_min_nz;
			if (d_mnz) {self().fill32(d_mnz, 0x3F800000u /*1.0f*/, n);}
			self().launch((size_t)n*stride*stride, [=] TERRA_LAMBDA (size_t i) {
				unsigned const t = (unsigned)(i / (stride*stride)), p = (unsigned)(i % (stride*stride)), y = p / stride, x = p % stride;
				float nv[3];
				tile_normal(d_zvals + (size_t)t*zv*zv, x, y, dxv, dyv, dxy, nv);
				uint8_t *o = d_normals + i*4;
				o[0] = (uint8_t)(127.0*((double)nv[0] + 1.0)); o[1] = (uint8_t)(127.0*((double)nv[1] + 1.0)); o[2] = (uint8_t)(127.0*((double)nv[2] + 1.0)); o[3] = 0;
				if (d_mnz && nv[2] < 1.0f) { // min_normal_z = min(min_normal_z, norm.z), seeded with 1.0; norm.z = dxdy/mag >= 0 so uint order == float order; NaN never wins
					uint32_t u; memcpy(&u, &nv[2], 4);
					TERRA_ATOMIC_MIN(&d_mnz[t], u);
				}
			});
		}
	}
	// tile erosion, wave form: the clamp-padded copies live in HBM/L2, ONE WAVE per tile walks the droplets in order through a 32x32 LDS window
	// (10 KB of LDS per tile instead of 76 KB: ~15 tiles per CU in flight instead of 2)
	void tile_erosion_windowed(uint32_t n, float *zvals, erosion_consts_t const &ec, uint32_t iters, float *padded /* n*NX*NY */) {
		int const NX = ec.NX, NY = ec.NY, xs = ec.xsize, ys = ec.ysize;
		self().launch((size_t)n*NX*NY, [=] TERRA_LAMBDA (size_t i) { // clamp-padded copy (src/erosion.cpp:31-37)
			unsigned const t = (unsigned)(i / ((size_t)NX*NY)), p = (unsigned)(i % ((size_t)NX*NY));
			int const X = (int)(p % NX), Z = (int)(p / NX);
			padded[i] = zvals[(size_t)t*xs*ys + (size_t)imax(imin(Z - EROSION_PAD, ys-1), 0)*xs + imax(imin(X - EROSION_PAD, xs-1), 0)];
		});
		self().launch_waves(n, [=] TERRA_LAMBDA (size_t t, wave_scratch_t const &ws) {
			window_mem_t<grid_back_t> mem;
			mem.init(ws.win, ws.dirty, NX, NY);
			mem.back.g.interior = padded + t*(size_t)NX*NY; mem.back.g.border = nullptr; mem.back.g.xsize = xs; mem.back.g.ysize = ys; mem.back.g.NX = NX; mem.back.g.NY = NY;
			mem.back.touched = nullptr; mem.back.touched_count = nullptr; mem.back.touched_cap = 0;
			for (uint32_t it = 0; it < iters; ++it) {simulate_droplet((int)it, mem, ec);} // the window carries over from droplet to droplet
			mem.finish();
		});
		self().launch((size_t)n*xs*ys, [=] TERRA_LAMBDA (size_t i) { // unpad + clamp (src/erosion.cpp:158-162)
			unsigned const t = (unsigned)(i / ((size_t)xs*ys)), p = (unsigned)(i % ((size_t)xs*ys));
			int const x = (int)(p % xs), y = (int)(p / xs);
			zvals[i] = max_std(ec.min_zval, padded[(size_t)t*NX*NY + (size_t)(y + EROSION_PAD)*NX + (x + EROSION_PAD)]);
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
	// heightmap_t::from_floats + write_pixel_16_bits (src/heightmap.cpp:205-215, src/Textures.cpp:1889-1893): one logical thread per cell
	void quantize16_simple(float const *vals, size_t n, float val_add, float val_div, uint8_t *pix) {
		self().launch(n, [=] TERRA_LAMBDA (size_t i) {
			float const v = (vals[i] - val_add)*val_div;
			uint8_t const hi = (uint8_t)v;
			pix[(i<<1)+1] = hi;
			pix[i<<1]     = (uint8_t)(256.0f*(v - (float)hi));
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
	void voxel_noise_simple(float *out, size_t nvox, vox_noise_job_t const &J, bool perlin) {
		self().launch(nvox, [=] TERRA_LAMBDA (size_t i) {out[i] = voxel_noise_cell(i, J, perlin);});
	}
	// voxel sine field: val = sum_k xv[k]*yv[k]*zv[k] (src/upsurface.cpp:60-70); d_tab = [nx + ny + nz][60]
	// fused ("gen.fused"): (xv*yv) rounds as in the reference, its multiply-add with zv and the z term round once each (k_sine_grid_mx<SGF_VOXELS> is this, bit for bit)
	void voxel_sines_simple(float *out, uint32_t nx, uint32_t ny, uint32_t nz, float const *d_tab, float zscale, int normalize, int fused = 0) {
		self().launch((size_t)nx*ny*nz, [=] TERRA_LAMBDA (size_t i) {
			unsigned const z = (unsigned)(i % nz), x = (unsigned)((i / nz) % nx), y = (unsigned)(i / ((size_t)nz*nx));
			float const *xv = d_tab + (size_t)x*VOX_SINES, *yv = d_tab + ((size_t)nx + y)*VOX_SINES, *zvp = d_tab + ((size_t)nx + ny + z)*VOX_SINES;
			float val = 0.0f;
			if (fused) {
				for (unsigned k = 0; k < VOX_SINES; ++k) {val = fmaf(xv[k]*yv[k], zvp[k], val);}
				val = fmaf((float)z, zscale, val);
				if (normalize) {val = clip_pm1(val);}
				out[i] = val;
				return;
			}
			for (unsigned k = 0; k < VOX_SINES; ++k) {val += xv[k]*yv[k]*zvp[k];}
			val += (float)z*zscale;
			if (normalize) {val = clip_pm1(val);}
			out[i] = val;
		});
	}
};

} // namespace terra
