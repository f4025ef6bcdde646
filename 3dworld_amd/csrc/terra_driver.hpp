// terra_driver.hpp -- host-side logic of libterra_hip: scene derivation, kernel sequencing, erosion rounds.
//
// Templated on a BACKEND that owns memory and launches "one call per logical thread" bodies:
//   * hip_backend_t (terra_hip.hip)       -- the product: HIP kernels on gfx950, LDS-tiled fast paths for the hot kernels;
//   * cpu_backend_t (tests/emul)          -- TEST ONLY: runs the same bodies in host loops so the sequencing logic
//                                            (scene start-up, speculative erosion rounds, tile batching) is checkable
//                                            against the oracle on machines without a GPU.  Never part of the product.
// Citations are relative to the 3DWorld reference tree.
#pragma once
#include "terra_common.hpp"
#include "terra_noise.hpp"
#include "terra_erosion.hpp"
#include "terra_landscape.hpp"
#include "terra_modmap.hpp"
#include "../../include/terra.h"
#include <vector>
#include <map>
#include <string>
#include <algorithm>
#include <stdexcept>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace terra {

// per-k constants of mesh_xy_grid_cache_t::build_arrays (src/mesh_gen.cpp:604-626), computed on the host in fp32
struct sine_k_t {float xmdx[F_TABLE_SIZE], xconst[F_TABLE_SIZE], ymdy[F_TABLE_SIZE], yconst[F_TABLE_SIZE], yscale[F_TABLE_SIZE];};

struct grid_job_t { // one build_arrays() + eval loop
	float mx0, my0, mdx, mdy;
	uint32_t nx, ny, nxp, nyp; // padded table row lengths
	int mode, shape, kstart, glaciate, use_sine_mag;
	float sine_offset;
	int plain_only; // sine mode: no cell can leave the short epilogue (see terra_engine::sine_plain_only): the kernel variant without finish_cell() is exact
	int fused = 0; // TERRA_GEN_FUSED (tolerance mode; sine mode with plain_only): every multiply-add of the sum and of the tail rounds once (sine_cell_fused / finish_cell_fused); 2 = TERRA_GEN_FAST
	float fast_amax = 0.0f; // TERRA_GEN_FAST: the largest |value| the y table can hold (its power-of-two scale on the half-precision matrix pipe, terra_fused.hpp)
	uint32_t row0 = 0; // the job covers rows [row0, row0 + ny) of a taller grid (row strips of one heightmap on several GPUs): cell row y is eval_index's y + row0
};

// noise_gen_3d constants (src/upsurface.h:10-16)
constexpr unsigned VOX_SINES = 60, VOX_PARAMS = 7;

inline uint32_t round_up(uint32_t v, uint32_t m) {return (v + m - 1)/m*m;}

// float <-> order-preserving uint (for atomic min/max of floats)
TERRA_HD uint32_t f2ord(float f) {uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u);}
TERRA_HD float ord2f(uint32_t o) {uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; memcpy(&f, &u, 4); return f;}

// ---- "simple" (one logical thread per cell) bodies shared by the CPU emulator and the GPU cross-check kernels
struct tile_ref_pod_t {int32_t tx, ty; uint32_t xi, yi;};
// a band of a tile's tw x tw field (tile_fields_dev): twx x twy cells per tile whose coordinate c stands for field coordinate c + (c >= split ? gap : 0); ostride = tw.
// The AO context without its centre: rows [0, 36) + [166, 201) x all 201 columns, then rows [36, 166) x columns [0, 36) + [166, 201)
struct tile_band_t {uint32_t twx, twy, ostride, xsplit, xgap, ysplit, ygap;};
TERRA_HD uint32_t tile_band_coord(uint32_t c, uint32_t split, uint32_t gap) {return c + ((c >= split) ? gap : 0u);}

TERRA_HD float sine_cell(grid_job_t const &job, float const *xt, float const *yt, unsigned x, unsigned y) {
	float z = 0.0f;
	for (int k = job.kstart; k < F_TABLE_SIZE; ++k) {z += xt[(size_t)k*job.nxp + x]*yt[(size_t)k*job.nyp + y];}
	return z;
}
TERRA_HD float finish_cell(float z, grid_job_t const &job, noise_consts_t const &nc, sin_lut_t const &L, float const *smx, float const *smy, unsigned x, unsigned y) {
	if (job.mode == MGEN_SINE) {z = apply_noise_shape_final(z, job.shape, nc.hp);}
	if (job.glaciate) {
		float const xg = ((float)x*job.mdx + job.mx0)*nc.DX_VAL_INV, yg = ((float)(y + job.row0)*job.mdy + job.my0)*nc.DY_VAL_INV;
		z = glaciate_epilogue(z, job.use_sine_mag ? smx[x] : 0.0f, job.use_sine_mag ? smy[y] : 0.0f, job.sine_offset, xg, yg, nc, L);
	}
	return z;
}
// TERRA_GEN_FUSED: the reference's expression tree (src/mesh_gen.cpp:766-790) with every a*b + c contracted into one fused multiply-add -- the sum's 80 terms, glaciate's
// last step, the island term -- and nothing else changed.  Only for jobs whose every cell takes the short epilogue (plain_only): k_sine_grid_mx (terra_fused.hpp) is this
// function on the matrix pipe, bit for bit (the matrix instruction accumulates in k order, one rounding per term).
TERRA_HD float sine_cell_fused(grid_job_t const &job, float const *xt, float const *yt, unsigned x, unsigned y) {
	float z = 0.0f;
	for (int k = job.kstart; k < F_TABLE_SIZE; ++k) {z = fmaf(xt[(size_t)k*job.nxp + x], yt[(size_t)k*job.nyp + y], z);}
	return z;
}
TERRA_HD float finish_cell_fused(float z, grid_job_t const &job, noise_consts_t const &nc, float const *smx, float const *smy, unsigned x, unsigned y) {
	if (job.glaciate) {
		if (nc.glaciate) {float const relh = (z + nc.zmax_est)*nc.zmax_est2_inv; z = fmaf((relh*relh)*relh, nc.zmax_est2, -nc.zmax_est);} // (custom_glaciate_exp == 0: plain_only)
		if (job.use_sine_mag) {z = z + fmaf(smx[x], smy[y], job.sine_offset);}
	}
	return z;
}
TERRA_HD float noise_cell(grid_job_t const &job, noise_consts_t const &nc, unsigned x, unsigned y) {
	float const xval = ((float)x*job.mdx + job.mx0)*nc.DX_VAL_INV, yval = ((float)(y + job.row0)*job.mdy + job.my0)*nc.DY_VAL_INV;
	switch (job.mode) {
	case MGEN_PERLIN:      return noise_zval<MGEN_PERLIN>(xval, yval, job.shape, nc);
	case MGEN_DWARP_GPU:   return noise_zval<MGEN_DWARP_GPU>(xval, yval, job.shape, nc);
	case MGEN_SIMPLEX_GPU: return noise_zval<MGEN_SIMPLEX_GPU>(xval, yval, job.shape, nc);
	default:               return noise_zval<MGEN_SIMPLEX>(xval, yval, job.shape, nc);
	}
}


// ---- row f2: mesh shadows (mesh_shadow_gen::trace_shadow_path, src/visibility.cpp:422-487; do_line_clip src/Math3d.cpp:1029-1086; get_region src/inlines.h:522-528)
struct shadow_consts_t {
	float X_SCENE_SIZE, Y_SCENE_SIZE, DX_VAL, DY_VAL, DX_VAL_INV, DY_VAL_INV, zmin, zmax, dist, dirx, diry, dirz;
	int xsize, ysize;
	uint32_t mask_fill; // what the caller pre-filled the shadow mask with, 4 bytes at a time (0, or MESH_SHADOW everywhere when the light is below the mesh)
	TERRA_HD int xpos(float xval) const {return (int)((double)((xval + X_SCENE_SIZE)*DX_VAL_INV) + 0.5);} // get_xpos (src/mesh.h:129)
	TERRA_HD int ypos(float yval) const {return (int)((double)((yval + Y_SCENE_SIZE)*DY_VAL_INV) + 0.5);}
	TERRA_HD float xval(int xp) const {return -X_SCENE_SIZE + DX_VAL*(float)xp;}                           // get_xval (src/mesh.h:122)
	TERRA_HD float yval(int yp) const {return -Y_SCENE_SIZE + DY_VAL*(float)yp;}
};
struct shadow_pt_t {float x, y, z;};
TERRA_HD int shadow_region(shadow_pt_t v, float const d[3][2]) {
	int region = 0;
	if (v.x < d[0][0]) {region |= 0x01;} else if (v.x >= d[0][1]) {region |= 0x02;}
	if (v.y < d[1][0]) {region |= 0x04;} else if (v.y >= d[1][1]) {region |= 0x08;}
	if (v.z < d[2][0]) {region |= 0x10;} else if (v.z >= d[2][1]) {region |= 0x20;}
	return region;
}
TERRA_HD bool shadow_line_clip(shadow_pt_t &v1, shadow_pt_t &v2, float const d[3][2]) {
	int const region1 = shadow_region(v1, d), region2 = shadow_region(v2, d);
	if (region1 & region2) return false;
	int const region3 = region1 | region2;
	if (region3 == 0) return true;
	float tmin = 0.0f, tmax = 1.0f;
	shadow_pt_t const dv = {v2.x - v1.x, v2.y - v1.y, v2.z - v1.z};
#define TERRA_SH_CLIP(reg, va, vb, vd, vc) if (region3 & (reg)) {float const t = ((va) - (vb))/(vd); if ((vc) > 0.0f) {if (t > tmin) tmin = t;} else {if (t < tmax) tmax = t;} if (tmin >= tmax) return false;}
	TERRA_SH_CLIP(0x01, d[0][0], v1.x, dv.x,  dv.x)
	TERRA_SH_CLIP(0x02, d[0][1], v1.x, dv.x, -dv.x)
	TERRA_SH_CLIP(0x04, d[1][0], v1.y, dv.y,  dv.y)
	TERRA_SH_CLIP(0x08, d[1][1], v1.y, dv.y, -dv.y)
	TERRA_SH_CLIP(0x10, d[2][0], v1.z, dv.z,  dv.z)
	TERRA_SH_CLIP(0x20, d[2][1], v1.z, dv.z, -dv.z)
#undef TERRA_SH_CLIP
	if ((double)tmax > 1.0E-12) {v2.x = v1.x + dv.x*tmax; v2.y = v1.y + dv.y*tmax; v2.z = v1.z + dv.z*tmax;}
	if ((double)tmin < (1.0 - 1.0E-12)) {v1.x += dv.x*tmin; v1.y += dv.y*tmin; v1.z += dv.z*tmin;}
	return true;
}
// Sweep number p in the reference's single-threaded order: p < 2*ysize are run_x's sweeps (y = p), the rest run_y's (x = p - 2*ysize).
// OUT::shadow(x, y) sets the MESH_SHADOW bit; OUT::out_x / out_y(index, order, value) record an outgoing edge height -- `order` grows with the
// sequential execution order (sweep, then step), the writer with the highest order must win.
// IN::x(ix) / IN::y(iy): incoming edge heights (MESH_MIN_Z = none)
template<class IN, class OUT> TERRA_HD void shadow_trace_path(shadow_consts_t const &c, float const *mh, IN const &in, unsigned p, OUT &out) {
	shadow_pt_t v1;
	if (p < 2u*(unsigned)c.ysize) {v1.x = c.xval((c.dirx > 0) ? 0 : c.xsize); v1.y = (float)((double)-c.Y_SCENE_SIZE + 0.5*(double)c.DY_VAL*(double)(int)p); v1.z = 0.0f;}
	else {int const xx = (int)(p - 2u*(unsigned)c.ysize); v1.x = (float)((double)-c.X_SCENE_SIZE + 0.5*(double)c.DX_VAL*(double)xx); v1.y = c.yval((c.diry > 0) ? 0 : c.ysize); v1.z = 0.0f;}
	shadow_pt_t v2 = {v1.x + c.dirx*c.dist, v1.y + c.diry*c.dist, v1.z + 0.0f};
	float const d[3][2] = {{-c.X_SCENE_SIZE, c.xval(c.xsize)}, {-c.Y_SCENE_SIZE, c.yval(c.ysize)}, {c.zmin, c.zmax}};
	if (!shadow_line_clip(v1, v2, d)) return;
	int const xa = c.xpos(v1.x), ya = c.ypos(v1.y), xb = c.xpos(v2.x), yb = c.ypos(v2.y), dx = xb - xa, dy = yb - ya;
	bool const dim = (fabsf(c.dirx) < fabsf(c.diry));
	double const dir_ratio = (double)(c.dirz/(dim ? c.diry : c.dirx));
	bool inited = false;
	// a sweep is a chain of ~2*130 dependent steps: the step is kept short.  Of the current point and of the last unshadowed point `cur` only the coordinate
	// along the dominant light axis and the height enter shadow_z, so only those are carried (same values, same arithmetic as the reference's points)
	float cur_d = 0.0f, cur_z = 0.0f;
	float const org_d = dim ? -c.Y_SCENE_SIZE : -c.X_SCENE_SIZE, step_d = dim ? c.DY_VAL : c.DX_VAL; // get_yval / get_xval
	int x = xa, y = ya, dx1 = 0, dy1 = 0, dx2 = 0, dy2 = 0;
	if (dx < 0) {dx1 = -1; dx2 = -1;} else if (dx > 0) {dx1 = 1; dx2 = 1;}
	if (dy < 0) {dy1 = -1;} else if (dy > 0) {dy1 = 1;}
	int longest = (dx < 0) ? -dx : dx, shortest = (dy < 0) ? -dy : dy;
	if (longest <= shortest) {
		int const tmp = longest; longest = shortest; shortest = tmp;
		if (dy < 0) {dy2 = -1;} else if (dy > 0) {dy2 = 1;}
		dx2 = 0;
	}
	int numerator = longest >> 1;
	for (int i = 0; i <= longest; i++) {
		if ((unsigned)x < (unsigned)c.xsize && (unsigned)y < (unsigned)c.ysize) { // x >= 0 && y >= 0 && x < xsize && y < ysize
			float const pt_d = org_d + step_d*(float)(dim ? y : x), pt_z = mh[y*c.xsize + x];
			float siv;
			if (x == xa && (siv = in.y(y)) > -1.0E6f) {cur_d = pt_d; cur_z = siv; inited = true;} // sh_in_y != NULL && x == xa && sh_in_y[y] > MESH_MIN_Z (src/mesh.h:9)
			else if (y == ya && (siv = in.x(x)) > -1.0E6f) {cur_d = pt_d; cur_z = siv; inited = true;}
			float const shadow_z = (float)((double)(pt_d - cur_d)*dir_ratio + (double)cur_z);
			if (inited && shadow_z > pt_z) {
				out.shadow(x, y);
				uint32_t const order = p*1024u + (uint32_t)i + 1u; // sweeps are at most ~2*130 steps long
				if (x == xb) {out.out_y(y, order, shadow_z);}
				if (y == yb) {out.out_x(x, order, shadow_z);}
			}
			else {cur_d = pt_d; cur_z = pt_z;}
			inited = true;
		}
		numerator += shortest;
		if (numerator >= longest) {numerator -= longest; x += dx1; y += dy1;}
		else {x += dx2; y += dy2;}
	}
}

// A sweep's walk, set up once: end cells, Bresenham steps (the statements of shadow_trace_path above, for the callers that want them apart: the lean sweep of the LDS
// kernels and the host's lane order).  false: the sweep misses the tile.
struct shadow_path_t {int xa, ya, xb, yb, longest, shortest, dx1, dy1, dx2, dy2;};
TERRA_HD bool shadow_path_setup(shadow_consts_t const &c, unsigned p, shadow_path_t &w) {
	shadow_pt_t v1;
	if (p < 2u*(unsigned)c.ysize) {v1.x = c.xval((c.dirx > 0) ? 0 : c.xsize); v1.y = (float)((double)-c.Y_SCENE_SIZE + 0.5*(double)c.DY_VAL*(double)(int)p); v1.z = 0.0f;}
	else {int const xx = (int)(p - 2u*(unsigned)c.ysize); v1.x = (float)((double)-c.X_SCENE_SIZE + 0.5*(double)c.DX_VAL*(double)xx); v1.y = c.yval((c.diry > 0) ? 0 : c.ysize); v1.z = 0.0f;}
	shadow_pt_t v2 = {v1.x + c.dirx*c.dist, v1.y + c.diry*c.dist, v1.z + 0.0f};
	float const d[3][2] = {{-c.X_SCENE_SIZE, c.xval(c.xsize)}, {-c.Y_SCENE_SIZE, c.yval(c.ysize)}, {c.zmin, c.zmax}};
	if (!shadow_line_clip(v1, v2, d)) return false;
	w.xa = c.xpos(v1.x); w.ya = c.ypos(v1.y); w.xb = c.xpos(v2.x); w.yb = c.ypos(v2.y);
	int const dx = w.xb - w.xa, dy = w.yb - w.ya;
	w.dx1 = 0; w.dy1 = 0; w.dx2 = 0; w.dy2 = 0;
	if (dx < 0) {w.dx1 = -1; w.dx2 = -1;} else if (dx > 0) {w.dx1 = 1; w.dx2 = 1;}
	if (dy < 0) {w.dy1 = -1;} else if (dy > 0) {w.dy1 = 1;}
	w.longest = (dx < 0) ? -dx : dx; w.shortest = (dy < 0) ? -dy : dy;
	if (w.longest <= w.shortest) {
		int const tmp = w.longest; w.longest = w.shortest; w.shortest = tmp;
		if (dy < 0) {w.dy2 = -1;} else if (dy > 0) {w.dy2 = 1;}
		w.dx2 = 0;
	}
	return true;
}
// The walk's first and last zone in closed form: steps [0, first_end) are on the walk's first column or row (x == xa || y == ya), steps [last_begin, longest] on its last
// (x == xb || y == yb); the dominant coordinate moves at every step, the other one has moved m(i) = floor(((longest >> 1) + i*shortest)/longest) times when step i is made.
TERRA_HD void shadow_path_zones(int longest, int shortest, int &first_end, int &last_begin) {
	int const n0 = longest >> 1;
	if (longest <= 0 || shortest <= 0) {first_end = longest + 1; last_begin = 0; return;} // the minor coordinate never moves: every step is in both zones
	first_end = (longest - n0 + shortest - 1)/shortest;            // m(i) == 0  <=>  n0 + i*shortest < longest
	last_begin = (shortest*longest - n0 + shortest - 1)/shortest;  // m(i) == shortest  <=>  n0 + i*shortest >= shortest*longest
	if (first_end < 1) {first_end = 1;}
	if (last_begin > longest) {last_begin = longest;}
}
// ... checked against the walk itself for every pair a tile of up to 256 cells can have (once per process, on the host; the kernels that rely on the zones are not used if it fails)
inline bool shadow_path_zones_hold() {
	static int const ok = [] {
		for (int longest = 0; longest <= 257; ++longest) {
			for (int shortest = 0; shortest <= longest; ++shortest) {
				int fe, lb; shadow_path_zones(longest, shortest, fe, lb);
				int numerator = longest >> 1, major = 0, minor = 0;
				for (int i = 0; i <= longest; ++i) {
					bool const on_first = (major == 0 || minor == 0), on_last = (major == longest || minor == shortest);
					if (on_first != (i < fe) || on_last != (i >= lb)) return 0;
					numerator += shortest;
					if (numerator >= longest) {numerator -= longest; ++major; ++minor;} else {++major;}
				}
			}
		}
		return 1;
	}();
	return ok != 0;
}
// Which sweep a lane of the LDS kernels takes.  A tile's sweeps have every length from 0 to the tile's width, a wave costs its longest sweep, and what a tile costs is the
// instructions of the waves that share a SIMD (waves i, i + 4, i + 8 of a workgroup do) -- so the sweeps are sorted by length, cut into waves of 64, and the waves dealt to
// the four SIMDs longest with shortest: chunks 0 1 2 3 | 7 6 5 4 | 8 ...  Any order gives the same bytes (a sweep's writes carry the sweep's number, not the lane's).
// lanes: a multiple of 64; 0xFFFF = an idle lane.  false: more sweeps than lanes.
inline bool shadow_lane_order(shadow_consts_t const &c, uint32_t npaths, uint32_t lanes, uint16_t *lane_path) {
	if (npaths > lanes || npaths >= 0xFFFFu || (lanes & 63u) || c.xsize > 256 || c.ysize > 256 || !shadow_path_zones_hold()) return false;
	std::vector<std::pair<int, uint32_t>> len(npaths);
	for (uint32_t p = 0; p < npaths; ++p) {shadow_path_t w; len[p] = std::make_pair(shadow_path_setup(c, p, w) ? -(w.longest + 1) : 0, p);}
	std::stable_sort(len.begin(), len.end());
	uint32_t const nw = lanes/64;
	for (uint32_t k = 0; k < nw; ++k) { // chunk k -> wave
		uint32_t const round = k/4, pos = k % 4, wave = 4*round + ((round & 1u) ? 3 - pos : pos);
		uint32_t const dst = (wave < nw) ? wave : k; // (a last, partial round keeps its order)
		for (uint32_t l = 0; l < 64; ++l) {uint32_t const j = 64*k + l; lane_path[64*dst + l] = (j < npaths) ? (uint16_t)len[j].second : (uint16_t)0xFFFF;}
	}
	return true;
}

// terrain_hmap_manager_t's sampling of a heightmap texture (src/heightmap.cpp:60-84,310-407; value scaling src/mesh_gen.cpp:120): the image stays where the
// caller put it in HBM (1 byte per pixel, or 2 = {fraction, integer} as written by terra_quantize16_dev / write_pixel_16_bits)
struct hmap_view_t {
	uint8_t const *pix; int width, height, ncolors;
	float mesh_scale, mesh_height_scale, mesh_file_scale, mesh_file_tz, mesh_scale_z_inv;
	TERRA_HD float scale_val(float val) const {float const READ_MESH_H_SCALE = 0.0008f; return (READ_MESH_H_SCALE*mesh_height_scale*mesh_file_scale*val + mesh_file_tz)*mesh_scale_z_inv;}
	TERRA_HD float value(unsigned x, unsigned y) const { // heightmap_t::get_heightmap_value with hmap_filter_width = 0: 0 .. 256
		unsigned const ix = (unsigned)width*y + x;
		if (ncolors == 2) {return (float)((double)pix[ix<<1]/256.0 + (double)pix[(ix<<1)+1]);}
		return (float)pix[ix];
	}
	TERRA_HD void clamp_no_scale(int &x, int &y) const { // TEX_EDGE_MODE = 2 (mirror), allow_wrap: always on the texture afterwards
		x += width/2; y += height/2;
		if (x >= 0 && y >= 0 && x < width && y < height) return;
		int const ax = (x < 0) ? -x : x, ay = (y < 0) ? -y : y; // abs()
		int const xmod = ax % width, ymod = ay % height, xdiv = x/width, ydiv = y/height;
		x = (xdiv & 1) ? (width  - xmod - 1) : xmod;
		y = (ydiv & 1) ? (height - ymod - 1) : ymod;
	}
	TERRA_HD float raw_height(int x, int y) const {return scale_val(value((unsigned)x, (unsigned)y));}
	TERRA_HD float interpolate_height(float x, float y) const { // terrain_hmap_manager_t::interpolate_height (src/heightmap.cpp:394-402), bilinear
		float const sx = mesh_scale*x, sy = mesh_scale*y;
		int xlo = (int)floor((double)sx), ylo = (int)floor((double)sy), xhi = (int)ceil((double)sx), yhi = (int)ceil((double)sy);
		float const xv = sx - (float)xlo, yv = sy - (float)ylo;
		clamp_no_scale(xlo, ylo); clamp_no_scale(xhi, yhi);
		return    yv *(xv*raw_height(xhi, yhi) + (1.0f-xv)*raw_height(xlo, yhi)) +
			(1.0f-yv)*(xv*raw_height(xhi, ylo) + (1.0f-xv)*raw_height(xlo, ylo));
	}
	TERRA_HD static int round_fp(float v) {return (v > 0.0f) ? f2i_x86(v + 0.5f) : f2i_x86(v - 0.5f);} // src/inlines.h:63
	TERRA_HD void clamp_xy(int &x, int &y, float fract_x, float fract_y) const { // src/heightmap.cpp:310-314
		x = round_fp(mesh_scale*((float)x + fract_x));
		y = round_fp(mesh_scale*((float)y + fract_y));
		clamp_no_scale(x, y);
	}
	TERRA_HD float clamped_height(int x, int y) const { // get_clamped_height (src/heightmap.cpp:385-392)
		if (mesh_scale < 1.0f) {return interpolate_height((float)x, (float)y);}
		clamp_xy(x, y, 0.0f, 0.0f);
		return raw_height(x, y);
	}
};

// one texel of tile_t::calc_mesh_ao_lighting (src/tiled_mesh.cpp:634-659): 8 directions x 8 steps at offsets 1,3,6,...,36 cells, the ray rises by
// dz per step; the first context cell above the ray attenuates by (8 - step).  ctx(cx, cy): context value at context coordinates.
template<class CTX> TERRA_HD uint8_t tile_ao_texel(float z_start, int x, int y, float dz, CTX ctx) {
	unsigned atten = 0;
	for (int dy = -1; dy <= 1; ++dy) {
		for (int dx = -1; dx <= 1; ++dx) { // ao_dirs order (src/tiled_mesh.cpp:593-597); the sum does not depend on it
			if (dx == 0 && dy == 0) continue;
			float z0 = z_start;
			int stepx = dx, stepy = dy, vx = x, vy = y;
			for (unsigned s = 0; s < 8; ++s) {
				vx += stepx; vy += stepy;
				z0 += dz;
				stepx += dx; stepy += dy;
				if (ctx(vx + 36, vy + 36) > z0) {atten += (8 - s); break;}
			}
		}
	}
	float const ao_scale = (float)(1.0 - (double)((float)atten/(float)64));
	return (uint8_t)(255.0*(double)ao_scale);
}

// tile_t::get_norm (src/tiled_mesh.h:281-284): n = normalize(DY*(z - z[+1]), DX*(z - z[+zvsize]), dxdy), pointT::get_norm (src/3DWorld.h:297-300, TOLERANCE :50)
// v_rsq_f32 for k_tile_post's byte test (terra_kernels.hpp): the proof there needs |rsq(x)*sqrt(x) - 1| <= 2^-23 for normal x > 0 -- checked over every fp32 input by
// terra_selftest_hot_sqrt.  The host build (tests/emul) never runs that kernel; it gets a value with the same property.
TERRA_HD float rsq_approx(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_rsqf(x);
#else
	return (float)(1.0/sqrt((double)x));
#endif
}
TERRA_HD void tile_normal_v(float zc, float zr, float zd, float dxv, float dyv, float dxy, float nv[3]) { // zc = z[ix], zr = z[ix + 1], zd = z[ix + zvsize]
	nv[0] = dyv*(zc - zr); nv[1] = dxv*(zc - zd); nv[2] = dxy;
	float const mag = sqrtf(nv[0]*nv[0] + nv[1]*nv[1] + nv[2]*nv[2]);
	if (!(mag < 1.0E-12f)) {nv[0] /= mag; nv[1] /= mag; nv[2] /= mag;}
}
TERRA_HD void tile_normal(float const *z, unsigned x, unsigned y, float dxv, float dyv, float dxy, float nv[3]) {
	unsigned const zv = 130, ix2 = y*zv + x;
	tile_normal_v(z[ix2], z[ix2 + 1], z[ix2 + zv], dxv, dyv, dxy, nv);
}

// ---- terra_set_option (include/terra.h): every behaviour switch of the library in one place.  Nothing in the library reads the process environment; tests and tools
// translate their TERRA_* variables into these calls (3dworld_amd/terra.py: options_from_env).  Except "gen.fused", no option changes a result.
struct options_t {
	int gen_fused = 0;            // "gen.fused" 0 / 1 / 2: every generator call behaves as if TERRA_GEN_FUSED (1) / TERRA_GEN_FAST (2) were given (the calls without a flags argument: tiles, voxels)
	int ero_lead = 2;             // "ero.lead" 0..2: where a recentred droplet window lies (a cache placement)
	int ero_batch = 0;            // "ero.batch" >= 1: rounds per host read-back of the multi-version scheduler (0: automatic)
	int ero_fuse = 3;             // "ero.fuse" bits: 1 = a batch of rounds is ONE graph, 2 = marks + unlink + publish in one launch, 4 = commit + resume + hand-over + end of round in one launch
	                              // (results never depend on it).  Measured, same box (profiles/r06_erosion_fuse_ab.txt): 1 and 2 are neutral (-0.7 %), 4 LOSES 9-36 %: not on by default
	int ero_sparse = -1;          // "ero.sparse" 0 / 1: never / always try the sparse scheduler (-1 = "auto": by droplet density)
	int ero_sparse_retraces = -1; // "ero.sparse_retraces" >= 0: re-traces before the sparse scheduler hands over (-1: default 8)
	int ero_live = 1;             // "ero.live" 0 / 1: a droplet's first trace is visible to higher droplets while it grows
	int ero_diag = 0;             // "ero.diag" 0 / 1: per-round clock diagnostics in the erosion report
	int ero_ck_steps = 0, ero_ck_max = -1; // "ero.ck" "steps:max": checkpoint spacing / count (default SPEC_CK_STEPS : SPEC_CK_MAX)
	bool ero_near_set = false; int ero_near = 0; // "ero.near" n (negative: ring / -n): droplets next in line for the commit that trace to the end
	long long ero_mem_budget = -1; // "ero.mem_budget" bytes: pretend this much device memory is free when the ring has to grow (tests)
	int simple_kernels = 0;       // "kernels.simple" 0 / 1: the one-thread-per-cell cross-check kernels instead of the tiled ones
	int ao_bands = 1;             // "ao.bands" 0 / 1: the AO context of a tile batch as four bands around each tile (the centre comes from the tile's own heights anyway) where the backend can (results never depend on it)
	int ao_whole = 1;             // "ao.whole" 0 / 1: the AO rays of a tile from ONE workgroup holding the tile's whole 201 x 201 context in LDS (k_tile_ao_tile); 0: four 33-row bands per tile, each staging its context rows twice (k_tile_ao) (results never depend on it)
	int voxels_cols = 1;          // "voxels.cols" 0 / 1: the lane-per-column voxel sine kernel (no P array) wherever the depth is a multiple of 4; 0: the z-lane kernel over the P stream everywhere (results never depend on it)
	int graphs = 1;               // "graphs" 0 / 1: replay the erosion rounds as hipGraphs
	int sg_kc = 27, sg_kc_tiles = 27; // "sg.kc" 20 / 27 / 45, "sg.kc_tiles" 27 / 45: terms per LDS chunk of k_sine_grid (heightmap / tile batch)
	int sg_rowgroup = 4;          // "sg.rowgroup" 1..1024: tile rows walked together by k_sine_grid
	int tile_erosion_window = 0;  // "tile_erosion" "lds" / "window": tile erosion through the 32 x 32 window over an HBM copy instead of the whole tile in LDS
	int weights_simple = 0;       // "weights.simple" 0 / 1: the per-texel form of the weights-texture pass (cross-check of k_tile_weights)
	int shadows_levels = 0;       // "shadows.levels" 0 / 1: one launch per dependency level instead of the one dataflow launch (cross-check)
	// returns false for an unknown key or a value outside the key's range (nothing is changed then)
	bool set(char const *key, char const *value) {
		if (!key || !value) return false;
		std::string const k(key), v(value);
		char *end = nullptr;
		long long const n = strtoll(value, &end, 10);
		bool const is_int = (end != value && *end == '\0');
		auto const flag = [&](int &dst) {if (!is_int || (n != 0 && n != 1)) return false; dst = (int)n; return true;};
		if (k == "gen.fused") {if (!is_int || n < 0 || n > 2) return false; gen_fused = (int)n; return true;}
		if (k == "ero.lead") {if (!is_int || n < 0 || n > 2) return false; ero_lead = (int)n; return true;}
		if (k == "ero.fuse") {if (!is_int || n < 0 || n > 7) return false; ero_fuse = (int)n; return true;}
		if (k == "ero.batch") {if (!is_int || n < 0 || n > (1 << 20)) return false; ero_batch = (int)n; return true;}
		if (k == "ero.sparse") {if (v == "auto") {ero_sparse = -1; return true;} return flag(ero_sparse);}
		if (k == "ero.sparse_retraces") {if (!is_int || n < -1 || n > (1 << 20)) return false; ero_sparse_retraces = (int)n; return true;}
		if (k == "ero.live") return flag(ero_live);
		if (k == "ero.diag") return flag(ero_diag);
		if (k == "ero.ck") {int a = 0, b = 0; if (v == "default") {ero_ck_steps = 0; ero_ck_max = -1; return true;} if (sscanf(value, "%d:%d", &a, &b) != 2 || a < 1 || b < 0 || b > (int)SPEC_CK_MAX) return false; ero_ck_steps = a; ero_ck_max = b; return true;}
		if (k == "ero.near") {if (v == "default") {ero_near_set = false; return true;} if (!is_int || n < -(1 << 20) || n > (1 << 20)) return false; ero_near_set = true; ero_near = (int)n; return true;}
		if (k == "ero.mem_budget") {if (!is_int || n < -1) return false; ero_mem_budget = n; return true;}
		if (k == "kernels.simple") return flag(simple_kernels);
		if (k == "voxels.cols") return flag(voxels_cols);
		if (k == "ao.bands") return flag(ao_bands);
		if (k == "ao.whole") return flag(ao_whole);
		if (k == "graphs") return flag(graphs);
		if (k == "sg.kc") {if (!is_int || (n != 20 && n != 27 && n != 45)) return false; sg_kc = (int)n; return true;}
		if (k == "sg.kc_tiles") {if (!is_int || (n != 27 && n != 45)) return false; sg_kc_tiles = (int)n; return true;}
		if (k == "sg.rowgroup") {if (!is_int || n < 1 || n > 1024) return false; sg_rowgroup = (int)n; return true;}
		if (k == "tile_erosion") {if (v == "lds") {tile_erosion_window = 0; return true;} if (v == "window") {tile_erosion_window = 1; return true;} return false;}
		if (k == "weights.simple") return flag(weights_simple);
		if (k == "shadows.levels") return flag(shadows_levels);
		return false;
	}
};

// voxel_manager::create_procedural's lattice field (src/voxels.cpp:312-345): what one voxel needs
struct vox_noise_job_t {float l0, l1, l2, v0, v1, v2, o0, o1, o2, frx, fry, mag, freq, zscale; int32_t nn, normalize; uint32_t nx, nz, y0;};
TERRA_HD float voxel_noise_cell(size_t i, vox_noise_job_t const &J, bool perlin) {
	unsigned const z = (unsigned)(i % J.nz), x = (unsigned)((i / J.nz) % J.nx), y = (unsigned)(i / ((size_t)J.nz*J.nx)) + J.y0;
	float const px = ((float)x*J.v0 + J.l0) + J.o0, py = ((float)y*J.v1 + J.l1) + J.o1, pz = ((float)z*J.v2 + J.l2) + J.o2; // get_pt_at (src/voxels.h:146) + offset
	float val = 0.0f, nmag = J.mag, nfreq = (float)(0.25*(double)J.freq);
	for (int n = 0; n < J.nn; ++n) {
		float const ax = nfreq*px + J.frx, ay = nfreq*py + J.fry, az = nfreq*pz + (J.frx - J.fry);
		val += nmag*(perlin ? perlin3(ax, ay, az) : simplex3(ax, ay, az));
		nmag *= 0.5f; nfreq *= 1.92f;
	}
	val += (float)z*J.zscale;
	if (J.normalize) {val = clip_pm1(val);}
	return val;
}

// the same for the voxels (x, y, z) and (x, y, z + 1) of a column, lattice part from the 3-D table (terra_noise.hpp: perlin3_lut_z2 / simplex3_lut): bit-identical to two
// calls of voxel_noise_cell (tests/emul: terra_emul_noise3_lut_mismatches; the GPU parity tests of the fields)
template<bool PERLIN> TERRA_HD nv2 voxel_noise_pair(unsigned x, unsigned y, unsigned z, vox_noise_job_t const &J, char const *tab) {
	float const px = ((float)x*J.v0 + J.l0) + J.o0, py = ((float)y*J.v1 + J.l1) + J.o1;
	nv2 const zf = {(float)z, (float)(z + 1)};
	nv2 const pz = (zf*J.v2 + J.l2) + J.o2;
	nv2 val = {0.0f, 0.0f};
	float nmag = J.mag, nfreq = (float)(0.25*(double)J.freq);
	float const frz = J.frx - J.fry;
	for (int n = 0; n < J.nn; ++n) {
		float const ax = nfreq*px + J.frx, ay = nfreq*py + J.fry;
		nv2 const az = nfreq*pz + frz;
		nv2 const v = PERLIN ? perlin3_lut_z2<true>(ax, ay, az, tab) : simplex3_lut<true>(nv2{ax, ax}, nv2{ay, ay}, az, tab);
		val += nmag*v;
		nmag *= 0.5f; nfreq *= 1.92f;
	}
	val += zf*J.zscale;
	if (J.normalize) {val = nv2{clip_pm1(val[0]), clip_pm1(val[1])};}
	return val;
}

template<class BE> struct terra_engine {
	BE be;
	options_t opt;
	terra_engine() {be.opt = &opt;}
	void set_option(char const *key, char const *value) {
		options_t o = opt;
		if (!o.set(key, value)) throw std::invalid_argument(std::string("terra_set_option: unknown key or bad value: ") + (key ? key : "(null)") + " = " + (value ? value : "(null)"));
		be.sync(); // (kernels in flight were launched under the old options)
		opt = o;
		be.options_changed();
	}
	terra_config cfg{};
	// ---- derived globals (the reference's process globals for this path)
	float MESH_HEIGHT = 0, XY_SCENE_SIZE = 0, DX_VAL = 0, DY_VAL = 0, HALF_DXY = 0, DX_VAL_INV = 0, DY_VAL_INV = 0, dxdy = 0;
	float sinTable[F_TABLE_SIZE][5] = {};
	int   start_eval_sin = 0, mode = 0, shape = 0, glaciate = 1;
	float mesh_scale = 1, mesh_scale_z_inv = 1, mesh_height_scale = 1;
	float zmin = 0, zmax = 0, zmax_est = 0, zmax_est2 = 1, zmax_est2_inv = 1, water_plane_z = 0, glaciate_exp = 1, clip_hd1 = 0;
	float relh_adj_tex = 0, erode_amount = 1, custom_glaciate_exp = 0, water_h_off = 0, water_h_off_rel = 0, ocean_wave_height = 0;
	float rx = 1, ry = 1, two_pi = 0, sscale = 0;
	hmap_params_t hp{};
	rand_gen_t sine_rgen{1, 1}; // the function-static rgen of gen_rand_sine_table_entries (src/mesh_gen.cpp:239)
	bool scene_ready = false, have_config = false;
	std::vector<float> h_sin_table;
	float *d_sin_table = nullptr;
	float *d_sinTable = nullptr; bool sinTable_dev_valid = false; // device copy of sinTable[90][5]: the per-grid constants of build_arrays are derived on the device (no upload per call)
	uint32_t *d_noise3_lut = nullptr; // lattice tables of the 3-D fields (noise3_lut_fill): simplex, then Perlin
	uint32_t *d_noise_lut = nullptr; // lattice tables of the fBm kernels (terra_noise.hpp: noise_lut_fill), built once per context on the device
	terra_erosion_report report{};

	// grow-only device scratch
	struct scratch_t {void *p = nullptr; size_t bytes = 0;};
	scratch_t s_xt, s_yt, s_smx, s_smy, s_misc, s_border, s_spec, s_spec_blocks, s_tiles, s_ao, s_shadow, s_shadow_map, s_shadow_gather, s_vox, s_sk, s_mm, s_hostgrid;
	bool tiled_mesh_ao = false; // enable_tiled_mesh_ao (src/3DWorld.cpp:73,1778)
	uint8_t const *hmap_pix = nullptr; int hmap_w = 0, hmap_h = 0, hmap_nc = 0; // terrain_hmap_manager's image (device memory, owned by the caller)
	float mesh_file_scale = 1.0f, mesh_file_tz = 0.0f;                          // src/mesh_gen.cpp:41, set by set_mesh_height_scales_for_zval_range
	uint32_t *spec_blocks_clean = nullptr; size_t spec_blocks_n = 0; // s_spec_blocks is known to be all-NIL for this pointer / block count
	template<class T> T *scratch(scratch_t &s, size_t count) {
		size_t const bytes = std::max<size_t>(count*sizeof(T), 256);
		if (bytes > s.bytes) {if (s.p) {be.sync(); be.free(s.p); s.p = nullptr; s.bytes = 0;} s.p = be.alloc(bytes); s.bytes = bytes;} // (a failed allocation leaves an empty buffer, not a stale pointer)
		return (T *)s.p;
	}
	float *host_grid_scratch(size_t bytes) {return (float *)scratch<uint8_t>(s_hostgrid, bytes);} // device copy of a caller's host array (the host-pointer entry points)
	// every grow-only device buffer of the context back to the allocator (they grow again on demand): the erosion ring of a 16384^2 map alone is ~8.5 GiB
	void release_scratch() {
		be.sync();
		for (scratch_t *s : {&s_xt, &s_yt, &s_smx, &s_smy, &s_misc, &s_border, &s_spec, &s_spec_blocks, &s_tiles, &s_ao, &s_shadow, &s_shadow_map, &s_shadow_gather, &s_vox, &s_sk, &s_mm, &s_hostgrid}) {if (s->p) {be.free(s->p); s->p = nullptr; s->bytes = 0;}}
		spec_blocks_clean = nullptr; spec_blocks_n = 0;
		be.release_scratch();
	}
	~terra_engine() {
		for (scratch_t *s : {&s_xt, &s_yt, &s_smx, &s_smy, &s_misc, &s_border, &s_spec, &s_spec_blocks, &s_tiles, &s_ao, &s_shadow, &s_shadow_map, &s_shadow_gather, &s_vox, &s_sk, &s_mm, &s_hostgrid}) {if (s->p) be.free(s->p);}
		if (d_sin_table) be.free(d_sin_table);
		if (d_noise_lut) be.free(d_noise_lut);
		if (d_noise3_lut) be.free(d_noise3_lut);
		if (d_sinTable) be.free(d_sinTable);
	}

	sin_lut_t lut() const {return sin_lut_t{d_sin_table, sscale};}
	sin_lut_t host_lut() const {return sin_lut_t{h_sin_table.data(), sscale};}
	noise_consts_t consts() const {
		noise_consts_t nc;
		nc.hp = hp; nc.mesh_scale = mesh_scale; nc.mesh_scale_z_inv = mesh_scale_z_inv; nc.DX_VAL_INV = DX_VAL_INV; nc.DY_VAL_INV = DY_VAL_INV;
		nc.MESH_HEIGHT = MESH_HEIGHT; nc.mesh_height_scale = mesh_height_scale; nc.zmax_est = zmax_est; nc.zmax_est2 = zmax_est2; nc.zmax_est2_inv = zmax_est2_inv;
		nc.custom_glaciate_exp = custom_glaciate_exp; nc.rx = rx; nc.ry = ry; nc.start_eval_sin = start_eval_sin; nc.glaciate = glaciate;
		return nc;
	}

	// ================================================================ scene start-up (a1, a3, a9)
	void create_sin_table() { // src/mesh_gen.cpp:72-81 (host libm, exactly like the reference), uploaded once
		if (!h_sin_table.empty()) return;
		two_pi = (float)(2.0*(double)PI_F);
		sscale = (float)TSIZE/two_pi;
		h_sin_table.resize(2*TSIZE);
		for (unsigned i = 0; i < (unsigned)TSIZE; ++i) {
			h_sin_table[i]       = sinf((float)i/sscale);
			h_sin_table[i+TSIZE] = cosf((float)i/sscale);
		}
		d_sin_table = (float *)be.alloc(2*TSIZE*sizeof(float));
		be.h2d(d_sin_table, h_sin_table.data(), 2*TSIZE*sizeof(float));
		uint32_t *nl = d_noise_lut = (uint32_t *)be.alloc(NOISE_LUT_DWORDS*sizeof(uint32_t));
		be.launch(NOISE_LUT_DWORDS, [=] TERRA_LAMBDA (size_t i) {nl[i] = noise_lut_fill((unsigned)i);}); // the per-cell code itself fills the table: same bits by construction
		uint32_t *n3 = d_noise3_lut = (uint32_t *)be.alloc(2*NOISE3_LUT_DWORDS*sizeof(uint32_t)); // [0]: simplex(vec3), [1]: perlin(vec3)
		be.launch(2*NOISE3_LUT_DWORDS, [=] TERRA_LAMBDA (size_t i) {n3[i] = noise3_lut_fill((unsigned)(i % NOISE3_LUT_DWORDS), i >= NOISE3_LUT_DWORDS);});
	}
	void set_scene_constants() { // src/matrix_ops.cpp:59-84
		MESH_HEIGHT   = 0.10f*cfg.scene_z;
		XY_SCENE_SIZE = 0.5f*(cfg.scene_x + cfg.scene_y);
		DX_VAL        = (2.0f*cfg.scene_x)/(float)cfg.mesh_x;
		DY_VAL        = (2.0f*cfg.scene_y)/(float)cfg.mesh_y;
		HALF_DXY      = 0.5f*(DX_VAL + DY_VAL);
		DX_VAL_INV    = 1.0f/DX_VAL;
		DY_VAL_INV    = 1.0f/DY_VAL;
		dxdy          = DX_VAL*DY_VAL;
	}
	void apply_mesh_rand_seed(rand_gen_t &r) const { // src/mesh_gen.cpp:213-216 (mesh_rgen_index = 0)
		if (cfg.mesh_seed != 0) {r.set_state(cfg.mesh_seed, 12345);}
		else if (mode != MGEN_SINE) {r.set_state(0+1, 12345);}
	}
	float const *sinTable_dev() { // uploaded when it changed (scene start-up / terra_set_state), not per grid
		if (!d_sinTable) {d_sinTable = (float *)be.alloc(sizeof(sinTable));}
		if (!sinTable_dev_valid) {be.h2d(d_sinTable, &sinTable[0][0], sizeof(sinTable)); sinTable_dev_valid = true;}
		return d_sinTable;
	}
	void gen_rand_sine_table_entries(float scaled_height) { // src/mesh_gen.cpp:219-254
		sinTable_dev_valid = false;
		float xf_scale = (float)cfg.mesh_y/(float)cfg.mesh_x, yf_scale = (float)(1.0/(double)xf_scale);
		if (cfg.scene_x > cfg.scene_y) yf_scale *= cfg.scene_y/cfg.scene_x;
		if (cfg.scene_y > cfg.scene_x) xf_scale *= cfg.scene_x/cfg.scene_y;
		float mags[NUM_FREQ_COMP], freqs[NUM_FREQ_COMP];
		freqs[0] = cfg.start_freq; mags[0] = cfg.start_mag;
		for (int i = 1; i < NUM_FREQ_COMP; ++i) {freqs[i] = freqs[i-1]*cfg.freq_mult; mags[i] = mags[i-1]*cfg.mag_mult;}
		float const mesh_h = (float)((double)scaled_height/sqrt(0.1*N_RAND_SIN2));
		apply_mesh_rand_seed(sine_rgen);
		for (int l = 0; l < NUM_FREQ_COMP; ++l) {
			float const x_freq = freqs[l]/((float)cfg.mesh_x), y_freq = freqs[l]/((float)cfg.mesh_y), mheight = mags[l]*mesh_h;
			for (int i = 0; i < N_RAND_SIN2; ++i) {
				float *st = sinTable[l*N_RAND_SIN2 + i];
				st[0] = sine_rgen.rand_uniform(0.2f, 1.0f)*mheight;  // magnitude
				st[1] = sine_rgen.rand_float()*two_pi;                // y phase
				st[2] = sine_rgen.rand_float()*two_pi;                // x phase
				st[3] = sine_rgen.rand_uniform(0.1f, 1.0f)*x_freq*yf_scale; // y frequency
				st[4] = sine_rgen.rand_uniform(0.1f, 1.0f)*y_freq*xf_scale; // x frequency
			}
		}
	}
	void compute_scale() { // src/mesh_gen.cpp:544-548
		int const iscale = (int)log2f(mesh_scale);
		start_eval_sin = N_RAND_SIN2*imax(0, imin(NUM_FREQ_COMP-3, iscale + cfg.mesh_freq_filter));
	}
	void gen_rx_ry() { // src/mesh_gen.cpp:581-586
		rand_gen_t r{1, 1};
		apply_mesh_rand_seed(r);
		rx = (float)((double)r.rand_float() + 1.0);
		ry = (float)((double)r.rand_float() + 1.0);
	}
	void set_zmax_est(float v) {zmax_est = v; zmax_est2 = (float)(2.0*(double)v); zmax_est2_inv = (float)(1.0/(double)zmax_est2);} // src/mesh_gen.cpp:162-167
	float get_rel_wpz() const {return clip01(0.42f + water_h_off_rel);}               // W_PLANE_Z, src/mesh_gen.cpp:362
	float get_water_z_height() const {                                               // src/mesh_gen.cpp:507-512
		float wpz = get_rel_wpz();
		if (glaciate) {wpz = glaciate_exp_fn(wpz, custom_glaciate_exp);}
		return wpz*zmax_est2 - zmax_est + water_h_off;
	}
	float get_max_sea_level() const {return get_water_z_height() + ocean_wave_height;} // src/tiled_mesh.cpp:141
	void set_zvals() {zmin = -zmax_est; zmax = zmax_est; water_plane_z = get_water_z_height();} // src/mesh_gen.cpp:494-504
	// init_terrain_mesh (src/mesh_gen.cpp:407-431) + gen_tex_height_tables (src/Textures.cpp:1757-1761): the landscape texture heights {sand, dirt, grass, rock, snow}
	void tex_heights(float h_dirt[5]) const {
		static float const mesh_rh_dirt[5] = {0.40f, 0.44f, 0.60f, 0.75f, 1.0f}; // src/mesh_gen.cpp:43
		float const rel_wpz = get_rel_wpz(), W_PLANE_Z = 0.42f;
		for (int i = 0; i < 5; ++i) {
			float const def_h = mesh_rh_dirt[i];
			float h;
			if (def_h < W_PLANE_Z) {h = def_h*rel_wpz/W_PLANE_Z;}
			else {
				float const rel_h = (def_h - W_PLANE_Z)/(1.0f - W_PLANE_Z); h = (float)((double)rel_wpz + (double)rel_h*(1.0 - (double)rel_wpz));
				if (i == LT_SNOW) { // snow can't get lower when water lowers; less snow with increasing temperature
					h = min_std(h, def_h);
					if ((double)ls.temperature > 40.0) {h = (float)((double)h + 0.01*((double)ls.temperature - 40.0));}
				}
			}
			h_dirt[i] = powf(h, glaciate_exp);
		}
	}
	void gen_tex_height_tables() {
		float h_dirt[5];
		tex_heights(h_dirt);
		clip_hd1 = (float)(0.90*(double)h_dirt[1] + 0.10*(double)h_dirt[0]);
	}
	terra_landscape ls = {1.0f, 20.0f, 0.0f, 1.0f, 0, 0, 1, 0u, 16u}; // the reference's defaults (src/3DWorld.cpp:109, src/3DWorld.h:87, src/grass.cpp:14, src/tiled_mesh.h:21)
	void set_landscape(terra_landscape const &p) {
		if (p.num_rnd_grass_blocks == 0) throw std::invalid_argument("terra_set_landscape: num_rnd_grass_blocks must be > 0");
		if (!(p.mesh_scale_z > 0.0f)) throw std::invalid_argument("terra_set_landscape: mesh_scale_z must be > 0");
		ls = p;
	}

	// the config-file values only (no derivation): what an engine that already owns the derived globals passes before terra_set_state
	void set_config(terra_config const &c) {
		if (c.mesh_x <= 0 || c.mesh_y <= 0 || !(c.scene_x > 0) || !(c.scene_y > 0) || !(c.mesh_scale > 0)) throw std::invalid_argument("terra config: bad mesh/scene size");
		if (c.mesh_gen_mode < 0 || c.mesh_gen_mode > MGEN_DWARP_GPU || c.mesh_gen_shape < 0 || c.mesh_gen_shape > 2) throw std::invalid_argument("terra config: bad mesh_gen_mode/shape");
		cfg = c;
		create_sin_table();
		mesh_height_scale = c.mesh_height; mesh_scale = c.mesh_scale; mesh_scale_z_inv = 1.0f; // config-file mesh_scale leaves mesh_scale_z at 1 (src/mesh_gen.cpp:862-874 only runs on runtime rescale)
		mode = c.mesh_gen_mode; shape = c.mesh_gen_shape; glaciate = c.glaciate; custom_glaciate_exp = c.custom_glaciate_exp;
		memcpy(&hp, c.hmap, sizeof(hp));
		erode_amount = c.erode_amount; water_h_off = c.water_h_off; water_h_off_rel = c.water_h_off_rel; relh_adj_tex = c.relh_adj_tex; ocean_wave_height = c.ocean_wave_height;
		have_config = true;
	}
	void init_scene(terra_config const &c) {
		set_config(c);
		set_scene_constants();
		// gen_mesh(0, 0, 1) at start-up (src/mesh_gen.cpp:257-356)
		compute_scale();
		gen_rand_sine_table_entries(MESH_HEIGHT*mesh_height_scale);
		gen_rx_ry();
		scene_ready = true;
		uint32_t const MX = c.mesh_x, MY = c.mesh_y;
		std::vector<float> h((size_t)std::max<uint32_t>(MX*MY, 128*128));
		float *d = scratch<float>(s_misc, h.size());
		gen_grid_dev((float)(0 - (int)MX/2), (float)(0 - (int)MY/2), DX_VAL, DY_VAL, MX, MY, 0, 0, d); // gen_mesh_sine_table (src/mesh_gen.cpp:201-210)
		be.d2h(h.data(), d, (size_t)MX*MY*sizeof(float));
		zmin = zmax = h[0]; // calc_zminmax
		for (size_t i = 0; i < (size_t)MX*MY; ++i) {zmin = min_std(zmin, h[i]); zmax = max_std(zmax, h[i]);}
		// estimate_zminmax(using_eq=1) (src/mesh_gen.cpp:447-485)
		set_zmax_est(max_std(zmax, -zmin));
		if (zmax == zmin) {set_zmax_est((float)((double)zmax_est + 1.0E-6));}
		else {
			float const rm_scale = (float)(1000.0*(double)XY_SCENE_SIZE/(double)mesh_scale);
			gen_grid_dev(0.0f, 0.0f, rm_scale, rm_scale, 128, 128, 0, 0, d);
			be.d2h(h.data(), d, 128*128*sizeof(float));
			float ze = zmax_est;
			for (size_t i = 0; i < 128*128; ++i) {ze = max_std(ze, fabsf(h[i]));}
			if (mode != MGEN_SINE) {ze = (float)((double)ze*1.2);}
			set_zmax_est((float)(1.1*(double)ze));
			set_zvals();
		}
		glaciate_exp = glaciate ? ((custom_glaciate_exp == 0.0f) ? 3.0f : custom_glaciate_exp) : 1.0f; // glaciate() / gen_terrain_map (src/mesh_gen.cpp:388-444)
		gen_tex_height_tables();
	}
	void get_state(terra_state &s) const {
		memcpy(s.sinTable, sinTable, sizeof(sinTable));
		s.start_eval_sin = start_eval_sin; s.MESH_HEIGHT = MESH_HEIGHT; s.DX_VAL = DX_VAL; s.DY_VAL = DY_VAL; s.DX_VAL_INV = DX_VAL_INV; s.DY_VAL_INV = DY_VAL_INV;
		s.HALF_DXY = HALF_DXY; s.dxdy = dxdy; s.XY_SCENE_SIZE = XY_SCENE_SIZE; s.mesh_scale = mesh_scale; s.mesh_scale_z_inv = mesh_scale_z_inv; s.mesh_height_scale = mesh_height_scale;
		s.zmax_est = zmax_est; s.zmin = zmin; s.zmax = zmax; s.water_plane_z = water_plane_z; s.glaciate_exp = glaciate_exp; s.clip_hd1 = clip_hd1; s.relh_adj_tex = relh_adj_tex;
		s.rx = rx; s.ry = ry;
	}
	void set_state(terra_state const &s) {
		if (!have_config) throw std::logic_error("terra_set_state: call terra_set_config (or terra_init_scene) first: hmap_params, modes and erosion scalars are not part of terra_state");
		create_sin_table();
		memcpy(sinTable, s.sinTable, sizeof(sinTable)); sinTable_dev_valid = false;
		start_eval_sin = s.start_eval_sin; MESH_HEIGHT = s.MESH_HEIGHT; DX_VAL = s.DX_VAL; DY_VAL = s.DY_VAL; DX_VAL_INV = s.DX_VAL_INV; DY_VAL_INV = s.DY_VAL_INV;
		HALF_DXY = s.HALF_DXY; dxdy = s.dxdy; XY_SCENE_SIZE = s.XY_SCENE_SIZE; mesh_scale = s.mesh_scale; mesh_scale_z_inv = s.mesh_scale_z_inv; mesh_height_scale = s.mesh_height_scale;
		set_zmax_est(s.zmax_est); zmin = s.zmin; zmax = s.zmax; water_plane_z = s.water_plane_z; glaciate_exp = s.glaciate_exp; clip_hd1 = s.clip_hd1; relh_adj_tex = s.relh_adj_tex;
		rx = s.rx; ry = s.ry;
		scene_ready = true;
	}
	void require_scene() const {if (!scene_ready) throw std::logic_error("terra: scene not initialised (call terra_init_scene or terra_set_state first)");}

	// ================================================================ generator (a4, a5, a6)
	sine_k_t make_sine_k(float mx0, float my0, float dx, float dy) const { // src/mesh_gen.cpp:607-613
		sine_k_t sk;
		float const msx = mesh_scale*DX_VAL_INV, msy = mesh_scale*DY_VAL_INV, ms2 = (float)(0.5*(double)mesh_scale);
		for (int k = 0; k < F_TABLE_SIZE; ++k) {
			float const x_mult = msx*sinTable[k][4], y_mult = msy*sinTable[k][3];
			sk.yscale[k] = mesh_scale_z_inv*sinTable[k][0];
			sk.xconst[k] = ms2*sinTable[k][4] + sinTable[k][2] + x_mult*mx0;
			sk.yconst[k] = ms2*sinTable[k][3] + sinTable[k][1] + y_mult*my0;
			sk.xmdx[k] = x_mult*dx; sk.ymdy[k] = y_mult*dy;
		}
		return sk;
	}

	// build_arrays + [enable_glaciate] + eval_index over the whole grid, device resident, async
	// h_minmax (optional): receives {min, max} of the generated grid (NaNs skipped); fused into the grid kernel where the backend can, and SYNCHRONOUS
	// The sine kernel's short epilogue (glaciate + island term) is the whole of eval_index's tail when the shape is linear, there is no
	// crack / volcano and no sum can exceed min(plat_bot, crat_h): |sum| <= sum_k |amplitude_k| because |SINF| <= 1 (+ rounding slack).
	bool sine_plain_only(int shp, int kstart) const {
		if (shp != 0 || hp.crack_lo < hp.crack_hi || (hp.volcano_width > 0.0f && hp.volcano_height > 0.0f) || consts().custom_glaciate_exp != 0.0f) return false;
		double bound = 0.0;
		for (int k = kstart; k < F_TABLE_SIZE; ++k) {bound += std::fabs((double)(mesh_scale_z_inv*sinTable[k][0]));}
		return (double)min_std(hp.plat_bot, hp.crat_h) > bound*1.001 + 1e-3;
	}

	// row0 / nrows (optional): only rows [row0, row0 + nrows) of the nx x ny grid, written to d_out as an nrows x nx array -- every value is the one the
	// full-grid call produces (the tables and cell coordinates use the row's index in the whole grid), so row strips evaluated on different GPUs tile the
	// heightmap exactly (heightmap_t::proc_gen's loop is row-independent, src/heightmap.cpp:139-143)
	// d_minmax (optional): DEVICE float[2] that receives {min, max} without the host ever seeing them (an enqueue-only proc_gen step: terra_apply_erosion_devmin_dev reads it)
	void gen_grid_dev(float x0, float y0, float dx, float dy, uint32_t nx, uint32_t ny, uint32_t flags, int min_start_sin, float *d_out, float *h_minmax = nullptr, uint32_t row0 = 0, uint32_t nrows = 0xFFFFFFFFu, float *d_minmax = nullptr) {
		require_scene();
		if (nx == 0 || ny == 0) throw std::invalid_argument("build_arrays: nx, ny must be > 0"); // assert(nx > 0 && ny > 0), src/mesh_gen.cpp:589
		if (nrows == 0xFFFFFFFFu) {if (row0 != 0) throw std::invalid_argument("gen_grid rows: row0 without a row count"); nrows = ny;}
		if (nrows == 0 || row0 >= ny || nrows > ny - row0) throw std::invalid_argument("gen_grid rows: [row0, row0 + nrows) must be a non-empty range inside the grid");
		grid_job_t job;
		job.mx0 = dx*x0; job.my0 = dy*y0; job.mdx = dx; job.mdy = dy; job.nx = nx; job.ny = nrows; job.row0 = row0; ny = nrows;
		job.nxp = round_up(nx, 128); job.nyp = round_up(ny, 128);
		bool const force_sine = (flags & TERRA_GEN_FORCE_SINE) != 0;
		job.mode = force_sine ? (int)MGEN_SINE : mode; job.shape = force_sine ? 0 : shape;
		job.kstart = imax(start_eval_sin, min_start_sin);
		job.glaciate = (flags & TERRA_GEN_GLACIATE) ? 1 : 0;
		job.use_sine_mag = (job.glaciate && hp.sine_mag > 0.0f) ? 1 : 0;
		job.sine_offset = hp.sine_bias*mesh_scale_z_inv;
		job.plain_only = (job.mode == MGEN_SINE && sine_plain_only(job.shape, job.kstart)) ? 1 : 0;
		job.fused = (((flags & (TERRA_GEN_FUSED | TERRA_GEN_FAST)) || opt.gen_fused) && fused_kernel_exists(job.mode, job.plain_only != 0)) ? (((flags & TERRA_GEN_FAST) || opt.gen_fused == 2) ? 2 : 1) : 0; // a permission, not a command
		job.fast_amax = sine_amp_max(job.kstart);
		noise_consts_t const nc = consts();
		sin_lut_t const L = lut();
		float *smx = scratch<float>(s_smx, job.nxp), *smy = scratch<float>(s_smy, job.nyp);
		uint32_t *d_mm = nullptr;
		bool const sine = (job.mode == MGEN_SINE);
		if (h_minmax || d_minmax || sine) {d_mm = scratch<uint32_t>(s_mm, 2); if (!sine) {be.fill32(d_mm, 0xFFFFFFFFu, 2);}} // (sine mode: reset by the table launch)
		bool fused;
		bool const sm_on = job.use_sine_mag != 0;
		float const sm_scale = hp.sine_mag*mesh_scale_z_inv, sm_freq = mesh_scale*hp.sine_freq, dxi = DX_VAL_INV, dyi = DY_VAL_INV; // enable_glaciate (src/mesh_gen.cpp:640-650)
		if (sm_on && job.mode != MGEN_SINE) { // (sine mode: folded into the table launch below)
			float const mx0 = job.mx0, my0 = job.my0, mdx = dx, mdy = dy;
			uint32_t const nxp = job.nxp, nyp = job.nyp; // zero padding up to the tile grid: the sine kernel reads whole float4 groups
			be.launch((size_t)nxp + nyp, [=] TERRA_LAMBDA (size_t i) {
				if (i < nxp) {smx[i] = (i < nx) ? sm_scale*L.COSF(((float)(unsigned)i*mdx + mx0)*dxi*sm_freq) : 0.0f;}
				else {unsigned const y = (unsigned)(i - nxp); smy[y] = (y < ny) ? L.COSF(((float)(y + row0)*mdy + my0)*dyi*sm_freq) : 0.0f;}
			});
		}
		if (job.mode == MGEN_SINE) {
			// ONE launch builds everything the grid kernel reads: the k-major tables xt[k*nxp + x] = SINF(xmdx*x + x_const), yt[k*nyp + y] = y_scale*SINF(ymdy*y + y_const) (zero
			// padded) and the island tables smx / smy.  The per-k constants of build_arrays (src/mesh_gen.cpp:607-613) are derived by every thread from the device copy of sinTable
			// -- the same fp32 expressions in the same order as make_sine_k, no contraction on either side, so the same bits -- instead of by a 90-thread launch of their own in
			// front of this one (three dependent launches per grid were ~25 us of a 0.88 ms step).
			float const *st = sinTable_dev();
			float const msx = mesh_scale*DX_VAL_INV, msy = mesh_scale*DY_VAL_INV, ms2 = (float)(0.5*(double)mesh_scale), mszi = mesh_scale_z_inv, jmx0 = job.mx0, jmy0 = job.my0, mdx = dx, mdy = dy;
			float *xt = scratch<float>(s_xt, (size_t)F_TABLE_SIZE*job.nxp), *yt = scratch<float>(s_yt, (size_t)F_TABLE_SIZE*job.nyp);
			uint32_t const nxp = job.nxp, nyp = job.nyp;
			size_t const ntab = (size_t)F_TABLE_SIZE*(nxp + nyp);
			uint32_t *const mmz = d_mm;
			be.launch(ntab + (sm_on ? (size_t)nxp + nyp : 0), [=] TERRA_LAMBDA (size_t i) {
				if (i == 0) {mmz[0] = 0xFFFFFFFFu; mmz[1] = 0xFFFFFFFFu;} // the fused min / max of the grid kernel start here (one launch less than a fill of its own)
				if (i < (size_t)F_TABLE_SIZE*nxp) {
					unsigned const k = (unsigned)(i / nxp), x = (unsigned)(i % nxp);
					float const *stk = st + 5*k;
					float const x_mult = msx*stk[4], xconst = ms2*stk[4] + stk[2] + x_mult*jmx0, xmdx = x_mult*mdx;
					xt[i] = (x < nx) ? L.SINF(xmdx*(float)x + xconst) : 0.0f;
				}
				else if (i < ntab) {
					size_t const j = i - (size_t)F_TABLE_SIZE*nxp;
					unsigned const k = (unsigned)(j / nyp), y = (unsigned)(j % nyp);
					float const *stk = st + 5*k;
					float const y_mult = msy*stk[3], yscale = mszi*stk[0], yconst = ms2*stk[3] + stk[1] + y_mult*jmy0, ymdy = y_mult*mdy;
					yt[j] = (y < ny) ? yscale*L.SINF(ymdy*(float)(y + row0) + yconst) : 0.0f;
				}
				else {
					size_t const q = i - ntab;
					if (q < nxp) {smx[q] = (q < nx) ? sm_scale*L.COSF(((float)(unsigned)q*mdx + jmx0)*dxi*sm_freq) : 0.0f;}
					else {unsigned const y = (unsigned)(q - nxp); smy[y] = (y < ny) ? L.COSF(((float)(y + row0)*mdy + jmy0)*dyi*sm_freq) : 0.0f;}
				}
			});
			fused = be.sine_grid(job, nc, L, xt, yt, smx, smy, d_out, (h_minmax || d_minmax) ? d_mm : nullptr);
		}
		else {fused = be.noise_grid(job, nc, L, smx, smy, d_out, (h_minmax || d_minmax) ? d_mm : nullptr, d_noise_lut);}
		if ((h_minmax || d_minmax) && !fused) {be.minmax(d_out, (size_t)nx*ny, d_mm);}
		if (d_minmax) {uint32_t const *mm = d_mm; be.launch(1, [=] TERRA_LAMBDA (size_t) {d_minmax[0] = ord2f(mm[0]); d_minmax[1] = ord2f(~mm[1]);}, 64);}
		if (h_minmax) {
			uint32_t out[2];
			be.d2h(out, d_mm, sizeof(out));
			h_minmax[0] = ord2f(out[0]); h_minmax[1] = ord2f(~out[1]);
		}
	}

	// ================================================================ point query + ground-mode glaciate (a8, a16)
	// eval_mesh_sin_terms (src/mesh_gen.cpp:797-805): scattered point queries (biome parameters, collision height) stay on the host
	float eval_mesh_sin_terms(float xv, float yv) const {
		require_scene();
		sin_lut_t const HL = host_lut();
		float zval = 0.0f;
		for (int k = start_eval_sin; k < F_TABLE_SIZE; ++k) {
			float const *stk = sinTable[k];
			zval += stk[0]*HL.SINF(stk[3]*yv + stk[1])*HL.SINF(stk[4]*xv + stk[2]);
		}
		return zval;
	}
	// ================================================================ all-modes point queries (a8)
	// kind 0: eval_mesh_sin_terms_scaled(x, y, xy_scale) (src/mesh_gen.cpp:807-813; index-space coordinates: the detail noise of heightmap tiles, the density fields of the
	// vegetation callers); kind 1: get_exact_zval(x, y, no_xyoff) (src/mesh_gen.cpp:816-847; world-space: collision / building placement / camera height) in the tiled-terrain
	// world -- index space, scroll offset, then the heightmap texture (+ detail) when one is set, else noise + glaciate + islands / volcano.  Not here: the two branches that
	// only read the caller's own state (the ground-mode mesh_height look-up :821-825; the `texture named but not loaded yet` constant :839-843).
	void eval_points_dev(float const *d_xy, uint32_t n, uint32_t kind, float xy_scale, int no_xyoff, int xoff2, int yoff2, float *d_out) {
		require_scene();
		if (n == 0) return;
		if (kind > 1) throw std::invalid_argument("eval_points: kind must be TERRA_POINTS_SCALED or TERRA_POINTS_EXACT");
		noise_consts_t const nc = consts();
		sin_lut_t const L = lut();
		float const *st = sinTable_dev();
		hmap_view_t const hv = hmap_view();
		bool const hm = using_hmap(), hm_detail = using_hmap_with_detail();
		int const md = mode, shp = shape, k0 = start_eval_sin;
		float const half_x = (float)(cfg.mesh_x >> 1), half_y = (float)(cfg.mesh_y >> 1), xss = cfg.scene_x, yss = cfg.scene_y, dxi = DX_VAL_INV, dyi = DY_VAL_INV, msc = mesh_scale, mszi = mesh_scale_z_inv;
		be.launch(n, [=] TERRA_LAMBDA (size_t i) {
			auto const scaled = [&](float xval, float yval, float xys) -> float { // eval_mesh_sin_terms_scaled
				float const xv = xys*(xval - half_x), yv = xys*(yval - half_y);
				if (md != MGEN_SINE) {
					switch (md) {
					case MGEN_PERLIN:      return noise_zval<MGEN_PERLIN>(xv, yv, shp, nc);
					case MGEN_DWARP_GPU:   return noise_zval<MGEN_DWARP_GPU>(xv, yv, shp, nc);
					case MGEN_SIMPLEX_GPU: return noise_zval<MGEN_SIMPLEX_GPU>(xv, yv, shp, nc);
					default:               return noise_zval<MGEN_SIMPLEX>(xv, yv, shp, nc);
					}
				}
				float const ax = msc*xv, ay = msc*yv;
				float zval = 0.0f; // eval_mesh_sin_terms (src/mesh_gen.cpp:797-805)
				for (int k = k0; k < F_TABLE_SIZE; ++k) {float const *stk = st + 5*k; zval += stk[0]*L.SINF(stk[3]*ay + stk[1])*L.SINF(stk[4]*ax + stk[2]);}
				return apply_noise_shape_final(zval*mszi, shp, nc.hp);
			};
			float const xin = d_xy[2*i], yin = d_xy[2*i + 1];
			if (kind == 0) {d_out[i] = scaled(xin, yin, xy_scale); return;}
			float xval = (float)((double)((xin + xss)*dxi) + 0.5), yval = (float)((double)((yin + yss)*dyi) + 0.5); // real -> index space, `+ 0.5` formed in double (src/mesh_gen.cpp:818-819)
			if (!no_xyoff) {xval += (float)xoff2; yval += (float)yoff2;}
			float zval;
			if (hm) {
				if (!no_xyoff) {xval = (float)((double)xval - 0.5); yval = (float)((double)yval - 0.5);}
				zval = hv.interpolate_height(xval, yval);
				if (hm_detail) {zval += 0.01f*scaled(xval, yval, 16.0f);} // HMAP_DETAIL_MAG, HMAP_DETAIL_SCALE (src/heightmap.h:8-9)
			}
			else {
				zval = scaled(xval, yval, 1.0f);
				if (nc.glaciate) {float const relh = (zval + nc.zmax_est)*nc.zmax_est2_inv; zval = glaciate_exp_fn(relh, nc.custom_glaciate_exp)*nc.zmax_est2 - nc.zmax_est;}
				if (nc.hp.sine_mag > 0.0f) { // apply_mesh_sine (src/mesh_gen.cpp:373-379)
					float const fx = xval - half_x, fy = yval - half_y, freq = nc.mesh_scale*nc.hp.sine_freq;
					zval += (nc.hp.sine_mag*L.COSF(fx*freq)*L.COSF(fy*freq) + nc.hp.sine_bias)*nc.mesh_scale_z_inv;
					if (nc.hp.volcano_width > 0.0f && nc.hp.volcano_height > 0.0f) {zval += volcano_height(fx, fy, nc, L);}
				}
			}
			d_out[i] = zval;
		});
	}
	void eval_points(float const *h_xy, uint32_t n, uint32_t kind, float xy_scale, int no_xyoff, int xoff2, int yoff2, float *h_out) {
		require_scene();
		if (n == 0) return;
		float *d = scratch<float>(s_misc, (size_t)n*3 + 64); // (sinTable_dev / consts use buffers of their own)
		be.h2d(d, h_xy, (size_t)n*2*sizeof(float));
		eval_points_dev(d, n, kind, xy_scale, no_xyoff, xoff2, yoff2, d + (size_t)n*2);
		be.d2h(h_out, d + (size_t)n*2, (size_t)n*sizeof(float));
	}
	// glaciate() (src/mesh_gen.cpp:388-404): in-place apply_glaciate + apply_mesh_sine over a MESH_X x MESH_Y ground mesh; zbottom/ztop = min/max after
	void glaciate_mesh_dev(float *d_mesh, uint32_t nx, uint32_t ny, int xoff2, int yoff2, float *h_zbottom_ztop) {
		require_scene();
		if (nx == 0 || ny == 0) throw std::invalid_argument("glaciate: empty mesh");
		noise_consts_t const nc = consts();
		sin_lut_t const L = lut();
		int const hx = (int)nx/2, hy = (int)ny/2;
		be.launch((size_t)nx*ny, [=] TERRA_LAMBDA (size_t i) {
			unsigned const x = (unsigned)(i % nx), y = (unsigned)(i / nx);
			float z = d_mesh[i];
			if (nc.glaciate) {float const relh = (z + nc.zmax_est)*nc.zmax_est2_inv; z = glaciate_exp_fn(relh, nc.custom_glaciate_exp)*nc.zmax_est2 - nc.zmax_est;}
			if (nc.hp.sine_mag > 0.0f) { // apply_mesh_sine (src/mesh_gen.cpp:373-379), point form
				float const fx = (float)((int)x + xoff2 - hx), fy = (float)((int)y + yoff2 - hy), freq = nc.mesh_scale*nc.hp.sine_freq;
				z += (nc.hp.sine_mag*L.COSF(fx*freq)*L.COSF(fy*freq) + nc.hp.sine_bias)*nc.mesh_scale_z_inv;
				if (nc.hp.volcano_width > 0.0f && nc.hp.volcano_height > 0.0f) {z += volcano_height(fx, fy, nc, L);}
			}
			d_mesh[i] = z;
		});
		if (h_zbottom_ztop) {minmax_dev(d_mesh, (size_t)nx*ny, h_zbottom_ztop[0], h_zbottom_ztop[1]);}
	}

	// ================================================================ reductions / quantise (a12, K10)
	void minmax_dev(float const *d_vals, size_t n, float &mn, float &mx) { // min_eq/max_eq folds (std::min/std::max): NaNs never win a comparison
		uint32_t *d = scratch<uint32_t>(s_misc, 2);
		uint32_t const init[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
		be.h2d(d, init, sizeof(init));
		be.minmax(d_vals, n, d); // d[0] = min of f2ord(v), d[1] = min of ~f2ord(v)
		uint32_t out[2];
		be.d2h(out, d, sizeof(out));
		mn = ord2f(out[0]); mx = ord2f(~out[1]);
	}
	// heightmap_t::from_floats + write_pixel_16_bits (src/heightmap.cpp:205-215, src/Textures.cpp:1889-1893) with the scale of
	// set_mesh_height_scales_for_zval_range(min_z, dz/255) (src/mesh_gen.cpp:125-131)
	void quantize16_dev(float const *d_vals, size_t n, float min_z, float dz, uint8_t *d_pix) {
		float const READ_MESH_H_SCALE = 0.0008f;
		float const dzs = (float)((double)dz/255.0);
		float const file_scale = dzs/(READ_MESH_H_SCALE*mesh_height_scale*mesh_scale_z_inv), file_tz = min_z/mesh_scale_z_inv;
		float const mult = READ_MESH_H_SCALE*mesh_height_scale*file_scale*mesh_scale_z_inv, add = file_tz*mesh_scale_z_inv;
		float const val_div = (float)(1.0/(double)mult), val_add = add;
		be.quantize16(d_vals, n, val_add, val_div, d_pix);
	}

	// ---- the loaded-heightmap path: heightmap_t::to_floats / from_floats / postprocess_height (src/heightmap.cpp:117-128,191-215) with the pixel <-> height
	// scale of get_mh_texture_mult() / get_mh_texture_add() (src/mesh_gen.cpp:122-123): mesh_file_scale / mesh_file_tz are the two numbers after the file name
	// in the config line `mh_filename <png> <scale> <tz>` (src/3DWorld.cpp:2205), or what set_mesh_height_scales_for_zval_range left behind
	float mh_texture_mult() const {float const READ_MESH_H_SCALE = 0.0008f; return READ_MESH_H_SCALE*mesh_height_scale*mesh_file_scale*mesh_scale_z_inv;}
	float mh_texture_add() const {return mesh_file_tz*mesh_scale_z_inv;}
	void heightmap_to_floats_dev(uint8_t const *d_pix, size_t n, int ncolors, float *d_vals) {
		require_scene();
		if (ncolors != 1 && ncolors != 2) throw std::invalid_argument("heightmap to_floats: one or two byte grayscale only"); // assert(ncolors == 1 || ncolors == 2), src/heightmap.cpp:122
		if (n == 0) throw std::invalid_argument("heightmap to_floats: empty image");                                         // assert(!vals.empty()), :194
		float const val_mult = mh_texture_mult(), val_add = mh_texture_add();
		// HBM bound (1-2 B read, 4 B written per pixel): four pixels per thread, one 4- / 8-byte load and one 16-byte store where the pointers allow it
		bool const wide = (((uintptr_t)d_pix & 7u) == 0) && (((uintptr_t)d_vals & 15u) == 0);
		size_t const n4 = wide ? n/4 : 0;
		if (n4) {
			be.launch(n4, [=] TERRA_LAMBDA (size_t q) {
				float o[4];
				if (ncolors == 2) {
					uint64_t w; memcpy(&w, d_pix + q*8, 8);
					for (int k = 0; k < 4; ++k) {unsigned const lo = (unsigned)(w >> (16*k)) & 255u, hi = (unsigned)(w >> (16*k + 8)) & 255u; o[k] = val_mult*(float)((double)lo/256.0 + (double)hi) + val_add;}
				}
				else {
					uint32_t w; memcpy(&w, d_pix + q*4, 4);
					for (int k = 0; k < 4; ++k) {o[k] = val_mult*(float)((w >> (8*k)) & 255u) + val_add;}
				}
				memcpy(d_vals + q*4, o, 16);
			});
		}
		size_t const done = n4*4;
		be.launch(n - done, [=] TERRA_LAMBDA (size_t j) {
			size_t const i = done + j;
			float v = (ncolors == 2) ? (float)((double)d_pix[i<<1]/256.0 + (double)d_pix[(i<<1)+1]) : (float)d_pix[i];
			d_vals[i] = val_mult*v + val_add;
		});
	}
	// from_floats with the scale in force (NOT proc_gen's rescale to the value range); returns how many values fell outside [0, 256), where the reference asserts (:210)
	uint32_t heightmap_from_floats_dev(float const *d_vals, size_t n, int ncolors, uint8_t *d_pix) {
		require_scene();
		if (ncolors != 1 && ncolors != 2) throw std::invalid_argument("heightmap from_floats: one or two byte grayscale only");
		float const val_div = (float)(1.0/(double)mh_texture_mult()), val_add = mh_texture_add();
		uint32_t *d_bad = scratch<uint32_t>(s_mm, 2);
		be.fill32(d_bad, 0, 1);
		be.launch((n + 3)/4, [=] TERRA_LAMBDA (size_t q) {
			size_t const b = q*4, e = (b + 4 < n) ? b + 4 : n;
			uint32_t bad = 0;
			for (size_t i = b; i < e; ++i) {
				float const v = (d_vals[i] - val_add)*val_div;
				if (!(v >= 0.0f && v < 256.0f)) {++bad;}
			}
			if (bad) {TERRA_ATOMIC_ADD(d_bad, bad);}
		});
		if (ncolors == 2) {be.quantize16(d_vals, n, val_add, val_div, d_pix);}
		else {
			be.launch((n + 3)/4, [=] TERRA_LAMBDA (size_t q) {
				size_t const b = q*4, e = (b + 4 < n) ? b + 4 : n;
				for (size_t i = b; i < e; ++i) {d_pix[i] = (uint8_t)f2i_x86((d_vals[i] - val_add)*val_div);} // data[i] = (unsigned char)v
			});
		}
		uint32_t bad = 0;
		be.d2h(&bad, d_bad, 4);
		return bad;
	}
	// heightmap_t::postprocess_height (src/heightmap.cpp:117-128), called right after the image was loaded (terrain_hmap_manager_t::load, :351): pixels -> floats ->
	// run_erosion (min_zval = min(vals), erosion_iters_tt droplets over the whole image) -> [run_city_gen: other subsystem] -> pixels, in place.
	// d_vals: width*height floats of device scratch supplied by the caller, left holding the eroded heights.
	uint32_t heightmap_postprocess_dev(uint8_t *d_pix, uint32_t width, uint32_t height, int ncolors, uint32_t iters_tt, float *d_vals) {
		require_scene();
		if (iters_tt == 0) return 0; // "no erosion or cities => no need to update height values"
		size_t const n = (size_t)width*height;
		heightmap_to_floats_dev(d_pix, n, ncolors, d_vals);
		if (erode_amount > 0.0f) { // (apply_erosion's own early-out, src/erosion.cpp:16)
			float mn, mx; minmax_dev(d_vals, n, mn, mx);
			apply_erosion_dev(d_vals, (int)width, (int)height, mn, iters_tt, TERRA_ERODE_MINZ_IS_MIN);
		}
		return heightmap_from_floats_dev(d_vals, n, ncolors, d_pix);
	}

	// Self test of the droplet step's square roots (terra_erosion.hpp: sqrt_rn, sqrt_rn_direction) over every stride-th fp32 bit pattern: against the compiler's full sqrtf
	// expansion and against the double-precision route (float)sqrt((double)x), which is correctly rounded (53 >= 2*24 + 2 bits); and of the error bound k_tile_post assumes
	// for v_rsq_f32.  Returns the number of disagreements.
	uint64_t selftest_hot_sqrt(uint32_t stride) {
		if (stride == 0) {stride = 1;}
		// 64-bit counter: up to 4 disagreements per input x 2^32 inputs would wrap a 32-bit one (a sqrt_rn that is wrong everywhere would add up to 0 mod 2^32)
		unsigned long long *d_bad = (unsigned long long *)scratch<uint32_t>(s_mm, 2);
		be.fill32(d_bad, 0, 2);
		uint64_t const n = (0x100000000ull + stride - 1)/stride, per = 4096, nthreads = (n + per - 1)/per;
		be.launch((size_t)nthreads, [=] TERRA_LAMBDA (size_t t) {
			auto same = [](float a, float b) {uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4); return ua == ub || (a != a && b != b);};
			uint32_t bad = 0;
			uint64_t const k1 = ((t + 1)*per < n) ? (t + 1)*per : n;
			for (uint64_t k = t*per; k < k1; ++k) {
				uint32_t const bits = (uint32_t)(k*stride);
				float x; memcpy(&x, &bits, 4);
				float const a = sqrt_rn(x), b = sqrtf(x), c = (float)sqrt((double)x);
				bad += (same(a, b) ? 0u : 1u) + (same(a, c) ? 0u : 1u);
				if (!(x < 0.0f)) { // a sum of squares (or a NaN): the direction root only has to agree where it passes the FLT_EPSILON test, and on the test itself
					float const r = sqrt_rn_direction(x);
					bool const pr = r > FLT_EPSILON, pb = b > FLT_EPSILON;
					bad += (pr != pb) ? 1u : 0u;
					if (pr && !same(r, b)) {++bad;}
				}
				if (x >= 0x1p-126f && x < INFINITY) { // k_tile_post: the reciprocal square root it decides a normal's bytes with is within 2^-23 (relative) of the real one
					double const rel = (double)rsq_approx(x)*sqrt((double)x) - 1.0;
					bad += (rel > 0x1p-23 || rel < -0x1p-23) ? 1u : 0u;
				}
			}
			if (bad) {TERRA_ATOMIC_ADD(d_bad, (unsigned long long)bad);}
		});
		unsigned long long bad = 0;
		be.d2h(&bad, d_bad, 8);
		return (uint64_t)bad;
	}

	// ================================================================ erosion (a11)
	erosion_consts_t make_erosion_consts(int xsize, int ysize, float min_zval) const {
		erosion_consts_t ec;
		ec.xsize = xsize; ec.ysize = ysize; ec.NX = xsize + 2*EROSION_PAD; ec.NY = ysize + 2*EROSION_PAD;
		ec.max_path_len = 4u*(unsigned)ec.NX*(unsigned)ec.NY;
		ec.erode_amount = erode_amount; ec.water_thresh = water_plane_z - HALF_DXY;
		ec.relh_adj_tex = relh_adj_tex; ec.zmin = zmin; ec.zrange = zmax - zmin; ec.clip_hd1 = clip_hd1; ec.two_pi = two_pi; ec.min_zval = min_zval;
		make_rock_threshold(ec);
		ec.lead_mode = opt.ero_lead; // where a recentred window lies never changes a result (it is a cache)
		return ec;
	}

	// the reference seeds droplet `iter` with (iter + 11, 79*iter + 121) in `int` (src/erosion.cpp:67-69): 79*iter + 121 first exceeds INT_MAX at iter = 27 183 336
	// ((2^31 - 1 - 121)/79 = 27 183 335.8), from there on it is signed overflow (undefined behaviour in the reference binary), so there is nothing to be identical
	// to -- refused instead of silently diverging.  Droplets 0 .. 27 183 335 are well defined: at most 27 183 336 droplets.
	static constexpr uint32_t MAX_EROSION_ITERS = 27183336u;
	static_assert(79ll*(MAX_EROSION_ITERS - 1) + 121 <= 2147483647ll && 79ll*MAX_EROSION_ITERS + 121 > 2147483647ll, "last droplet whose seed fits an int");
	static void check_erosion_iters(uint32_t num_iters) {if (num_iters > MAX_EROSION_ITERS) throw std::invalid_argument("apply_erosion: more than 27183336 droplets (the reference's int seed 79*iter+121 overflows)");}

	uint32_t spec_batch_override() const {return (uint32_t)opt.ero_batch;} // "ero.batch": rounds per host read-back (0: automatic)
	// TERRA_GEN_FUSED is honoured where a fused kernel meets BASELINE's 1e-5 * zmax_est: sine sums whose every cell takes the short tail (k_sine_grid_mx), simplex / Perlin fBm
	// (terra_fz.hip).  NOT the domain warp (src/mesh_gen.cpp:734-751): its outer sum is sampled at positions the inner sums displace, so their last-bit differences come back
	// multiplied by the outer field's slope -- measured 1.25e-5 * zmax_est on a 300 x 283 grid with contraction allowed throughout, for 1.12x.  It keeps the exact kernel.
	float sine_amp_max(int kstart) const { // max_k |y_scale| of build_arrays (src/mesh_gen.cpp:611): bounds the y table of the sine sum
		float m = 0.0f;
		for (int k = imax(kstart, 0); k < F_TABLE_SIZE; ++k) {float const a = fabsf(mesh_scale_z_inv*sinTable[k][0]); if (a > m) {m = a;}}
		return m;
	}
	static bool fused_kernel_exists(int mode, bool sine_plain_only_) {return (mode == MGEN_SINE) ? sine_plain_only_ : (mode != MGEN_DWARP_GPU);}
	struct spec_cfg_t {uint32_t window = 0 /* auto */, maxb = 256, bshift = 3, slice_steps = 96, max_rounds = 4000000, near_count = 128;} spec_cfg;

	// d_min (optional): min_zval is read from this DEVICE float when the final clamp runs (the only place apply_erosion uses it, src/erosion.cpp:158-162) -- the caller's
	// noise kernel left it there (gen_grid_dev's d_minmax), no host round trip between a heightmap's noise and its erosion
	// sh (optional): one phase of a sharded run (sparse_shard_t).  Phase 1 only traces -- nothing is written to the grid, and when the sparse scheduler would not be tried for
	// this run it does nothing at all; phase 2 is this call with the traces taken from the ranks' arenas instead of made here.
	void apply_erosion_dev(float *d_hmap, int xsize, int ysize, float min_zval, uint32_t num_iters, uint32_t flags, float const *d_min = nullptr, sparse_shard_t const *sh = nullptr) {
		require_scene();
		if (d_min) {min_zval = 0.0f;} // (kept out of every launch argument and of the hipGraph key: a new minimum must not mean a new capture)
		report = terra_erosion_report{};
		if (num_iters == 0 || erode_amount <= 0.0f) return; // erosion disabled (src/erosion.cpp:16)
		if (xsize <= 0 || ysize <= 0 || (uint64_t)(xsize + 8)*(uint64_t)(ysize + 8) >= (1ull << 30)) throw std::invalid_argument("apply_erosion: bad grid size");
		check_erosion_iters(num_iters);
		erosion_consts_t const ec = make_erosion_consts(xsize, ysize, min_zval);
		if (sh && (flags & (TERRA_ERODE_SERIAL | TERRA_ERODE_SERIAL_WAVE))) throw std::invalid_argument("sharded erosion: not with the serial flags");
		if (sh && sh->phase == 1 && !sparse_wanted(ec, num_iters)) return; // (the eroding rank will not look at the arenas either: same test, same answer)
		grid_view_t g;
		g.interior = d_hmap; g.xsize = xsize; g.ysize = ysize; g.NX = ec.NX; g.NY = ec.NY;
		size_t const nborder = grid_view_t::border_floats(xsize, ysize);
		g.border = scratch<float>(s_border, nborder);
		report.droplets = num_iters;
		if (sh && sh->phase == 1) {uint32_t first = 0; (void)sparse_erosion(g, ec, num_iters, false, nullptr, first, sh); return;} // (its graph holds the border launch too)
		be.launch(nborder, [=] TERRA_LAMBDA (size_t i) {border_init_body(g, i);});

		if (flags & TERRA_ERODE_SERIAL) {
			be.launch(1, [=] TERRA_LAMBDA (size_t) {
				direct_mem_t m{g};
				for (uint32_t it = 0; it < num_iters; ++it) {simulate_droplet((int)it, m, ec);}
			}, 64);
			report.windows = 1; report.serial_fallbacks = num_iters;
		}
		else if (flags & TERRA_ERODE_SERIAL_WAVE) { // droplets one after another, each by a whole wave through the LDS window, directly on the grid
			be.launch_waves(1, [=] TERRA_LAMBDA (size_t, wave_scratch_t const &ws) {
				for (uint32_t it = 0; it < num_iters; ++it) {direct_droplet_wave(g, ec, it, nullptr, ws);}
			});
			report.windows = 1; report.serial_fallbacks = num_iters;
		}
		else {
			bool const sparse = (flags & TERRA_ERODE_MINZ_IS_MIN) != 0;
			// a few droplets on a big map: lean traces on the grid + one round per conflicted droplet (sparse_erosion); whatever it leaves -- everything, when it is not
			// tried -- goes through the multi-version scheduler, which then clamps the whole grid (its record of written cells starts where it starts)
			uint32_t first = 0;
			if (sparse_wanted(ec, num_iters)) {
				if (sparse_erosion(g, ec, num_iters, sparse, d_min, first, sh)) return; // complete, sparse clamp applied
			}
			if (first < num_iters && speculative_erosion(g, ec, num_iters, sparse && first == 0, d_min, first)) return; // sparse clamp already applied to every written cell
		}
		// remove padding and clamp to min_zval (src/erosion.cpp:158-162): in place, so only the clamp remains
		size_t const n = (size_t)xsize*ysize;
		be.launch((n + 3)/4, [=] TERRA_LAMBDA (size_t q) {
			size_t const b = q*4, e = (b + 4 < n) ? b + 4 : n;
			float const mz = d_min ? *d_min : min_zval;
			for (size_t i = b; i < e; ++i) {d_hmap[i] = max_std(mz, d_hmap[i]);}
		});
	}

	// ---- the sparse scheduler (terra_erosion.hpp: "sparse regime").  Tried when the expected number of conflicting droplet pairs is small: droplets are ~uniform over the map and
	// two of them meet when their footprints (some tens of 8x8 blocks each) share a block -- measured ~1 pair in 5*10^5 at 1000 droplets on 16384^2, i.e. pairs*8.4/blocks;
	// with N^2 <= 2*blocks that is at most ~8 expected conflicts, each one round (a re-trace, a check, a commit).  TERRA_ERO_SPARSE=0 / 1 (read per call): never / always try
	// it (tests); TERRA_ERO_SPARSE_RETRACES: re-traces before it gives up (default 8).  Results never depend on either.
	static constexpr uint32_t SPARSE_MAX_DROPLETS = 8192, SPARSE_MAX_RETRACES = 8;
	bool sparse_wanted(erosion_consts_t const &ec, uint32_t num_iters) const {
		if (num_iters > SPARSE_MAX_DROPLETS) return false;
		if (opt.ero_sparse == 0) return false;
		if (opt.ero_sparse == 1) return true;
		uint64_t const nblocks = (uint64_t)(((uint32_t)ec.NX >> 3) + 1)*(((uint32_t)ec.NY >> 3) + 1);
		return (uint64_t)num_iters*num_iters <= 2*nblocks;
	}
	// true: all droplets are committed and the clamp is applied (sparsely).  false: droplets [0, first) are on the grid, the caller continues from `first` with the
	// general scheduler (and clamps the whole grid) -- first == num_iters: only the clamp is left.
	// A SHARDED run (one grid over several GPUs; include/terra.h: terra_erosion_shard_*): the sparse scheduler's read-only phases run where the rows live.
	//   phase 1 (every rank, the eroding one included): probe + lean traces of the droplets that start in rows [row0, row1), into the caller's arena -- nothing else, no marks;
	//   phase 2 (the eroding rank): gather every rank's traces into its own arena (sparse_gather_wave), then check / commit / re-trace / clamp as a single context would.
	static uint32_t sparse_touched_cap(uint32_t N) {return (uint32_t)std::min<uint64_t>((uint64_t)N*1024u + 65536u, 64u << 20);}
	size_t sparse_arena_bytes(uint32_t N) const { // the layout below with the record of written cells included: what a rank's arena must hold
		uint32_t const maxb = std::min<uint32_t>(std::max<uint32_t>(spec_cfg.maxb, 16), SPEC_MAXB);
		auto up = [](size_t b) {return (b + 255) & ~(size_t)255;};
		return 2*(up((size_t)N*maxb*SPEC_PAGE*4) + up((size_t)N*maxb*8) + up((size_t)N*maxb*4) + up((size_t)N*4)) + up((size_t)N*4*6) + up(sizeof(sparse_ctl_t)) + up((size_t)sparse_touched_cap(N)*4 + 4);
	}
	bool sparse_erosion(grid_view_t const &g, erosion_consts_t const &ec, uint32_t N, bool record_touched, float const *d_min, uint32_t &first, sparse_shard_t const *sh = nullptr) {
		sparse_buffers_t sb{};
		sb.grid = g; sb.ec = ec; sb.N = N;
		if (sh && sh->phase == 1) {sb.shard = 1; sb.row0 = sh->row0; sb.row1 = sh->row1;}
		sb.maxb = std::min<uint32_t>(std::max<uint32_t>(spec_cfg.maxb, 16), SPEC_MAXB);
		sb.nbx = ((uint32_t)ec.NX >> 3) + 1; sb.nby = ((uint32_t)ec.NY >> 3) + 1;
		sb.max_retraces = SPARSE_MAX_RETRACES;
		if (opt.ero_sparse_retraces >= 0) {sb.max_retraces = (uint32_t)opt.ero_sparse_retraces;}
		size_t const nblocks = (size_t)sb.nbx*sb.nby;
		uint32_t const touched_cap = record_touched ? sparse_touched_cap(N) : 0u;
		size_t off = 0;
		auto carve = [&](size_t bytes) {size_t const o = off; off += (bytes + 255) & ~(size_t)255; return o;};
		size_t o_vals[2], o_mask[2], o_bl[2], o_bc[2];
		for (int b = 0; b < 2; ++b) {o_vals[b] = carve((size_t)N*sb.maxb*SPEC_PAGE*4); o_mask[b] = carve((size_t)N*sb.maxb*8); o_bl[b] = carve((size_t)N*sb.maxb*4); o_bc[b] = carve((size_t)N*4);}
		size_t const o_slot = carve((size_t)N*4*6), o_ctl = carve(sizeof(sparse_ctl_t)), o_touched = carve((size_t)touched_cap*4 + 4); // (the record comes last: the arenas of a sharded run agree on everything in front of it)
		// ~137 KB per droplet.  When the buffer has to grow, the request must fit what the device has free (the same budget rule as the general scheduler's ring, incl. the
		// "ero.mem_budget" test knob); if it does not -- or the allocation fails anyway -- nothing has been touched yet: the general scheduler takes the whole run
		if (!sh && off > s_spec.bytes) {
			size_t avail = be.mem_free() + s_spec.bytes, reserve = (size_t)1 << 30;
			if (opt.ero_mem_budget >= 0) {avail = (size_t)opt.ero_mem_budget; reserve = 0;}
			if (off + reserve > avail) {first = 0; return false;}
		}
		uint8_t *base = sh ? sh->arena : nullptr; // (a sharded run works in the caller's arena: sparse_arena_bytes())
		if (!sh) {
			try {base = scratch<uint8_t>(s_spec, off);} // (the general scheduler's ring lives in the same grow-only buffer: the two never run at the same time)
			catch (std::exception const &) {first = 0; return false;}
		}
		for (int b = 0; b < 2; ++b) {
			sb.page_vals[b] = (float *)(base + o_vals[b]); sb.page_mask[b] = (unsigned long long *)(base + o_mask[b]);
			sb.blk_list[b] = (uint32_t *)(base + o_bl[b]); sb.blk_cnt[b] = (uint32_t *)(base + o_bc[b]);
		}
		uint32_t *slot_arrays = (uint32_t *)(base + o_slot);
		sb.cur = slot_arrays; sb.state = slot_arrays + N; sb.nsteps = slot_arrays + 2*(size_t)N; sb.nan = slot_arrays + 3*(size_t)N; sb.work = slot_arrays + 4*(size_t)N; sb.queued = slot_arrays + 5*(size_t)N;
		sb.trace_groups = (N <= 64) ? N : std::max<uint32_t>(64, N/4); // a quarter of the droplets at most get a wave of their own at once: enough for a map with some ocean, and a fully dry map's waves then take four droplets each
		sb.ctl = (sparse_ctl_t *)(base + o_ctl);
		sb.touched = record_touched ? (uint32_t *)(base + o_touched) : nullptr; sb.touched_cap = touched_cap;
		if (sh && sh->phase == 1) { // a tracer: the first step of its droplets, then their lean traces -- into the arena, nothing else (no marks: wmin stays with the eroding rank)
			sparse_buffers_t const s = sb;
			// one graph: a rank makes this call every step of the one-grid pipeline, from a host thread that also enqueues the step's noise and collectives
			struct {sparse_buffers_t s; uint32_t tag;} gkey;
			memset(&gkey, 0, sizeof(gkey)); gkey.s = s; gkey.tag = 0x53504831u;
			first = 0;
			if (be.graph_replay(&gkey, sizeof(gkey))) return false;
			bool const cap = be.graph_begin();
			try {
				grid_view_t const gg = g;
				be.launch(grid_view_t::border_floats(g.xsize, g.ysize), [=] TERRA_LAMBDA (size_t i) {border_init_body(gg, i);});
				be.launch(1, [=] TERRA_LAMBDA (size_t) {sparse_ctl_t c{}; c.base = 0; c.c = s.N; *s.ctl = c;}, 64);
				be.launch(N, [=] TERRA_LAMBDA (size_t i) {sparse_probe_body(s, (uint32_t)i);});
				be.launch_waves_lean(s.trace_groups, [=] TERRA_LAMBDA (size_t i, lean_scratch_t const &ws) {sparse_trace_wave(s, (uint32_t)i, ws);});
			} catch (...) {be.graph_abort(); throw;}
			if (cap) {be.graph_end(&gkey, sizeof(gkey));}
			return false;
		}
		uint32_t *blk_arrays = scratch<uint32_t>(s_spec_blocks, 2*nblocks); // [head | dirty_min] of the general scheduler, all SPEC_NIL between runs: wmin borrows the second half
		if (spec_blocks_clean != blk_arrays || spec_blocks_n != nblocks) {be.fill32(blk_arrays, SPEC_NIL, 2*nblocks);}
		spec_blocks_clean = nullptr;
		sb.wmin = blk_arrays + nblocks;
		sparse_buffers_t const s = sb;
		sparse_shard_t const shv = sh ? *sh : sparse_shard_t{};
		bool const gather = sh != nullptr; // (phase 2)
		auto rounds = [&](bool with_first) { // [control block, probe, trace, check, commit,] then: re-trace the lowest conflicted droplet, check, commit -- one hipGraph each way
			struct {sparse_buffers_t s; uint32_t with_first; uint32_t tag; sparse_shard_t sh;} gkey;
			memset(&gkey, 0, sizeof(gkey)); gkey.s = s; gkey.s.ec.min_zval = 0.0f; gkey.with_first = with_first ? 1u : 0u; gkey.tag = 0x53505253u; // (min_zval: read by the clamp only, which is not part of the graph)
			if (gather) {gkey.sh.phase = 2; gkey.sh.world = shv.world; gkey.sh.self = shv.self; gkey.sh.stride = shv.stride; gkey.sh.rows = shv.rows;}
			if (be.graph_replay(&gkey, sizeof(gkey))) return;
			bool const cap = be.graph_begin();
			try {
				if (with_first) {
					be.launch(1, [=] TERRA_LAMBDA (size_t) {sparse_ctl_t c{}; c.base = 0; c.c = s.N; *s.ctl = c;}, 64);
					if (gather) { // the traces were made where the rows live (phase 1 on every rank): fetch them, make the marks, build the work list
						sparse_rows_t const rows = shv.rows; uint32_t const world = shv.world, self = shv.self; long long const stride = shv.stride;
						be.launch_waves_nolds(N, [=] TERRA_LAMBDA (size_t i) {sparse_gather_wave(s, (uint32_t)i, rows, world, self, stride);});
					}
					else {
					be.launch(N, [=] TERRA_LAMBDA (size_t i) {sparse_probe_body(s, (uint32_t)i);}); // the first step of every droplet: most end there
					be.launch_waves_lean(s.trace_groups, [=] TERRA_LAMBDA (size_t i, lean_scratch_t const &ws) {sparse_trace_wave(s, (uint32_t)i, ws);});
					}
					be.launch_waves_nolds(N, [=] TERRA_LAMBDA (size_t i) {sparse_check_wave(s, (uint32_t)i);});
					be.launch_waves_nolds(s.trace_groups, [=] TERRA_LAMBDA (size_t i) {sparse_commit_wave(s, (uint32_t)i);});
				}
				for (int r = 0; r < 1; ++r) { // one more conflicted droplet per replay (the usual run has none or one: its read-back is then the only host round trip)
					be.launch_waves_lean(1, [=] TERRA_LAMBDA (size_t, lean_scratch_t const &ws) {sparse_retrace_wave(s, ws);});
					be.launch_waves_nolds(N, [=] TERRA_LAMBDA (size_t i) {sparse_check_wave(s, (uint32_t)i);});
					be.launch_waves_nolds(s.trace_groups, [=] TERRA_LAMBDA (size_t i) {sparse_commit_wave(s, (uint32_t)i);});
				}
			} catch (...) {be.graph_abort(); throw;}
			if (cap) {be.graph_end(&gkey, sizeof(gkey));}
		};
		// behind the rounds, BEFORE the host knows how they went: reset the marks and clamp the written cells if the run turns out complete (both look at the control block on
		// the device and do nothing otherwise) -- in the usual case the control block read-back below is the last thing the call waits for
		float const mz = ec.min_zval;
		uint32_t const clamp_threads = record_touched ? std::min<uint32_t>(touched_cap, 1u << 18) : 0u;
		auto finish = [&](bool force) {
			be.launch_waves_nolds(s.trace_groups, [=] TERRA_LAMBDA (size_t i) {sparse_unmark_wave(s, (uint32_t)i, force);});
			if (clamp_threads) {be.launch(clamp_threads, [=] TERRA_LAMBDA (size_t i) {sparse_clamp_body(s, (uint32_t)i, clamp_threads, d_min ? *d_min : mz, false);});} // (never forced: an incomplete run's clamp is the caller's, at the end)
		};
		sparse_ctl_t hc{};
		rounds(true);
		finish(false);
		be.d2h(&hc, sb.ctl, sizeof(hc)); // the one host round trip of a run with at most one conflicted droplet
		uint32_t batches = 1;
		bool finished_on_device = (hc.c >= N) && !hc.bail;
		while (hc.c < N && !hc.bail) { // after the last commit: [base, c) is on the grid, c is conflicted
			if (batches++ > N) throw std::runtime_error("sparse erosion: no progress");
			rounds(false);
			finish(false);
			be.d2h(&hc, sb.ctl, sizeof(hc));
			finished_on_device = (hc.c >= N) && !hc.bail;
		}
		bool const complete = (hc.c >= N) && !hc.bail;
		first = complete ? N : (hc.bail ? hc.base : hc.c);
		if (!finished_on_device) {be.launch_waves_nolds(s.trace_groups, [=] TERRA_LAMBDA (size_t i) {sparse_unmark_wave(s, (uint32_t)i, true);});} // handed over: wmin[] all SPEC_NIL again (head[] was never touched)
		spec_blocks_clean = blk_arrays; spec_blocks_n = nblocks;
		report.rounds = 1 + hc.retraces; report.traces = N + hc.retraces; report.traced_steps = hc.traced_steps; report.steps = hc.steps; report.nan_droplets = hc.nan_droplets;
		report.windows = 1; report.sparse_droplets = first; report.sparse_retraces = hc.retraces; report.sparse_probe_only = N - hc.nwork;
		return complete && record_touched && hc.touched <= sb.touched_cap; // (else the caller clamps the whole grid: handed over, no record wanted, or the record overflowed)
	}

	// Defaults from the measurements in profiles/r02_erosion_near_far_sweep.txt: the 512 droplets next in line for the commit trace to the end, the rest of the ring
	// advances 128 steps per round (13 % faster on 4096^2 with 10^6 droplets than "everybody to the end", 37 % on 1024^2 with 30 000, neutral on sparse maps).
	// Sliding ring of W in-flight droplets (terra_erosion.hpp).  One round = every unfinished droplet advances by at most `slice` steps,
	// finished versions are published, dependants of changed versions start over, the valid finished prefix is flushed to the grid and its
	// slots are handed to the next droplets.  While droplets are waiting for a slot the traces are sliced, so that one long path (they run
	// to thousands of steps at ~1 us each) does not hold up a whole window; once everything is admitted the remaining traces run to the end.
	// returns true when the final clamp was applied sparsely (record_touched and the record did not overflow)
	// first: droplets [0, first) are already on the grid (committed by sparse_erosion); the run covers [first, num_iters) and adds its counts to `report`
	bool speculative_erosion(grid_view_t const &g, erosion_consts_t const &ec, uint32_t num_iters, bool record_touched, float const *d_min = nullptr, uint32_t first = 0) {
		spec_buffers_t sb{};
		sb.grid = g; sb.ec = ec; sb.num_iters = num_iters;
		// ring slots: more droplets in flight = more parallel work and fewer rounds, but also more speculation on stale cells and longer writer lists.  Measured
		// (MI355X, 10^5..10^6 droplets, profiles/r02_erosion_ring_size_sweep.txt): best near one slot per 8K cells on sparse maps (8192^2: 8192 slots, 16384^2: 32768),
		// 3072-4096 slots on a dense 4096^2, 2048 on 1024^2 (4096 slots there double the re-traces).  Per slot and buffer: 256 pages of 64 floats (64 KiB) + 256 masks and block ids
		// (3 KiB) + up to 16 checkpoints (their masks: 32 KiB) + the undo log (32 KiB): ~133 KiB, twice = ~266 KiB per slot (+ 6 KiB of list records and dirty lists) -- a 32768-slot ring (16384^2 map) is ~8.5 GiB of the 288
		// (bench.py keeps 4 contexts in flight: ~34 GiB).  The checkpoint and undo arrays are sized by the checkpoint count in force (none with TERRA_ERO_CK=n:0).
		uint64_t const ncells = (uint64_t)ec.NX*ec.NY;
		uint32_t auto_w = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ncells >> 13, 2048), 32768);
		if (ncells >= (1ull << 24)) {auto_w = std::max<uint32_t>(auto_w, 4096);}
		uint32_t W = std::min<uint32_t>(spec_cfg.window ? spec_cfg.window : auto_w, num_iters - first);
		sb.near_count = spec_cfg.near_count;
		sb.ck_steps = SPEC_CK_STEPS; sb.ck_max = SPEC_CK_MAX;
		sb.live_partial = opt.ero_live ? 1u : 0u; // (options "ero.live", "ero.diag", "ero.ck": results never depend on them)
		sb.diag = opt.ero_diag ? 1u : 0u;
		if (opt.ero_ck_steps >= 1) {sb.ck_steps = (uint32_t)opt.ero_ck_steps; sb.ck_max = (uint32_t)opt.ero_ck_max;}
		sb.maxb = std::min<uint32_t>(std::max<uint32_t>(spec_cfg.maxb, 16), SPEC_MAXB); sb.bshift = 3; // a version page is the 8 x 8 cells of a block
		// where a cell is read from is a 31-bit float index into a version buffer, (slot*maxb + entry)*64 + cell, with bit 31 naming the buffer (spec_back_t::source,
		// spec_cand_t::page): the ring must keep W*maxb*64 below 2^31 -- a larger request is served with the largest ring that does (results never depend on W)
		{uint64_t const most = (((1ull << 31) - 1)/SPEC_PAGE)/sb.maxb; if (W > most) {W = (uint32_t)most;}}
		sb.W = W;
		if (opt.ero_near_set) {int const v = opt.ero_near; sb.near_count = (v >= 0) ? (uint32_t)v : W/(uint32_t)(-v);} // "ero.near" (negative: a fraction of the ring); results never depend on it
		sb.nbx = ((uint32_t)ec.NX >> sb.bshift) + 1; sb.nby = ((uint32_t)ec.NY >> sb.bshift) + 1;
		size_t const nblocks = (size_t)sb.nbx*sb.nby;
		// carve one allocation.  The ring is the one big buffer of the library (~266 KiB per slot): it must fit what the device has free right now -- several contexts share
		// a GPU (bench.py keeps 4 heightmaps in flight), and a result never depends on W -- so a ring that would not fit is made smaller until it does
		size_t off = 0;
		auto carve = [&](size_t bytes) {size_t const o = off; off += (bytes + 255) & ~(size_t)255; return o;};
		size_t o_vals[2], o_mask[2], o_bl[2], o_bc[2], o_cks[2], o_ckn[2], o_cku[2], o_ckm[2], o_ckc[2], o_ui[2], o_uv[2], o_un[2];
		size_t o_slot = 0, o_state = 0, o_resume = 0, o_next = 0, o_nodeblk = 0, o_dlist = 0, o_ctl = 0, o_touched = 0, o_done = 0;
		uint32_t const touched_cap = record_touched ? (uint32_t)std::min<uint64_t>((uint64_t)num_iters*1024u + 65536u, 64u << 20) : 0u;
		auto layout = [&](uint32_t W) {
			off = 0;
			for (int b = 0; b < 2; ++b) {o_vals[b] = carve((size_t)W*sb.maxb*SPEC_PAGE*4); o_mask[b] = carve((size_t)W*sb.maxb*8); o_bl[b] = carve((size_t)W*sb.maxb*4); o_bc[b] = carve(W*4);}
			for (int b = 0; b < 2; ++b) {
				// (indexed with the stride SPEC_CK_MAX / SPEC_UNDO_MAX per slot; without checkpoints nothing is ever read or written there, so nothing is allocated)
				size_t const ckn = sb.ck_max ? SPEC_CK_MAX : 0, unn = sb.ck_max ? SPEC_UNDO_MAX : 0;
				o_cks[b] = carve((size_t)W*ckn*sizeof(droplet_state_t)); o_ckn[b] = carve((size_t)W*ckn*4); o_cku[b] = carve((size_t)W*ckn*4);
				o_ckm[b] = carve((size_t)W*ckn*sb.maxb*8); o_ckc[b] = carve(W*4); o_ui[b] = carve((size_t)W*unn*4); o_uv[b] = carve((size_t)W*unn*4); o_un[b] = carve(W*4);
			}
			o_slot = carve((size_t)W*4*14); // it, phase, has_ver, cur, changed, restart, run_nblk, flags, nsteps, linked, rsrc, rat, rentry, vbuf
			o_state = carve((size_t)W*sizeof(droplet_state_t)); o_resume = carve((size_t)W*sizeof(spec_resume_t));
			o_next = carve((size_t)W*sb.maxb*sizeof(spec_u32x4)); o_nodeblk = carve((size_t)W*sb.maxb*4); o_dlist = carve((size_t)W*sb.maxb*16); o_ctl = carve(sizeof(spec_ctl_t));
			o_touched = carve((size_t)touched_cap*4 + 4);
			o_done = carve(((size_t)W + 63)/64*4);
			return off;
		};
		{
			size_t need = layout(W);
			if (need > s_spec.bytes) { // it has to grow: what is free now + what the old ring gives back, minus a reserve for everybody else's next allocation
				size_t avail = be.mem_free() + s_spec.bytes, reserve = (size_t)1 << 30;
				if (opt.ero_mem_budget >= 0) {avail = (size_t)opt.ero_mem_budget; reserve = 0;} // "ero.mem_budget" (tests): pretend this many bytes are free
				while (W > 256 && need + reserve > avail) {W = W - W/4; need = layout(W);}
				sb.W = W;
			}
		}
		uint8_t *base = scratch<uint8_t>(s_spec, off);
		for (int b = 0; b < 2; ++b) {
			sb.page_vals[b] = (float *)(base + o_vals[b]); sb.page_mask[b] = (unsigned long long *)(base + o_mask[b]); // nothing to initialise: only entries below a version's count are ever read
			sb.ck_state[b] = (droplet_state_t *)(base + o_cks[b]); sb.ck_nblk[b] = (uint32_t *)(base + o_ckn[b]); sb.ck_undo[b] = (uint32_t *)(base + o_cku[b]);
			sb.ck_masks[b] = (unsigned long long *)(base + o_ckm[b]); sb.ck_cnt[b] = (uint32_t *)(base + o_ckc[b]);
			sb.undo_idx[b] = (uint32_t *)(base + o_ui[b]); sb.undo_val[b] = (float *)(base + o_uv[b]); sb.undo_n[b] = (uint32_t *)(base + o_un[b]);
			sb.blk_list[b] = (uint32_t *)(base + o_bl[b]); sb.blk_cnt[b] = (uint32_t *)(base + o_bc[b]);
		}
		uint32_t *slot_arrays = (uint32_t *)(base + o_slot);
		sb.it = slot_arrays; sb.phase = slot_arrays + W; sb.has_ver = slot_arrays + 2*(size_t)W; sb.cur = slot_arrays + 3*(size_t)W; sb.changed = slot_arrays + 4*(size_t)W;
		sb.restart = slot_arrays + 5*(size_t)W; sb.run_nblk = slot_arrays + 6*(size_t)W; sb.flags = slot_arrays + 7*(size_t)W; sb.nsteps = slot_arrays + 8*(size_t)W;
		sb.linked = slot_arrays + 9*(size_t)W; sb.rsrc = slot_arrays + 10*(size_t)W; sb.rat = slot_arrays + 11*(size_t)W; sb.rentry = slot_arrays + 12*(size_t)W; sb.vbuf = slot_arrays + 13*(size_t)W;
		sb.state = (droplet_state_t *)(base + o_state); sb.resume = (spec_resume_t *)(base + o_resume);
		sb.touched = record_touched ? (uint32_t *)(base + o_touched) : nullptr; sb.touched_cap = touched_cap;
		sb.done_cnt = (uint32_t *)(base + o_done);
		sb.node_rec = (spec_u32x4 *)(base + o_next); sb.node_blk = (uint32_t *)(base + o_nodeblk); sb.dirty_list = (uint32_t *)(base + o_dlist); sb.dirty_list2[0] = sb.dirty_list + 2*(size_t)W*sb.maxb; sb.dirty_list2[1] = sb.dirty_list + 3*(size_t)W*sb.maxb; sb.ctl = (spec_ctl_t *)(base + o_ctl);
		// block -> list head and block -> dirty mark: one entry per 8x8 block of the padded grid.  Every run resets exactly the entries it set
		// (spec_unlink_body / spec_undirty_body), so the O(grid) fill is paid only when the arrays are (re)allocated or the grid shape changes.
		uint32_t *blk_arrays = scratch<uint32_t>(s_spec_blocks, 2*nblocks);
		sb.head = blk_arrays; sb.dirty_min = blk_arrays + nblocks;
		if (spec_blocks_clean != blk_arrays || spec_blocks_n != nblocks) {be.fill32(blk_arrays, SPEC_NIL, 2*nblocks);}
		spec_blocks_clean = nullptr; // not clean again until this run has taken its lists apart
		spec_buffers_t const s = sb;
		be.fill32(slot_arrays, 0, (size_t)W*11);
		be.fill32(sb.rat, SPEC_NIL, (size_t)W*2); // rat, rentry
		be.fill32(sb.vbuf, SPEC_VIS_NONE, W);
		be.fill32(sb.blk_cnt[0], 0, W); be.fill32(sb.blk_cnt[1], 0, W);
		for (int b = 0; b < 2; ++b) {be.fill32(sb.ck_cnt[b], 0, W); be.fill32(sb.undo_n[b], 0, W);}
		be.fill32(sb.node_blk, SPEC_NIL, (size_t)W*sb.maxb);
		be.fill32(sb.done_cnt, 0, ((size_t)W + 63)/64);
		be.launch(W, [=] TERRA_LAMBDA (size_t i) { // the first W droplets take the slots (num_iters - first >= W); droplet `it` lives in slot it % W
			uint32_t const it = first + (uint32_t)i, slot = it % s.W;
			s.it[slot] = it; s.phase[slot] = SPEC_FRESH;
			if (i == 0) {spec_ctl_t c{}; c.base = first; c.new_base = first + s.W; c.stop_at = SPEC_NIL; c.new_stop = SPEC_NIL; *s.ctl = c;}
		});
		report.windows = (num_iters - first + W - 1)/W;
		uint32_t const slice = std::max<uint32_t>(spec_cfg.slice_steps, 1);
		uint32_t host_base = first, launched = 0;
		spec_ctl_t hc{};
		// One round = 7 dependent launches (the trace waves; marks + unlink + publish; link + mark; restarts + commit point; commit + resume; hand-over; end of round -- "ero.fuse" 4 folds the last three
		// into one, which loses), a batch of rounds captured into ONE hipGraph and
		// replayed.  Nothing in a round needs a host decision -- the step budget, the commit point and the pause behind a failed droplet are all taken from the
		// device-resident control block -- so the host queues rounds in batches and reads the control block back once per batch; a round after the end (or while the
		// lowest droplet waits for its serial fall-back) finds nothing to do.
		// nrounds rounds as ONE graph: consecutive graph launches leave ~9-13 us between them (profiles/r06_erosion_round_anatomy.txt), launches inside a graph none
		int const fuse = opt.ero_fuse;
		auto some_rounds = [&](uint32_t nrounds) {
			struct {spec_buffers_t s; uint32_t slice; uint32_t tag; uint32_t nrounds; uint32_t fuse;} gkey;
			memset(&gkey, 0, sizeof(gkey)); gkey.s = s; gkey.s.ec.min_zval = 0.0f; gkey.slice = slice; gkey.tag = 0x524e4432u; gkey.nrounds = nrounds; gkey.fuse = (uint32_t)fuse; // (min_zval: read by the clamp only, which is not part of the graph)
			if (be.graph_replay(&gkey, sizeof(gkey))) return;
			bool const cap = be.graph_begin();
			try {
			  for (uint32_t rr = 0; rr < nrounds; ++rr) {
				// workgroups are dispatched in index order and a ring has more droplets than the chip holds waves: workgroup i takes the i-th in-flight droplet (slot (base + i) % W), so
				// that the droplets next in line for the commit -- the long, unbudgeted traces everybody waits for -- start first instead of wherever their slot number falls
				be.launch_waves(W, [=] TERRA_LAMBDA (size_t i, wave_scratch_t const &ws) {spec_trace_wave(s, (uint32_t)(((uint64_t)s.ctl->base + i) % s.W), slice, ws);});
				// dirty marks from the OLD published versions, then the slot's nodes out of the writer lists and its finished version published (one wave per slot: spec_post_unlink_flip_wave)
				if (fuse & 2) {be.launch_waves_nolds(W, [=] TERRA_LAMBDA (size_t i) {spec_post_unlink_flip_wave(s, (uint32_t)i);});}
				else {
					be.launch_waves_nolds(W, [=] TERRA_LAMBDA (size_t i) {spec_post_wave(s, (uint32_t)i);});
					be.launch((size_t)W*64, [=] TERRA_LAMBDA (size_t i) {
						uint32_t const slot = (uint32_t)(i >> 6), l = (uint32_t)(i & 63u), n = s.linked[slot];
						for (uint32_t e = l; e < n; e += 64) {spec_unlink_body(s, slot*s.maxb + e);}
						if (l == 0) {spec_flip_body(s, slot);}
					});
				}
				// the two passes below: 64 logical threads per slot that walk the slot's entries up to the count in use (a footprint holds ~30 of its 256 entries;
				// one thread per (slot, entry) made these passes cost as much as the traces on a 32768-slot ring)
				be.launch((size_t)W*64, [=] TERRA_LAMBDA (size_t i) { // rebuild the writer lists from the published versions; who must start over
					uint32_t const slot = (uint32_t)(i >> 6), l = (uint32_t)(i & 63u);
					uint32_t const pub = spec_visible_count(s, slot), run = s.run_nblk[slot], n = (pub > run) ? pub : run; // (pub: entries of the version higher droplets read)
					for (uint32_t e = l; e < n; e += 64) {spec_link_body(s, slot, e); spec_mark_body(s, slot, e);}
					if (l == 0) {s.linked[slot] = pub;}
				});
				be.launch((size_t)W*64, [=] TERRA_LAMBDA (size_t i) { // apply the restarts, find the commit point; reset the dirty marks
					uint32_t const nd1 = s.ctl->ndirty, nd2 = (s.ctl->par & 1u) ? s.ctl->nd2[0] : s.ctl->nd2[1], nd = (nd1 > nd2) ? nd1 : nd2;
					for (size_t k = i; k < nd; k += (size_t)s.W*64) {spec_undirty_body(s, (uint32_t)k);}
					if (i < s.W) {spec_scan_body(s, (uint32_t)i);}
				});
				// commit; a re-trace that can resume from a checkpoint becomes a suspended trace at that checkpoint; committed slots go to the next droplets; the last wave closes the round
				if (fuse & 4) {be.launch_waves_nolds(W, [=] TERRA_LAMBDA (size_t i) {spec_close_wave(s, (uint32_t)i);});}
				else {
					be.launch_waves_nolds(W, [=] TERRA_LAMBDA (size_t i) {spec_flush_wave(s, (uint32_t)i); spec_resume_wave(s, (uint32_t)i);});
					be.launch(W, [=] TERRA_LAMBDA (size_t i) {spec_admit_body(s, (uint32_t)i);});
					be.launch(1, [=] TERRA_LAMBDA (size_t) {spec_advance_body(s);});
				}
			  }
			} catch (...) {be.graph_abort(); throw;}
			if (cap) {be.graph_end(&gkey, sizeof(gkey));}
		};
		while (host_base < num_iters) {
			// the first batch is short (a sparse map is done after two rounds); later ones amortise the read-back over 8 rounds
			uint32_t batch = (launched == 0) ? 2u : 8u;
			if (spec_batch_override()) {batch = spec_batch_override();}
			if (launched + batch > spec_cfg.max_rounds) throw std::runtime_error("speculative erosion: round limit reached");
			if (fuse & 1) {some_rounds(batch);} else {for (uint32_t r = 0; r < batch; ++r) {some_rounds(1);}}
			launched += batch;
			be.d2h(&hc, sb.ctl, sizeof(hc)); // the one host round trip of the batch
			host_base = hc.base;
			if (host_base < num_iters && hc.stop_at == host_base) { // the lowest uncommitted droplet overflowed its block list: it runs alone, directly on the grid
				uint32_t const it = host_base;
				grid_view_t const gg = g; erosion_consts_t const ee = ec;
				uint32_t *fb = &sb.ctl->fb_steps, *tcount = &sb.ctl->touched;
				uint32_t *tch = sb.touched; uint32_t const tcap = sb.touched_cap;
				be.launch_waves(1, [=] TERRA_LAMBDA (size_t, wave_scratch_t const &ws) {direct_droplet_wave(gg, ee, it, fb, ws, tch, tcount, tcap);});
				be.launch(W, [=] TERRA_LAMBDA (size_t i) {spec_fallback_reset_body(s, (uint32_t)i);});
				be.launch(1, [=] TERRA_LAMBDA (size_t) {spec_fallback_advance_body(s);});
				++report.serial_fallbacks; ++host_base;
			}
		}
		be.launch((size_t)W*sb.maxb, [=] TERRA_LAMBDA (size_t i) {spec_unlink_body(s, (uint32_t)i);}); // leave head[] all-NIL (dirty_min[] already is)
		spec_blocks_clean = blk_arrays; spec_blocks_n = nblocks;
		be.d2h(&hc, sb.ctl, sizeof(hc));
		report.rounds += hc.rounds; report.retraces_same = hc.retraces_same; report.checkpoint_resumes = hc.ck_resumes; report.checkpoint_steps_saved = hc.ck_steps_saved;
		report.traces += hc.traces; report.traced_steps += hc.traced_steps; report.steps += hc.steps; report.nan_droplets += hc.nan_droplets; // (+=: sparse_erosion may have committed a prefix)
		report.window_shifts = hc.n_shift;
		report.critical_steps = hc.crit_steps; report.critical_shifts = hc.crit_shifts;
		report.clk_wave = hc.clk_wave; report.clk_init = hc.clk_init; report.clk_shift = hc.clk_shift; report.clk_tail = hc.clk_tail; report.clk_critical = hc.clk_crit;
		report.clk_shift_flush = hc.clk_sh_flush; report.clk_shift_prep = hc.clk_sh_prep; report.clk_shift_load = hc.clk_sh_load;
		report.crit_clk_flush = hc.crit_own_flush; report.crit_clk_load = hc.crit_own_load; report.crit_clk_prep = hc.crit_own_prep;
		report.crit_clk_shift = hc.crit_own_shift; report.crit_clk_edge = hc.crit_own_edge; report.crit_steps_own = hc.crit_own_steps;
		if (!record_touched) return false;
		uint32_t const ntouched = hc.touched;
		if (ntouched > sb.touched_cap) return false; // record overflowed: the caller clamps the whole grid
		uint32_t const *tch = sb.touched; float const mz = ec.min_zval; grid_view_t const gg = g;
		be.launch(ntouched, [=] TERRA_LAMBDA (size_t i) {touched_clamp_body(gg, tch, (uint32_t)i, d_min ? *d_min : mz);});
		return true;
	}

	// ================================================================ tiles (a10, a13, K6, K7)
	// One height field of tw x tw cells per tile, origin (tile*128 - shift) cells: setup_height_gen_async(height_gen, x1 - shift, y1 - shift, tw, tw)
	// + the eval_index loop (src/tiled_mesh.cpp:458-464,480-488,494-505).  tw = 130, shift = 0: the tile's zvals; tw = 201, shift = 36: its AO context.
	// Returns the device copy of the tile references (valid until the next tile call of this context).
	tile_ref_pod_t const *tile_fields_dev(int32_t const *tile_xy, uint32_t n, uint32_t tw, int shift, float *d_out, float xy_scale = 1.0f, // xy_scale 0: only the tile references
		bool glac = true, bool force_sine = false, int min_start_sin = 0, // enable_glaciate() after build_arrays; build_arrays' force_sine_mode; eval_index's min_start_sin
		tile_band_t const *band = nullptr, bool *band_used = nullptr)         // band: only these cells of every tile's field, if the backend can (*band_used); else the whole fields
	{
		uint32_t const size = 128, zv = tw;
		if (band_used) {*band_used = false;}
		float const fdx = xy_scale*DX_VAL, fdy = xy_scale*DY_VAL; // setup_height_gen_async: build_arrays(..., xy_scale*DX_VAL, xy_scale*DY_VAL, ...)
		// a tile's X table depends only on its tile x, its Y table only on its tile y: build each distinct one once
		std::vector<int32_t> ux, uy;
		for (uint32_t i = 0; i < n; ++i) {ux.push_back(tile_xy[2*i]); uy.push_back(tile_xy[2*i+1]);}
		std::sort(ux.begin(), ux.end()); ux.erase(std::unique(ux.begin(), ux.end()), ux.end());
		std::sort(uy.begin(), uy.end()); uy.erase(std::unique(uy.begin(), uy.end()), uy.end());
		typedef tile_ref_pod_t tile_ref_t;
		std::vector<tile_ref_t> refs(n);
		for (uint32_t i = 0; i < n; ++i) {
			refs[i].tx = tile_xy[2*i]; refs[i].ty = tile_xy[2*i+1];
			refs[i].xi = (uint32_t)(std::lower_bound(ux.begin(), ux.end(), refs[i].tx) - ux.begin());
			refs[i].yi = (uint32_t)(std::lower_bound(uy.begin(), uy.end(), refs[i].ty) - uy.begin());
		}
		uint32_t const nux = (uint32_t)ux.size(), nuy = (uint32_t)uy.size();
		bool unique_tiles = true; // a batch may name a tile twice: the scatter of the virtual grid could serve only one of the copies
		{std::vector<std::pair<int32_t, int32_t>> tt; for (uint32_t i = 0; i < n; ++i) {tt.push_back(std::make_pair(tile_xy[2*i], tile_xy[2*i+1]));} std::sort(tt.begin(), tt.end()); unique_tiles = (std::adjacent_find(tt.begin(), tt.end()) == tt.end());}
		// tables of all distinct tile columns / rows side by side, k-major like the big-grid tables: xt[k][u*tw + c], yt[k][u*tw + c].
		// The batch is then ONE "virtual" (nux*tw) x (nuy*tw) sine grid whose cells are exactly the requested tiles' cells.
		int const md_ = force_sine ? (int)MGEN_SINE : mode;
		bool const banded = band && xy_scale != 0.0f && d_out && be.tile_band_ok(n, nux, nuy, band->twx, unique_tiles, md_ == MGEN_SINE && sine_plain_only(force_sine ? 0 : shape, imax(start_eval_sin, min_start_sin)), md_);
		if (band_used) {*band_used = banded;}
		tile_band_t const bd = banded ? *band : tile_band_t{zv, zv, zv, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u};
		uint32_t const zvx = bd.twx, zvy = bd.twy; // cells per tile in the virtual grid
		uint32_t const nxpv = round_up(nux*zvx, 128), nypv = round_up(nuy*zvy, 128);
		size_t const tab_floats = (size_t)F_TABLE_SIZE*(nxpv + nypv), sm_floats = (size_t)nux*zvx + (size_t)nuy*zvy;
		// one parameter block: tile references | origins of the distinct columns / rows | their per-k constants -- assembled on the host, ONE asynchronous upload
		size_t const o_refs = 0, o_m0 = (refs.size()*sizeof(tile_ref_t) + 255) & ~(size_t)255, o_sk = o_m0 + (((size_t)(nux + nuy)*4 + 255) & ~(size_t)255);
		size_t const par_bytes = o_sk + (((nux + nuy)*sizeof(sine_k_t) + 255) & ~(size_t)255);
		size_t const bytes = par_bytes + (tab_floats + sm_floats)*4 + 1024;
		uint8_t *base = scratch<uint8_t>(s_tiles, bytes);
		tile_ref_t *d_refs = (tile_ref_t *)(base + o_refs);
		float *d_m0 = (float *)(base + o_m0);
		sine_k_t *d_sk = (sine_k_t *)(base + o_sk);
		float *d_tab = (float *)(base + par_bytes);
		float *d_sm = d_tab + tab_floats;
		std::vector<uint8_t> par((xy_scale == 0.0f) ? o_m0 : par_bytes, 0);
		memcpy(par.data() + o_refs, refs.data(), refs.size()*sizeof(tile_ref_t));
		if (xy_scale == 0.0f) {be.h2d_async(base, par.data(), par.size()); return d_refs;}
		// per distinct tx / ty: build_arrays((x0 - MESH_X_SIZE/2), (y0 - MESH_Y_SIZE/2), DX_VAL, DY_VAL, tw, tw) with x0 = x1 - shift (src/tiled_mesh.cpp:458-464)
		{
			float *h_m0 = (float *)(par.data() + o_m0); sine_k_t *sks = (sine_k_t *)(par.data() + o_sk);
			for (uint32_t i = 0; i < nux; ++i) {float const x0 = (float)((ux[i]*(int)size - shift) - cfg.mesh_x/2); h_m0[i] = fdx*x0; sks[i] = make_sine_k(h_m0[i], 0.0f, fdx, fdy);}
			for (uint32_t i = 0; i < nuy; ++i) {float const y0 = (float)((uy[i]*(int)size - shift) - cfg.mesh_y/2); h_m0[nux+i] = fdy*y0; sks[nux+i] = make_sine_k(0.0f, h_m0[nux+i], fdx, fdy);}
		}
		be.h2d_async(base, par.data(), par.size());
		noise_consts_t const nc = consts();
		sin_lut_t const L = lut();
		int const md = force_sine ? (int)MGEN_SINE : mode, shp = force_sine ? 0 : shape, kstart = imax(start_eval_sin, min_start_sin);
		bool const use_sm = glac && (hp.sine_mag > 0.0f);
		float const dxv = fdx, dyv = fdy, dxi = DX_VAL_INV, dyi = DY_VAL_INV;
		if (use_sm) { // enable_glaciate per distinct tx / ty
			float const sm_scale = hp.sine_mag*mesh_scale_z_inv, freq = mesh_scale*hp.sine_freq;
			size_t const nsx = (size_t)nux*zvx;
			be.launch(sm_floats, [=] TERRA_LAMBDA (size_t i) { // (c: the cell's coordinate in the tile's field -- a band's coordinates skip its gap)
				if (i < nsx) {unsigned const u = (unsigned)(i / zvx), c = tile_band_coord((unsigned)(i % zvx), bd.xsplit, bd.xgap); d_sm[i] = sm_scale*L.COSF(((float)c*dxv + d_m0[u])*dxi*freq);}
				else {size_t const j = i - nsx; unsigned const u = (unsigned)(j / zvy), c = tile_band_coord((unsigned)(j % zvy), bd.ysplit, bd.ygap); d_sm[i] = L.COSF(((float)c*dyv + d_m0[nux + u])*dyi*freq);}
			});
		}
		float *d_taby = d_tab + (size_t)F_TABLE_SIZE*nxpv;
		if (md == MGEN_SINE) {
			be.launch(tab_floats, [=] TERRA_LAMBDA (size_t i) {
				bool const isx = i < (size_t)F_TABLE_SIZE*nxpv;
				size_t const j = isx ? i : i - (size_t)F_TABLE_SIZE*nxpv;
				unsigned const rowlen = isx ? nxpv : nypv, per = isx ? zvx : zvy, k = (unsigned)(j / rowlen), v = (unsigned)(j % rowlen), u = v / per;
				unsigned const c = isx ? tile_band_coord(v % per, bd.xsplit, bd.xgap) : tile_band_coord(v % per, bd.ysplit, bd.ygap); // the cell's coordinate in the tile's field
				float val = 0.0f; // zero padding up to a multiple of 128
				if (u < (isx ? nux : nuy)) {
					sine_k_t const &sk = d_sk[isx ? u : nux + u];
					val = isx ? L.SINF(sk.xmdx[k]*(float)c + sk.xconst[k]) : sk.yscale[k]*L.SINF(sk.ymdy[k]*(float)c + sk.yconst[k]);
				}
				d_tab[i] = val;
			});
		}
		float const sine_offset = hp.sine_bias*mesh_scale_z_inv;
		bool const plain = md == MGEN_SINE && sine_plain_only(shp, kstart);
		be.tile_grid(n, d_refs, nux, nuy, d_tab, d_taby, nxpv, nypv, d_sm, d_m0, md, shp, kstart, use_sm, sine_offset, nc, L, dxv, dyv, d_out, plain, banded ? zvx : tw, unique_tiles, glac, d_noise_lut, fused_kernel_exists(md, plain) ? opt.gen_fused : 0, sine_amp_max(kstart),
			banded ? &bd : nullptr);
		return d_refs;
	}

	static constexpr uint32_t AO_DIRS = 8, AO_STEPS = 8, AO_RAY_LEN = AO_STEPS*(AO_STEPS + 1)/2, AO_CTX = 129 + 2*AO_RAY_LEN; // src/tiled_mesh.cpp:41-43: 36, 201
	// enable_tiled_mesh_ao with the GL noise modes: create_zvals clips the zvals from the AO context grid (src/tiled_mesh.cpp:478-488,505)
	bool ao_context_zvals() const {return tiled_mesh_ao && mode >= MGEN_SIMPLEX_GPU;}

	// tiles from a heightmap texture (using_tiled_terrain_hmap_tex / using_hmap_with_detail, src/tiled_mesh.cpp:273-274,447-451)
	bool using_hmap() const {return hmap_pix != nullptr;}
	bool using_hmap_with_detail() const {return using_hmap() && mesh_scale < 0.75f;}
	hmap_view_t hmap_view() const {return hmap_view_t{hmap_pix, hmap_w, hmap_h, hmap_nc, mesh_scale, mesh_height_scale, mesh_file_scale, mesh_file_tz, mesh_scale_z_inv};}
	void set_mesh_height_scales_for_zval_range(float min_z, float dz) { // src/mesh_gen.cpp:125-131
		if (!(dz > 0.0f)) throw std::invalid_argument("set_mesh_height_scales_for_zval_range: dz must be > 0");
		float const READ_MESH_H_SCALE = 0.0008f;
		mesh_file_scale = dz/(READ_MESH_H_SCALE*mesh_height_scale*mesh_scale_z_inv);
		mesh_file_tz    = min_z/mesh_scale_z_inv;
	}
	// tw x tw field per tile sampled from the heightmap texture (+ HMAP_DETAIL_MAG * the detail noise grid when mesh_scale < 0.75): src/tiled_mesh.cpp:499-503,623-627
	tile_ref_pod_t const *tile_hmap_fields_dev(int32_t const *tile_xy, uint32_t n, uint32_t tw, int shift, float *d_out) {
		bool const add_detail = using_hmap_with_detail();
		tile_ref_pod_t const *d_refs = tile_fields_dev(tile_xy, n, tw, shift, d_out, add_detail ? 16.0f : 0.0f); // HMAP_DETAIL_SCALE (src/heightmap.h:8)
		hmap_view_t const hv = hmap_view();
		be.launch((size_t)n*tw*tw, [=] TERRA_LAMBDA (size_t i) {
			unsigned const t = (unsigned)(i / ((size_t)tw*tw)), p = (unsigned)(i % ((size_t)tw*tw)), y = p / tw, x = p % tw;
			tile_ref_pod_t const r = d_refs[t];
			float zval = hv.clamped_height(r.tx*128 - shift + (int)x, r.ty*128 - shift + (int)y);
			if (add_detail) {zval += 0.01f*d_out[i];} // HMAP_DETAIL_MAG (src/heightmap.h:9)
			d_out[i] = zval;
		});
		return d_refs;
	}

	void tiles_create_zvals_dev(int32_t const *tile_xy, uint32_t n, uint32_t iters_tt, float *d_zvals, terra_tile_stats *d_stats, uint8_t *d_normals, float *d_min_nz) {
		require_scene();
		if (n == 0) return;
		uint32_t const size = 128, zv = 130;
		tile_ref_pod_t const *d_refs;
		if (using_hmap()) {d_refs = tile_hmap_fields_dev(tile_xy, n, zv, 0, d_zvals); iters_tt = 0;} // "heightmap is eroded during load" (src/tiled_mesh.cpp:515)
		else if (ao_context_zvals()) {
			float *d_ctx = scratch<float>(s_ao, (size_t)n*AO_CTX*AO_CTX);
			d_refs = tile_fields_dev(tile_xy, n, AO_CTX, (int)AO_RAY_LEN, d_ctx);
			uint32_t const cs = AO_CTX, rl = AO_RAY_LEN;
			be.launch((size_t)n*zv*zv, [=] TERRA_LAMBDA (size_t i) {
				unsigned const t = (unsigned)(i / (zv*zv)), p = (unsigned)(i % (zv*zv)), y = p / zv, x = p % zv;
				d_zvals[i] = d_ctx[(size_t)t*cs*cs + (size_t)(y + rl)*cs + (x + rl)];
			});
		}
		else {d_refs = tile_fields_dev(tile_xy, n, zv, 0, d_zvals);}
		float const dxv = DX_VAL, dyv = DY_VAL;
		// erosion: every tile alone on its clamp-padded 138x138 copy, droplets in order (src/tiled_mesh.cpp:515)
		if (iters_tt > 0 && erode_amount > 0.0f) {
			check_erosion_iters(iters_tt);
			erosion_consts_t const ec = make_erosion_consts((int)zv, (int)zv, zmin);
			be.tile_erosion(n, d_zvals, ec, iters_tt);
		}
		// sub-block z ranges + water bbox (src/tiled_mesh.cpp:517-541) and normals (src/tiled_mesh.h:281-284, src/tiled_mesh.cpp:865-880)
		if (d_stats || d_normals) {
			float const wpz_max = get_max_sea_level();
			float const rad_c = (dxv*dxv + dyv*dyv)*size*size;
			be.tile_post(n, d_refs, d_zvals, d_stats, d_normals, d_min_nz, wpz_max, rad_c, dxv, dyv, dxdy);
		}
	}

	// the post pass alone over the caller's zvals (terra_tiles_post_dev)
	void tiles_post_dev(int32_t const *tile_xy, uint32_t n, float const *d_zvals, terra_tile_stats *d_stats, uint8_t *d_normals, float *d_min_nz) {
		require_scene();
		if (n == 0 || !(d_stats || d_normals)) return;
		tile_ref_pod_t const *d_refs = tile_fields_dev(tile_xy, n, 130, 0, nullptr, 0.0f); // (only the tile references)
		float const dxv = DX_VAL, dyv = DY_VAL, rad_c = (dxv*dxv + dyv*dyv)*128*128;
		be.tile_post(n, d_refs, d_zvals, d_stats, d_normals, d_min_nz, get_max_sea_level(), rad_c, dxv, dyv, dxdy);
	}

	// tile_t::calc_shadows_for_light + calc_mesh_shadows (src/tiled_mesh.cpp:664-692, src/visibility.cpp:510-520) for a batch and one directional light:
	// smask[n][130][130] gets the MESH_SHADOW bit.  A tile's sweeps start from the edge heights its two neighbours toward the light left behind
	// (sh_out -> sh_in), so the batch is processed in dependency levels (anti-diagonals); tiles of one level run in parallel, every sweep of a tile too.
	// edge_in / edge_in_present / edge_out (host, optional): the halo of a batch that is only part of the terrain (another GPU owns the rest).
	// edge_in[i][0] = sh_in_x, edge_in[i][1] = sh_in_y of tile i, used where present[i][d] != 0 and the neighbour toward the light is not in the batch;
	// edge_out[i][0..1] = the tile's sh_out_x / sh_out_y (MESH_MIN_Z where nothing was written), for the owner of the next tile away from the light.
	// d_edge_in / d_edge_out (optional): the same halo arrays in DEVICE memory ([n][2][130] floats; d_edge_in is used with the host flags edge_in_present): a
	// terrain spread over several GPUs of one process hands the border edges from device to device (hipMemcpyPeerAsync, terra_multi.hpp) without a host copy
	void tiles_mesh_shadows_dev(int32_t const *tile_xy, uint32_t n, float const *d_zvals, float const lpos[3], uint8_t *d_smask,
		float const *edge_in = nullptr, uint8_t const *edge_in_present = nullptr, float *edge_out = nullptr, float const *d_edge_in = nullptr, float *d_edge_out = nullptr)
	{
		require_scene();
		if (n == 0) return;
		uint32_t const zv = 130;
		if (edge_in && d_edge_in) throw std::invalid_argument("tiles_mesh_shadows: incoming edges either on the host or on the device");
		if (edge_out) {for (size_t i = 0; i < (size_t)n*2*zv; ++i) edge_out[i] = -1.0E6f;} // sh_out[l][d].resize(zvsize, MESH_MIN_Z)
		if (d_edge_out) {float *eo = d_edge_out; be.launch((size_t)n*2*zv, [=] TERRA_LAMBDA (size_t i) {eo[i] = -1.0E6f;});}
		float const lx = lpos[0], ly = lpos[1], lz = lpos[2];
		bool const all_shadowed = (lz < zmin);
		be.fill8(d_smask, all_shadowed ? 0x02 : 0x00, (size_t)n*zv*zv); // MESH_SHADOW (src/3DWorld.h:1403)
		if ((double)lx == 0.0 && (double)ly == 0.0) return; // straight down = no mesh shadows
		shadow_consts_t c;
		c.X_SCENE_SIZE = cfg.scene_x; c.Y_SCENE_SIZE = cfg.scene_y; c.DX_VAL = DX_VAL; c.DY_VAL = DY_VAL; c.DX_VAL_INV = DX_VAL_INV; c.DY_VAL_INV = DY_VAL_INV;
		c.zmin = zmin; c.zmax = zmax; c.xsize = (int)zv; c.ysize = (int)zv; c.mask_fill = all_shadowed ? 0x02020202u : 0u;
		float const lmag = sqrtf(lx*lx + ly*ly + lz*lz); // dir = -lpos.get_norm()
		if ((double)lmag < 1.0E-12) {c.dirx = -lx; c.diry = -ly; c.dirz = -lz;} else {c.dirx = -(lx/lmag); c.diry = -(ly/lmag); c.dirz = -(lz/lmag);}
		c.dist = (float)(2.0*(double)(cfg.mesh_x + cfg.mesh_y)/(double)sqrtf(c.dirx*c.dirx + c.diry*c.diry)); // 2.0*XY_SUM_SIZE/sqrt(...)
		// dependency levels
		int const sx = (lx < 0.0f) ? -1 : 1, sy = (ly < 0.0f) ? -1 : 1;
		// which entry of the batch a tile is: a table over the batch's bounding box where that is not much larger than the batch (the usual case, a block of tiles: the
		// ordered map cost more host time than the lookups are worth), the map otherwise; a tile listed twice is its last entry either way
		std::map<std::pair<int32_t, int32_t>, uint32_t> index;
		std::vector<int32_t> grid; int64_t gx0 = 0, gy0 = 0, gw = 0, gh = 0;
		{
			int64_t x0 = tile_xy[0], x1 = x0, y0 = tile_xy[1], y1 = y0;
			for (uint32_t i = 1; i < n; ++i) {int64_t const tx = tile_xy[2*i], ty = tile_xy[2*i+1]; x0 = std::min(x0, tx); x1 = std::max(x1, tx); y0 = std::min(y0, ty); y1 = std::max(y1, ty);}
			int64_t const w = x1 - x0 + 1, h = y1 - y0 + 1;
			if (w <= (1 << 20) && h <= (1 << 20) && w*h <= 4*(int64_t)n + 64) {
				gx0 = x0; gy0 = y0; gw = w; gh = h; grid.assign((size_t)(w*h), -1);
				for (uint32_t i = 0; i < n; ++i) {grid[(size_t)((tile_xy[2*i+1] - y0)*w + (tile_xy[2*i] - x0))] = (int32_t)i;}
			}
			else {for (uint32_t i = 0; i < n; ++i) {index[std::make_pair(tile_xy[2*i], tile_xy[2*i+1])] = i;}}
		}
		auto const entry_of = [&](int64_t tx, int64_t ty) -> int32_t {
			if (gw) {int64_t const ix = tx - gx0, iy = ty - gy0; return (ix >= 0 && iy >= 0 && ix < gw && iy < gh) ? grid[(size_t)(iy*gw + ix)] : -1;}
			auto const it = index.find(std::make_pair((int32_t)tx, (int32_t)ty));
			return (tx < INT32_MIN || tx > INT32_MAX || ty < INT32_MIN || ty > INT32_MAX || it == index.end()) ? -1 : (int32_t)it->second;
		};
		std::vector<int32_t> adj(2*(size_t)n, -1); // [i][0]: neighbour in x toward the light (its sh_out_y is our sh_in_y), [i][1]: neighbour in y
		std::vector<uint32_t> level(n, 0), order(n);
		for (uint32_t i = 0; i < n; ++i) {
			adj[2*i] = entry_of((int64_t)tile_xy[2*i] + sx, tile_xy[2*i+1]);
			adj[2*i+1] = entry_of(tile_xy[2*i], (int64_t)tile_xy[2*i+1] + sy);
		}
		// halo: a present incoming edge becomes a virtual neighbour slot n + k whose out-array holds the received heights
		std::vector<unsigned long long> virt; // [nvirt][zv]
		std::vector<std::pair<uint32_t, uint32_t>> virt_slot; // (which array: 0 = out_x, 1 = out_y ; slot)
		std::vector<uint32_t> virt_src;                        // device edges: index (tile*2 + d) of the incoming array a virtual slot is filled from
		if ((edge_in || d_edge_in) && edge_in_present) {
			for (uint32_t i = 0; i < n; ++i) {
				for (uint32_t d = 0; d < 2; ++d) { // d = 0: sh_in_x (from the y-neighbour, adj[2i+1]); d = 1: sh_in_y (from the x-neighbour, adj[2i])
					int32_t &a = adj[2*i + (1 - d)];
					if (a >= 0 || !edge_in_present[2*i + d]) continue;
					a = (int32_t)(n + (uint32_t)virt_slot.size());
					virt_slot.push_back(std::make_pair(d, (uint32_t)a));
					if (d_edge_in) {virt_src.push_back(2*i + d); continue;}
					float const *src = edge_in + ((size_t)i*2 + d)*zv;
					for (uint32_t e = 0; e < zv; ++e) {uint32_t b; memcpy(&b, &src[e], 4); virt.push_back((src[e] > -1.0E6f) ? ((1ull << 32) | b) : 0ull);}
				}
			}
		}
		uint32_t const nslots = n + (uint32_t)virt_slot.size();
		{ // levels by relaxation over tiles sorted toward the light (a tile's dependencies lie strictly further toward the light in x or y)
			for (uint32_t i = 0; i < n; ++i) order[i] = i;
			std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
				int64_t const ka = (int64_t)sx*tile_xy[2*a] + (int64_t)sy*tile_xy[2*a+1], kb = (int64_t)sx*tile_xy[2*b] + (int64_t)sy*tile_xy[2*b+1];
				return ka > kb;
			});
			for (uint32_t i : order) {
				uint32_t lv = 0;
				if (adj[2*i] >= 0 && (uint32_t)adj[2*i] < n) lv = std::max(lv, level[adj[2*i]] + 1);
				if (adj[2*i+1] >= 0 && (uint32_t)adj[2*i+1] < n) lv = std::max(lv, level[adj[2*i+1]] + 1);
				level[i] = lv;
			}
			std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {return level[a] < level[b];});
		}
		size_t const bytes = (size_t)n*4 + adj.size()*4 + (size_t)2*nslots*zv*8 + ((size_t)n + 1)*4 + 256*4 + ((size_t)nslots + 2)*4 + 256;
		uint8_t *base = scratch<uint8_t>(s_shadow, bytes);
		uint32_t *d_order = (uint32_t *)base;
		int32_t *d_adj = (int32_t *)(base + (((size_t)n*4 + 255) & ~(size_t)255));
		unsigned long long *d_out = (unsigned long long *)((uint8_t *)d_adj + ((adj.size()*4 + 255) & ~(size_t)255)); // [2][n][zv]: (order << 32) | float bits, 0 = never written
		be.h2d_async(d_order, order.data(), (size_t)n*4);
		be.h2d_async(d_adj, adj.data(), adj.size()*4);
		be.fill32(d_out, 0, (size_t)2*nslots*zv*2);
		if (d_edge_in && !virt_slot.empty()) { // encode on the device, ONE launch for all virtual slots: (1 << 32 | float bits) where a height was handed over, 0 = nothing
			std::vector<uint32_t> vmap(2*virt_slot.size()); // (destination row of d_out, source row of d_edge_in) per virtual slot
			for (size_t k = 0; k < virt_slot.size(); ++k) {vmap[2*k] = virt_slot[k].first*nslots + virt_slot[k].second; vmap[2*k+1] = virt_src[k];}
			uint32_t *d_vmap = scratch<uint32_t>(s_shadow_map, vmap.size());
			be.h2d_async(d_vmap, vmap.data(), vmap.size()*4);
			float const *ein = d_edge_in;
			be.launch(virt_slot.size()*zv, [=] TERRA_LAMBDA (size_t j) {
				uint32_t const k = (uint32_t)(j / zv), e = (uint32_t)(j % zv);
				float const v = ein[(size_t)d_vmap[2*k+1]*zv + e]; uint32_t b; memcpy(&b, &v, 4);
				d_out[(size_t)d_vmap[2*k]*zv + e] = (v > -1.0E6f) ? ((1ull << 32) | b) : 0ull;
			});
		}
		else {
			for (size_t k = 0; k < virt_slot.size(); ++k) {
				unsigned long long *dst = d_out + ((size_t)virt_slot[k].first*nslots + virt_slot[k].second)*zv;
				be.h2d_async(dst, virt.data() + k*zv, (size_t)zv*8);
			}
		}
		uint32_t const npaths = 4*zv;
		uint32_t *d_sync = (uint32_t *)((uint8_t *)d_out + (((size_t)2*nslots*zv*8 + 255) & ~(size_t)255)); // the ticket counter of the dataflow launch
		if (be.tile_shadows_flow(c, n, nslots, d_order, d_adj, d_zvals, d_out, d_smask, npaths, d_sync)) {} // one launch; tiles start as their two upstream tiles publish
		else for (uint32_t first = 0; first < n;) { // "shadows.levels" / cross-check kernels / the emulator: one launch per dependency level
			uint32_t last = first;
			while (last < n && level[order[last]] == level[order[first]]) ++last;
			be.tile_shadows(c, last - first, d_order + first, d_adj, nslots, d_zvals, d_out, d_smask, npaths);
			first = last;
		}
		if (d_edge_out) { // the tiles' own outgoing edges, decoded on the device
			float *eo = d_edge_out; uint32_t const ns = nslots;
			be.launch((size_t)n*2*zv, [=] TERRA_LAMBDA (size_t j) {
				uint32_t const e = (uint32_t)(j % zv), d = (uint32_t)((j / zv) % 2), i = (uint32_t)(j / (2*zv));
				unsigned long long const v = d_out[((size_t)d*ns + i)*zv + e] & ~(1ull << 63); // (bit 63: the dataflow launch's `published` mark)
				if (v != 0) {uint32_t const b = (uint32_t)(v & 0xFFFFFFFFull); float f; memcpy(&f, &b, 4); eo[j] = f;}
			});
		}
		if (edge_out) { // the tiles' own outgoing edges, decoded
			std::vector<unsigned long long> h((size_t)2*nslots*zv);
			be.d2h(h.data(), d_out, h.size()*8);
			for (uint32_t i = 0; i < n; ++i) {
				for (uint32_t d = 0; d < 2; ++d) {
					for (uint32_t e = 0; e < zv; ++e) {
						unsigned long long const v = h[((size_t)d*nslots + i)*zv + e] & ~(1ull << 63); // (bit 63: the dataflow launch's `published` mark)
						if (v != 0) {uint32_t const b = (uint32_t)(v & 0xFFFFFFFFull); memcpy(&edge_out[((size_t)i*2 + d)*zv + e], &b, 4);}
					}
				}
			}
		}
	}

	// tile_t::calc_mesh_ao_lighting (src/tiled_mesh.cpp:586-661): zvals as create_zvals left them -> 129 x 129 ambient-occlusion bytes per tile.
	// The 201 x 201 context is the tile's own zvals inside the tile (possibly eroded) and eval_index() of the context grid around it; with
	// enable_tiled_mesh_ao and a GL noise mode it is the context grid everywhere (ao_zvals kept by create_zvals).
	void tiles_ao_lighting_dev(int32_t const *tile_xy, uint32_t n, float const *d_zvals, uint8_t *d_ao) {
		require_scene();
		if (n == 0) return;
		uint32_t const zv = 130, cs = AO_CTX, rl = AO_RAY_LEN;
		float *d_ctx = scratch<float>(s_ao, (size_t)n*cs*cs);
		// inside the tile the context is the tile's own zvals (src/tiled_mesh.cpp:622): the kernel takes those cells from d_zvals while it stages the context (a copy pass
		// into d_ctx first was 168 us for 4096 tiles)
		bool const own = using_hmap() || !ao_context_zvals();
		if (using_hmap()) {tile_hmap_fields_dev(tile_xy, n, cs, (int)rl, d_ctx);}
		else {
			// ... so with `own` the centre of the context -- 130^2 of 201^2 cells, 42 % -- is never read: where the backend can, only the four bands around the tile are evaluated
			// (a cell's value does not depend on its neighbours: the same bits), as two launches: the rows above and below the tile, then the columns beside it
			tile_band_t const rows_band = {cs, cs - zv, cs, 0xFFFFFFFFu, 0u, rl, zv}, cols_band = {cs - zv, zv, cs, rl, zv, 0u, rl};
			bool banded = false;
			tile_fields_dev(tile_xy, n, cs, (int)rl, d_ctx, 1.0f, true, false, 0, (own && opt.ao_bands) ? &rows_band : nullptr, &banded);
			if (banded) {tile_fields_dev(tile_xy, n, cs, (int)rl, d_ctx, 1.0f, true, false, 0, &cols_band, &banded);}
		}
		float const dz = (float)(0.5*(double)HALF_DXY);
		be.tile_ao(n, d_zvals, d_ctx, d_ao, dz, own);
	}

	// ================================================================ height edits of the heightmap texture and the map exporter (rest of f4)
	// hmap_brush_t::apply for a list of brushes in order (src/heightmap.cpp:36-58, apply_cur_brushes :438-440) on the image set by terra_hmap_set_dev.
	// One launch per brush, one thread per brush point (yp, xp, sy, sx); see modify_pixel for why the threads may arrive in any order.
	void hmap_apply_brushes_dev(hmap_brush_pod_t const *brushes, uint32_t n, int step_sz, uint32_t num_steps) {
		require_scene();
		if (!using_hmap()) throw std::logic_error("terra_hmap_apply_brushes_dev: no heightmap texture (terra_hmap_set_dev)");
		if (step_sz <= 0 || num_steps == 0) throw std::invalid_argument("hmap brush: step_sz and num_steps must be > 0");
		if (std::max(hmap_w, hmap_h) > 65536) throw std::invalid_argument("hmap brush: image larger than max_tex_ix() = 65536 (src/heightmap.h:42)");
		hmap_view_t const hv = hmap_view();
		uint8_t *pix = const_cast<uint8_t *>(hmap_pix);
		sin_lut_t const L = lut();
		for (uint32_t bi = 0; bi < n; ++bi) { // validate the whole list before the first edit
			hmap_brush_pod_t const &b = brushes[bi];
			if (b.shape < 0 || b.shape >= NUM_BSHAPES) throw std::invalid_argument("hmap brush: bad shape");
			if (b.radius > (1u << 20)) throw std::invalid_argument("hmap brush: radius too large");
			uint64_t const side = (uint64_t)(2*b.radius)/(uint64_t)step_sz + 1;
			if (side*side*num_steps*num_steps > (1ull << 32)) throw std::invalid_argument("hmap brush: more than 2^32 brush points");
		}
		for (uint32_t bi = 0; bi < n; ++bi) {
			hmap_brush_pod_t const b = brushes[bi];
			int const r = (int)b.radius, shape = b.shape, bx = b.x, by = b.y, delta = b.delta, step = step_sz;
			uint32_t const side = (uint32_t)(2*r)/(uint32_t)step_sz + 1, ns = num_steps;
			float const step_delta = (float)(1.0/(double)num_steps), r_inv = (float)(1.0/(double)std::max(1u, b.radius));
			bool const is_delta = !(shape == BSHAPE_FLAT_SQ || shape == BSHAPE_FLAT_CIR);
			be.launch((size_t)side*side*ns*ns, [=] TERRA_LAMBDA (size_t i) {
				unsigned const sx = (unsigned)(i % ns), sy = (unsigned)((i / ns) % ns);
				size_t const c = i / ((size_t)ns*ns);
				int const xp = bx - r + (int)(c % side)*step, yp = by - r + (int)(c / side)*step;
				float const dx = (float)sx*step_delta, dy = (float)sy*step_delta;
				float const ey = ((float)yp + dy) - (float)by, ex = ((float)xp + dx) - (float)bx;
				float const dist = sqrtf(ey*ey + ex*ex), dval = dist*r_inv;
				if (shape != BSHAPE_CONST_SQ && shape != BSHAPE_FLAT_SQ && (double)dval > 1.0) return; // round (instead of square)
				float mod_delta = (float)delta; // adjust_brush_weight (src/heightmap.cpp:27-33)
				float const PI_F = 3.141592654f;
				if      (shape == BSHAPE_LINEAR   ) {mod_delta *= 1.0f - dval;}
				else if (shape == BSHAPE_QUADRATIC) {mod_delta *= 1.0f - dval*dval;}
				else if (shape == BSHAPE_COSINE   ) {mod_delta *= L.COSF(0.5f*PI_F*dval);}
				else if (shape == BSHAPE_SINE     ) {mod_delta *= 0.5f*(1.0f + L.SINF(PI_F*dval + 0.5f*PI_F));}
				int x = xp, y = yp; // modify_height_value (src/tiled_mesh.cpp:259-266)
				hv.clamp_xy(x, y, dx, dy);
				modify_pixel(pix, hv.ncolors, (size_t)hv.width*(unsigned)y + (unsigned)x, hmap_view_t::round_fp(mod_delta), is_delta);
			});
		}
	}
	// add_mod for every element + apply_cur_mod_map (src/heightmap.cpp:216-222,431-436)
	void hmap_apply_mods_dev(hmap_mod_pod_t const *mods, uint32_t n) {
		require_scene();
		if (!using_hmap()) throw std::logic_error("terra_hmap_apply_mods_dev: no heightmap texture (terra_hmap_set_dev)");
		std::vector<hmap_mod_pod_t> const m = combine_mods(mods, n);
		for (hmap_mod_pod_t const &e : m) {if ((int)e.x >= hmap_w || (int)e.y >= hmap_h) throw std::invalid_argument("hmap mod outside the texture");}
		if (m.empty()) return;
		hmap_mod_pod_t *d_m = scratch<hmap_mod_pod_t>(s_misc, m.size());
		be.h2d(d_m, m.data(), m.size()*sizeof(hmap_mod_pod_t));
		uint8_t *pix = const_cast<uint8_t *>(hmap_pix);
		int const nc = hmap_nc, w = hmap_w;
		be.launch(m.size(), [=] TERRA_LAMBDA (size_t i) {modify_pixel(pix, nc, (size_t)w*d_m[i].y + d_m[i].x, d_m[i].delta, true);});
		be.sync();
	}
	void hmap_read_and_apply_mod_dev(char const *fn) { // terrain_hmap_manager_t::read_and_apply_mod (src/heightmap.cpp:424-429)
		std::vector<hmap_mod_pod_t> mods; std::vector<hmap_brush_pod_t> brushes;
		read_mod_file(fn, mods, brushes);
		hmap_apply_mods_dev(mods.data(), (uint32_t)mods.size());
		hmap_apply_brushes_dev(brushes.data(), (uint32_t)brushes.size(), 1, 1);
	}
	float get_xy_scale() const {bool const add_detail = using_hmap_with_detail(); if (!add_detail && using_hmap()) return 0.0f; return add_detail ? 16.0f : 1.0f;} // src/tiled_mesh.cpp:447-451
	// write_map_mode_heightmap_image (src/map_view.cpp:409-442) from the image origin on: heights (rows inverted, as the reference's `heights`) into d_vals,
	// 16-bit pixels = (h - min_z)*(255/dz) into d_pix; h_range = {min_z, dz}
	void export_heightmap_dev(float xstart, float ystart, uint32_t width, uint32_t height, float *d_vals, uint8_t *d_pix, float *h_range) {
		require_scene();
		if (width == 0 || height == 0) throw std::invalid_argument("export_heightmap: empty image");
		if (width > 16384 || height > 16384) throw std::invalid_argument("export_heightmap: heightmap image is too large, max size is 16384 pixels"); // src/map_view.cpp:417
		size_t const n = (size_t)width*height;
		float const xy_scale = get_xy_scale();
		if (xy_scale != 0.0f) {gen_grid_dev(xstart/DX_VAL, ystart/DY_VAL, xy_scale*DX_VAL, xy_scale*DY_VAL, width, height, TERRA_GEN_GLACIATE | TERRA_GEN_CACHE_VALUES, 0, d_vals);} // setup_height_gen_cached
		if (using_hmap()) { // get_mesh_height (src/map_view.cpp:97-105), nearest_texel = 0
			hmap_view_t const hv = hmap_view();
			bool const detail = using_hmap_with_detail();
			float const xs = xstart + cfg.scene_x, ys = ystart + cfg.scene_y, dxv = DX_VAL, dyv = DY_VAL, dxi = DX_VAL_INV, dyi = DY_VAL_INV;
			be.launch(n, [=] TERRA_LAMBDA (size_t k) {
				unsigned const i = (unsigned)(k / width), j = (unsigned)(k % width);
				float zval = hv.interpolate_height((xs + (float)j*dxv)*dxi, (ys + (float)i*dyv)*dyi);
				if (detail) {zval += 0.01f*d_vals[k];} // HMAP_DETAIL_MAG
				d_vals[k] = zval;
			});
		}
		be.launch((size_t)(height/2)*width, [=] TERRA_LAMBDA (size_t k) { // invert yval
			size_t const i = k / width, j = k % width, a = i*width + j, b = (size_t)(height - 1 - i)*width + j;
			float const t = d_vals[a]; d_vals[a] = d_vals[b]; d_vals[b] = t;
		});
		float mn, mx; minmax_dev(d_vals, n, mn, mx); // get_heightmap_z_range
		float const dz = max_std(1.0E-12f, (mx - mn)), height_scale = (float)(255.0/(double)dz);
		if (h_range) {h_range[0] = mn; h_range[1] = dz;}
		if (d_pix) {
			be.launch(n, [=] TERRA_LAMBDA (size_t i) { // write_pixel_16_bits (src/Textures.cpp:1889-1893)
				float const v = (d_vals[i] - mn)*height_scale;
				uint8_t const hi = (uint8_t)v;
				d_pix[(i<<1)+1] = hi;
				d_pix[i<<1]     = (uint8_t)(256.0f*(v - (float)hi));
			});
		}
	}

	// ================================================================ landscape weights texture (f3)
	// tile_t::update_terrain_params (src/tiled_mesh.cpp:321-343): out[tile][yp][xp] = {veg, grass, dirt}
	void tiles_terrain_params_dev(tile_ref_pod_t const *d_refs, uint32_t n, float *d_params) {
		if (!ls.enable_terrain_env) { // terrain_params_t defaults (src/tiled_mesh.h:193)
			be.launch((size_t)n*12, [=] TERRA_LAMBDA (size_t i) {d_params[i] = ((i % 3) == 2) ? 0.0f : 1.0f;});
			return;
		}
		float *d_st = scratch<float>(s_misc, F_TABLE_SIZE*5 + 16);
		be.h2d(d_st, &sinTable[0][0], sizeof(sinTable));
		sin_lut_t const L = lut();
		int const k0 = start_eval_sin;
		float const msc = mesh_scale, bxo = ls.biome_x_offset, dxv = DX_VAL, dyv = DY_VAL, xss = cfg.scene_x, yss = cfg.scene_y;
		be.launch((size_t)n*8, [=] TERRA_LAMBDA (size_t i) { // (tile, corner, {veg, dirt})
			unsigned const t = (unsigned)(i >> 3), yp = (unsigned)(i >> 2) & 1u, xp = (unsigned)(i >> 1) & 1u, which = (unsigned)i & 1u;
			tile_ref_pod_t const r = d_refs[t];
			int const x1 = r.tx*128, y1 = r.ty*128;
			float const xv1 = -xss + dxv*(float)x1, xv2 = xv1 + (float)128*dxv, yv1 = -yss + dyv*(float)y1, yv2 = yv1 + (float)128*dyv; // get_xval(x1), xv1 + (x2-x1)*DX_VAL
			float const xv = msc*(xp ? xv2 : xv1) + bxo, yv = msc*(yp ? yv2 : yv1);
			float const mult = which ? 1.0f : 5.0f; // dirt_mult, veg_mult
			float const ax = mult*xv, ay = mult*yv;
			float zval = 0.0f; // eval_mesh_sin_terms (src/mesh_gen.cpp:797-805); the terms are independent up to the final chain of adds: eight of them (their table look-ups) in flight at once
#pragma unroll 8
			for (int k = k0; k < F_TABLE_SIZE; ++k) {
				float const *stk = d_st + 5*k;
				zval += stk[0]*L.SINF(stk[3]*ay + stk[1])*L.SINF(stk[4]*ax + stk[2]);
			}
			float *o = d_params + (size_t)t*12 + 3*(2*yp + xp);
			if (which) {o[2] = clip01(5.0f*(zval + 1.0f));}
			else {o[0] = clip01(5.000f*(zval + 1.5f)); o[1] = clip01(100.0f*(zval + 3.0f));}
		});
	}
	void tiles_terrain_params(int32_t const *tile_xy, uint32_t n, float *h_params) {
		require_scene();
		if (n == 0) return;
		tile_ref_pod_t const *d_refs = tile_fields_dev(tile_xy, n, WT_TEX, 0, nullptr, 0.0f);
		float *d_params = scratch<float>(s_ao, (size_t)n*12);
		tiles_terrain_params_dev(d_refs, n, d_params);
		be.d2h(h_params, d_params, (size_t)n*12*sizeof(float));
	}
	landscape_consts_t landscape_consts() const {
		landscape_consts_t c;
		tex_heights(c.h_dirt);
		c.zmin = zmin; c.dz_inv = 1.0f/(zmax - zmin); c.relh_adj_tex = relh_adj_tex; c.water_level = get_water_z_height();
		float const MESH_NOISE_SCALE = 0.003f;
		c.noise_scale = (float)(((shape == 2) ? 2.0 : 1.0)*(double)MESH_NOISE_SCALE*(double)ls.mesh_scale_z); // more noise for ridged
		c.vnz_scale = (mode == MGEN_DWARP_GPU) ? (float)sqrt(2.0) : 1.0f; // steeper slopes are allowed under domain warping
		c.DX_VAL = DX_VAL; c.DY_VAL = DY_VAL; c.dxdy = dxdy;
		c.steep_mult_grass = 1.0f/(sthresh_v(0, 1) - sthresh_v(0, 0));
		c.steep_mult_snow  = 1.0f/(sthresh_v(1, 1) - sthresh_v(1, 0));
		c.steep_mult_rock  = 1.0f/(0.8f*sthresh_v(0, 0) - 0.5f*sthresh_v(0, 0));
		c.vegetation = ls.vegetation; c.snow_to_rock = (ls.water_is_lava || ls.disable_water == 2) ? 1 : 0;
		c.gen_grass_map = (ls.grass_density > 0 && ls.vegetation > 0.0f) ? 1 : 0; // GRASS_THRESH = 1.6 > 0 (src/tiled_mesh.cpp:29,126)
		c.num_rnd_grass_blocks = ls.num_rnd_grass_blocks;
		return c;
	}
	// tile_t::create_texture (src/tiled_mesh.cpp:1071-1240), terrain-only branch, for a batch of tiles whose zvals are on the device:
	// d_weights n x 129x129 RGBA8, d_blocks n x 32x32 grass blocks (or null), d_has_grass n bytes (or null)
	void tiles_create_weights_dev(int32_t const *tile_xy, uint32_t n, float const *d_zvals, uint8_t *d_weights, grass_block_pod_t *d_blocks, uint8_t *d_has_grass) {
		require_scene();
		if (n == 0) return;
		uint32_t const ts = WT_TEX, zv = WT_ZV;
		size_t const ntex = (size_t)n*ts*ts;
		uint8_t *base = scratch<uint8_t>(s_ao, ntex*5 + (size_t)n*(12*4 + 1) + 1024);
		float *d_rand = (float *)base, *d_params = d_rand + ntex;
		uint8_t *d_flags = (uint8_t *)(d_params + (size_t)n*12), *d_any = d_flags + ntex;
		// second noise field: build_arrays(x1 - MESH_X_SIZE/2, y1 - MESH_Y_SIZE/2, 80*DX_VAL, 80*DY_VAL, tsize, tsize, 0, force_sine_mode=1) + eval_index(x, y, 50)
		tile_ref_pod_t const *d_refs = tile_fields_dev(tile_xy, n, ts, 0, d_rand, 80.0f, false, true, 50);
		tiles_terrain_params_dev(d_refs, n, d_params);
		landscape_consts_t const c = landscape_consts();
		be.fill8(d_any, 0, n);
		uint32_t *d_w32 = (uint32_t *)d_weights; // n*129*129*4 bytes from a device allocation: 4-byte aligned
		if (((uintptr_t)d_weights & 3u) != 0) throw std::invalid_argument("tiles_create_weights: d_weights must be 4-byte aligned");
		if (!opt.weights_simple && be.tile_weights(c, d_refs, n, d_zvals, d_rand, d_params, d_w32, d_blocks, d_any)) { // k_tile_weights: texels + grass blocks in one launch
			if (d_has_grass) {be.launch(n, [=] TERRA_LAMBDA (size_t i) {d_has_grass[i] = d_any[i];});}
			return;
		}
		be.launch(ntex, [=] TERRA_LAMBDA (size_t i) { // the per-texel form ("weights.simple": cross-check; the emulator)
			unsigned const t = (unsigned)(i / (ts*ts)), p = (unsigned)(i % (ts*ts)), y = p / ts, x = p % ts;
			unsigned flags;
			d_w32[i] = weights_texel(c, d_zvals + (size_t)t*zv*zv, d_params + (size_t)t*12, d_rand[i], x, y, flags);
			d_flags[i] = (uint8_t)flags;
			if (flags & 1u) {d_any[t] = 1;} // has_any_grass: every writer stores the same value
		});
		if (d_blocks) {
			uint32_t const bd = GRASS_BLOCK_DIM;
			be.launch((size_t)n*bd*bd, [=] TERRA_LAMBDA (size_t i) {
				unsigned const t = (unsigned)(i / (bd*bd)), b = (unsigned)(i % (bd*bd));
				tile_ref_pod_t const r = d_refs[t];
				d_blocks[i] = grass_block(c, d_zvals + (size_t)t*zv*zv, d_flags + (size_t)t*ts*ts, r.tx*128, r.ty*128, b % bd, b / bd);
			});
		}
		if (d_has_grass) {be.launch(n, [=] TERRA_LAMBDA (size_t i) {d_has_grass[i] = d_any[i];});}
	}

	// ================================================================ voxels (a14, a15, K8, K9)
	// y0 / nys (optional): only the y slab [y0, y0 + nys) of the nx x ny x nz grid, written to d_out as an nys x nx x nz array with the values the full-grid
	// call produces (the axis positions are the full grid's running sums): voxels are independent, a field is split over GPUs in y slabs
	void voxel_fill_dev(float *d_out, uint32_t nx, uint32_t ny, uint32_t nz, float const lo[3], float const vsz[3], float const off[3],
		float mag, float freq, int rs1, int rs2, int gen_mode, float zscale, int normalize, uint32_t y0 = 0, uint32_t nys = 0xFFFFFFFFu)
	{
		require_scene();
		if (nx == 0 || ny == 0 || nz == 0) throw std::invalid_argument("voxel_fill: empty grid");
		if (nys == 0xFFFFFFFFu) {if (y0 != 0) throw std::invalid_argument("voxel_fill slab: y0 without a slab height"); nys = ny;}
		if (nys == 0 || y0 >= ny || nys > ny - y0) throw std::invalid_argument("voxel_fill slab: [y0, y0 + nys) must be a non-empty range inside the grid");
		size_t const nvox = (size_t)nx*nys*nz;
		sin_lut_t const L = lut();
		if (gen_mode == MGEN_SINE) {
			// noise_gen_3d::gen_sines (src/upsurface.cpp:16-38) on the host: 420 floats, handed to the table kernel by value (kernel argument, no upload)
			struct vox_rdata_t {float v[VOX_SINES*VOX_PARAMS];} rd;
			rand_gen_t r; r.set_state(rs1, rs2);
			float m = mag, f = freq;
			for (unsigned i = 0; i < 5; ++i) {
				for (unsigned j = 0; j < 12; ++j) {
					float *p = &rd.v[VOX_PARAMS*(12*i + j)];
					p[0] = r.rand_uniform(0.2f, 1.0f)*m;
					p[1] = r.rand_uniform(0.1f, 1.0f)*f; p[2] = (float)(r.randd()*(double)two_pi);
					p[3] = r.rand_uniform(0.1f, 1.0f)*f; p[4] = (float)(r.randd()*(double)two_pi);
					p[5] = r.rand_uniform(0.1f, 1.0f)*f; p[6] = (float)(r.randd()*(double)two_pi);
				}
				m *= 0.5f; f /= 0.4f; // M_ATTEN_FACTOR, F_ATTEN_FACTOR (src/upsurface.cpp:10-11)
			}
			size_t const ntab = ((size_t)nx + nys + nz)*VOX_SINES, npos = (size_t)nx + ny + nz; // tables of the slab's rows only, positions of the whole axes
			uint32_t const nzp = (nz + 63u) & ~63u; // the z table once more as [z/8][k][z%8]: what the lane-per-column kernel reads as scalar operands, eight z at a time
			float *d_base = scratch<float>(s_vox, ntab + npos + 64 + (size_t)nzp*VOX_SINES);
			float *d_tab = d_base, *d_pos = d_base + ntab, *d_zt = d_base + ((ntab + npos + 63) & ~(size_t)63);
			// gen_xyz_vals (src/upsurface.cpp:41-57): val accumulates `val += step` sequentially, so the positions of an axis are a serial prefix sum: one thread per axis
			float const s0 = lo[0] + off[0], s1 = lo[1] + off[1], s2 = lo[2] + off[2], v0 = vsz[0], v1 = vsz[1], v2 = vsz[2];
			be.launch(3, [=] TERRA_LAMBDA (size_t d) {
				float val = (d == 0) ? s0 : ((d == 1) ? s1 : s2);
				float const step = (d == 0) ? v0 : ((d == 1) ? v1 : v2);
				uint32_t const cnt = (d == 0) ? nx : ((d == 1) ? ny : nz);
				float *o = d_pos + ((d == 0) ? (size_t)0 : ((d == 1) ? (size_t)nx : (size_t)nx + ny));
				uint32_t i = 0;
				for (; i + 8 <= cnt; i += 8) { // (the chain of adds is the work: eight per trip of the loop instead of one -- 17 us -> a few for a 512-cell axis)
#pragma unroll
					for (uint32_t e = 0; e < 8; ++e) {o[i + e] = val; val += step;}
				}
				for (; i < cnt; ++i) {o[i] = val; val += step;}
			});
			be.launch(ntab, [=] TERRA_LAMBDA (size_t i) {
				size_t const e = i / VOX_SINES; unsigned const k = (unsigned)(i % VOX_SINES);
				unsigned const d = (e < nx) ? 0u : ((e < (size_t)nx + nys) ? 1u : 2u);
				size_t const pe = (d == 0) ? e : ((d == 1) ? e + y0 : e + (ny - nys)); // index into the whole-axis positions: x as is, y shifted by the slab start, z after all of y
				unsigned const index2 = VOX_PARAMS*k + 2*d;
				float v = L.SINF(rd.v[index2+1]*d_pos[pe] + rd.v[index2+2]);
				if (d == 0) {v *= rd.v[index2];}
				d_tab[i] = v;
				if (d == 2) {size_t const zz = e - nx - nys; d_zt[((zz >> 3)*VOX_SINES + k)*8 + (zz & 7)] = v;} // [z/8][k][z%8]; (entries behind nz are never stored to the field: left as they are)
			});
			float amax = 0.0f; // the largest magnitude p[0] of gen_sines: bounds xv (and xv*yv)
			for (unsigned k = 0; k < VOX_SINES; ++k) {float const a = fabsf(rd.v[VOX_PARAMS*k]); if (a > amax) {amax = a;}}
			be.voxel_sines(d_out, nx, nys, nz, d_tab, zscale, normalize, opt.gen_fused, amax, d_zt, nzp);
		}
		else {
			float const l0 = lo[0], l1 = lo[1], l2 = lo[2], v0 = vsz[0], v1 = vsz[1], v2 = vsz[2], o0 = off[0], o1 = off[1], o2 = off[2];
			float const frx = rx, fry = ry;
			int const nn = imax(1, 5 - cfg.mesh_freq_filter); // MAX_FREQ_BINS - mesh_freq_filter (src/voxels.cpp:332)
			bool const perlin = (gen_mode == MGEN_PERLIN);
			vox_noise_job_t J;
			J.l0 = l0; J.l1 = l1; J.l2 = l2; J.v0 = v0; J.v1 = v1; J.v2 = v2; J.o0 = o0; J.o1 = o1; J.o2 = o2; J.frx = frx; J.fry = fry; J.mag = mag; J.freq = freq; J.zscale = zscale;
			J.nn = nn; J.normalize = normalize; J.nx = nx; J.nz = nz; J.y0 = y0;
			be.voxel_noise(d_out, nvox, J, perlin, opt.gen_fused != 0, d_noise3_lut + (perlin ? NOISE3_LUT_DWORDS : 0u));
		}
	}
};

} // namespace terra
