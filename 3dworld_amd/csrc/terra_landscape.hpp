// Landscape weights texture of a tile: tile_t::create_texture (src/tiled_mesh.cpp:1071-1240) with update_terrain_params (:321-343), get_tids /
// update_lttex_ix (src/Textures.cpp:1289-1316) and add_grass_block_at (src/tiled_mesh.cpp:1354-1371).  Terrain-only branch: the city / tunnel / building
// queries and the tree map belong to subsystems outside the path.  Per-texel pure function of the tile's zvals, a second sine-noise field and four
// biome parameters per tile corner -> RGBA8 {sand, dirt, grass, rock} (snow is the remainder), plus the 4x4-texel grass blocks' z ranges.
// Same evaluation order and float/double promotions as the reference statements (x86-64 SSE2, no FMA).
#pragma once
#include "terra_common.hpp"

namespace terra {

enum {LT_SAND = 0, LT_DIRT = 1, LT_GROUND = 2, LT_ROCK = 3, LT_SNOW = 4}; // mesh_tids_dirt order (src/mesh_gen.cpp:42): the index identifies the texture
constexpr uint32_t WT_SIZE = 128, WT_TEX = 129, WT_ZV = 130, GRASS_BLOCK_SZ = 4, GRASS_BLOCK_DIM = 1 + (WT_SIZE - 1)/GRASS_BLOCK_SZ; // src/grass.h:10, src/tiled_mesh.h:315

struct landscape_consts_t {
	float h_dirt[5];
	float zmin, dz_inv, relh_adj_tex, water_level, noise_scale, vnz_scale, DX_VAL, DY_VAL, dxdy;
	float steep_mult_grass, steep_mult_snow, steep_mult_rock, vegetation;
	int snow_to_rock; // water_is_lava || DISABLE_WATER == 2
	int gen_grass_map; uint32_t num_rnd_grass_blocks;
};
TERRA_HD float sthresh_v(int snow, int hi) {return snow ? (hi ? 0.72f : 0.48f) : (hi ? 0.86f : 0.68f);} // sthresh[2][2], src/mesh_gen.cpp:44

TERRA_HD void update_lttex_ix(landscape_consts_t const &c, int &ix) { // src/Textures.cpp:1289-1292
	if (c.snow_to_rock && ix == LT_SNOW) {--ix;}
	if (c.vegetation == 0.0f && ix == LT_GROUND) {++ix;}
}
TERRA_HD void get_tids(landscape_consts_t const &c, float relh, int &k1, int &k2, float *t) { // src/Textures.cpp:1294-1316
	float const TEXTURE_SMOOTH = 0.01f; // src/Textures.cpp:12
	if      (relh < c.h_dirt[0]) {k1 = 0;}
	else if (relh < c.h_dirt[1]) {k1 = 1;}
	else if (relh < c.h_dirt[2]) {k1 = 2;}
	else if (relh < c.h_dirt[3]) {k1 = 3;}
	else                         {k1 = 4;}
	float const hd = c.h_dirt[(k1 < 4) ? k1 : 3];
	if (k1 < 4 && (hd - relh) < TEXTURE_SMOOTH) {
		if (t) {*t = (float)(1.0 - (double)((hd - relh)/TEXTURE_SMOOTH));}
		k2 = k1 + 1;
		update_lttex_ix(c, k1);
		update_lttex_ix(c, k2);
	}
	else {
		update_lttex_ix(c, k1);
		k2 = k1;
	}
}
TERRA_HD float bilinear_param(float const *prm, int var, float x, float y) { // BILINEAR_INTERP (src/tiled_mesh.cpp:189), prm[yp][xp][{veg, grass, dirt}]
	float const a00 = prm[var], a01 = prm[3 + var], a10 = prm[6 + var], a11 = prm[9 + var];
	return y*(x*a11 + (1.0f - x)*a10) + (1.0f - y)*(x*a01 + (1.0f - x)*a00);
}
TERRA_HD float weight_add(float w, double v) {return (float)((double)w + v);} // "float += double expression"
TERRA_HD uint32_t weight_to_u8(float w) {return ((double)w <= 0.01) ? 0u : (((double)w >= 0.99) ? 255u : (uint32_t)(unsigned char)(255.0*(double)w));}

// one texel (x, y) of the 129x129 weights texture; returns RGBA packed little-endian.  flags: bit 0 = grass (has_any_grass), bit 1 = contributes to its grass block
TERRA_HD uint32_t weights_texel(landscape_consts_t const &c, float const *zvals /*130x130*/, float const *prm /*12*/, float rand_val, unsigned x, unsigned y, unsigned &flags) {
	unsigned const ix = y*WT_ZV + x;
	float weights[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
	float const mh00 = zvals[ix], mh01 = zvals[ix+1], mh10 = zvals[ix+WT_ZV], mh11 = zvals[ix+WT_ZV+1];
	float const mhmin = min_std(min_std(mh00, mh01), min_std(mh10, mh11)), mhmax = max_std(max_std(mh00, mh01), max_std(mh10, mh11));
	float const rand_offset = c.noise_scale*rand_val;
	float const relh1 = c.relh_adj_tex + (mhmin - c.zmin)*c.dz_inv + rand_offset, relh2 = c.relh_adj_tex + (mhmax - c.zmin)*c.dz_inv + rand_offset;
	int k1, k2, k3, k4;
	get_tids(c, relh1, k1, k2, nullptr);
	get_tids(c, relh2, k3, k4, nullptr);
	bool const same_tid = (k1 == k4);
	float t = 0.0f;
	k2 = k4;
	if (!same_tid) {
		float const relh = c.relh_adj_tex + (mh00 - c.zmin)*c.dz_inv;
		get_tids(c, relh, k1, k2, &t);
	}
	float weight_scale = 1.0f;
	bool const grass = (k1 == LT_GROUND || k2 == LT_GROUND), snow = (k2 == LT_SNOW);
	flags = grass ? 1u : 0u;
	if (grass || snow) {
		float const st0 = sthresh_v(snow, 0), st1 = sthresh_v(snow, 1);
		float const nx = c.DY_VAL*(mh00 - mh01), ny = c.DX_VAL*(mh00 - mh10), nz = c.dxdy; // get_norm_not_normalized (src/tiled_mesh.h:281)
		float vnz = c.vnz_scale*nz/sqrtf(nx*nx + ny*ny + nz*nz);
		if (grass && vnz > st1) {vnz = clip01(1.0f + 20.0f*rand_offset);}
		if (vnz < st1) { // steep slopes: dirt / rock replaces grass, rock replaces snow
			if (grass) {
				float rock_weight = (k1 == LT_GROUND || k2 == LT_ROCK) ? t : 0.0f;
				float const steepness = (float)(1.0 - (double)clip01((vnz - 0.5f*st0)*c.steep_mult_rock));
				rock_weight  = (float)((double)rock_weight*(1.0 - (double)steepness) + (double)steepness);
				weight_scale = clip01((vnz - st0)*c.steep_mult_grass);
				weights[LT_ROCK] = weight_add(weights[LT_ROCK], (1.0 - (double)weight_scale)*(double)rock_weight);
				weights[LT_DIRT] = weight_add(weights[LT_DIRT], (1.0 - (double)weight_scale)*(1.0 - (double)rock_weight));
			}
			else {
				weight_scale = clip01(2.0f*(vnz - st0)*c.steep_mult_snow);
				weights[LT_ROCK] = weight_add(weights[LT_ROCK], 1.0 - (double)weight_scale);
			}
		}
	}
	weights[k2] += weight_scale*t;
	weights[k1] = weight_add(weights[k1], (double)weight_scale*(1.0 - (double)t));
	float const xy_mult = 1.0f/128.0f, xv = (float)x*xy_mult, yv = (float)y*xy_mult;
	if (c.vegetation > 0.0f) { // convert dirt to sand only when there is vegetation
		float const dirt_scale = bilinear_param(prm, 2, xv, yv);
		if (dirt_scale < 1.0f) {
			weights[LT_SAND] = weight_add(weights[LT_SAND], (1.0 - (double)dirt_scale)*(double)weights[LT_DIRT]);
			weights[LT_DIRT] *= dirt_scale;
		}
	}
	if (grass) {
		float const grass_scale = (mhmin < c.water_level) ? 0.0f : bilinear_param(prm, 1, xv, yv); // no grass under water
		if (grass_scale < 1.0f) { // convert grass to sand
			float const gscale = clip01(2.5f*(grass_scale - 0.5f) + 0.5f);
			weights[LT_SAND]   = weight_add(weights[LT_SAND], (1.0 - (double)gscale)*(double)weights[LT_GROUND]);
			weights[LT_GROUND] *= gscale;
		}
		if (grass_scale > 0.0f && c.gen_grass_map && x < WT_SIZE && y < WT_SIZE) {flags |= 2u;}
	}
	return weight_to_u8(weights[0]) | (weight_to_u8(weights[1]) << 8) | (weight_to_u8(weights[2]) << 16) | (weight_to_u8(weights[3]) << 24);
}

struct grass_block_pod_t {uint32_t ix; float zmin, zmax;}; // tile_t::grass_block_t (src/tiled_mesh.h:186); ix 0 = unused
// add_grass_block_at over one block's 4x4 texels in the reference's row-major order: the first contributing texel picks ix
TERRA_HD grass_block_pod_t grass_block(landscape_consts_t const &c, float const *zvals, uint8_t const *flags /*129x129*/, int x1, int y1, unsigned bx, unsigned by) {
	grass_block_pod_t gb = {0u, 0.0f, 0.0f};
	for (unsigned y = by*GRASS_BLOCK_SZ; y < (by + 1)*GRASS_BLOCK_SZ; ++y) {
		for (unsigned x = bx*GRASS_BLOCK_SZ; x < (bx + 1)*GRASS_BLOCK_SZ; ++x) {
			if (!(flags[y*WT_TEX + x] & 2u)) continue;
			unsigned const ix = y*WT_ZV + x;
			float const mh00 = zvals[ix], mh01 = zvals[ix+1], mh10 = zvals[ix+WT_ZV], mh11 = zvals[ix+WT_ZV+1];
			float const mhmin = min_std(min_std(mh00, mh01), min_std(mh10, mh11)), mhmax = max_std(max_std(mh00, mh01), max_std(mh10, mh11));
			if (gb.ix == 0) {
				gb.ix = ((((uint32_t)x1 + x) + 1567u*((uint32_t)y1 + y)) % c.num_rnd_grass_blocks) + 1; // int + unsigned: unsigned arithmetic
				gb.zmin = mhmin; gb.zmax = mhmax;
			}
			else {gb.zmin = min_std(gb.zmin, mhmin); gb.zmax = max_std(gb.zmax, mhmax);}
		}
	}
	return gb;
}

} // namespace terra
