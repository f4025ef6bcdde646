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
// ---- one texel.  The reference's per-texel body (src/tiled_mesh.cpp:1146-1200) is a long straight run of statements; here it is cut along what a texel can BE:
//   * a texel whose four corner heights fall into ONE layer that is neither grass nor snow (sea floor, beaches, bare dirt / rock bands: most of an ocean tile) has a
//     one-hot weight vector -- nothing to blend, no slope, no double arithmetic; only dirt under vegetation still asks the biome field for its sand share;
//   * everything else goes through blend_texel(), which evaluates the reference's statements in their order with their float / double promotions.
// Both give the bytes of the reference's statements (the one-hot case is those statements with t = 0, weight_scale = 1 folded: 0 + 1.0*(1.0 - 0.0) = 1, the other sums 0).
struct corner_heights_t {float z00, z01, z10, z11;}; // the texel's cell: (x, y), (x + 1, y), (x, y + 1), (x + 1, y + 1)
struct biome_corners_t {float v[12];};               // update_terrain_params: [yp][xp][{vegetation, grass, dirt}] at the tile's corners
TERRA_HD float biome_at(biome_corners_t const &b, int which, float fx, float fy) { // BILINEAR_INTERP (src/tiled_mesh.cpp:189)
	float const p00 = b.v[which], p01 = b.v[3 + which], p10 = b.v[6 + which], p11 = b.v[9 + which];
	return fy*(fx*p11 + (1.0f - fx)*p10) + (1.0f - fy)*(fx*p01 + (1.0f - fx)*p00);
}
TERRA_HD float add_dbl(float w, double v) {return (float)((double)w + v);} // a float accumulator taking a double expression
TERRA_HD uint32_t weight_byte(float w) {return ((double)w <= 0.01) ? 0u : (((double)w >= 0.99) ? 255u : (uint32_t)(unsigned char)(255.0*(double)w));}
TERRA_HD uint32_t pack_weights(float const (&w)[5]) {return weight_byte(w[0]) | (weight_byte(w[1]) << 8) | (weight_byte(w[2]) << 16) | (weight_byte(w[3]) << 24);} // (snow is the remainder)
TERRA_HD float min4_std(corner_heights_t const &h) {return min_std(min_std(h.z00, h.z01), min_std(h.z10, h.z11));}
TERRA_HD float max4_std(corner_heights_t const &h) {return max_std(max_std(h.z00, h.z01), max_std(h.z10, h.z11));}

// sand taken out of dirt (where the biome's dirt share is below one) and out of grass (biome's grass share, none under water); sets flag bit 1 when the texel feeds a grass block
TERRA_HD void biome_to_sand(landscape_consts_t const &c, biome_corners_t const &bio, float (&w)[5], bool grass, float lowest, unsigned x, unsigned y, unsigned &flags) {
	float const fx = (float)x*(1.0f/128.0f), fy = (float)y*(1.0f/128.0f);
	if (c.vegetation > 0.0f) {
		float const keep = biome_at(bio, 2, fx, fy);
		if (keep < 1.0f) {w[LT_SAND] = add_dbl(w[LT_SAND], (1.0 - (double)keep)*(double)w[LT_DIRT]); w[LT_DIRT] *= keep;}
	}
	if (grass) {
		float const share = (lowest < c.water_level) ? 0.0f : biome_at(bio, 1, fx, fy);
		if (share < 1.0f) {
			float const keep = clip01(2.5f*(share - 0.5f) + 0.5f);
			w[LT_SAND] = add_dbl(w[LT_SAND], (1.0 - (double)keep)*(double)w[LT_GROUND]); w[LT_GROUND] *= keep;
		}
		if (share > 0.0f && c.gen_grass_map && x < WT_SIZE && y < WT_SIZE) {flags |= 2u;}
	}
}
// the general texel: layers la (lower) / lb (upper) blended by t, slopes turning grass into dirt / rock and snow into rock
TERRA_HD uint32_t blend_texel(landscape_consts_t const &c, corner_heights_t const &h, biome_corners_t const &bio, float jitter, float lowest, int la, int lb, float t, unsigned x, unsigned y, unsigned &flags) {
	float w[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
	bool const grass = (la == LT_GROUND || lb == LT_GROUND), snow = (lb == LT_SNOW);
	flags = grass ? 1u : 0u;
	float cover = 1.0f; // what the slope leaves of the layers
	if (grass || snow) {
		float const lo = sthresh_v(snow, 0), hi = sthresh_v(snow, 1);
		float const gx = c.DY_VAL*(h.z00 - h.z01), gy = c.DX_VAL*(h.z00 - h.z10), gz = c.dxdy; // get_norm_not_normalized (src/tiled_mesh.h:281)
		float up = c.vnz_scale*gz/sqrtf(gx*gx + gy*gy + gz*gz);
		if (grass && up > hi) {up = clip01(1.0f + 20.0f*jitter);}
		if (up < hi) {
			if (grass) {
				float rock = (la == LT_GROUND || lb == LT_ROCK) ? t : 0.0f;
				float const steep = (float)(1.0 - (double)clip01((up - 0.5f*lo)*c.steep_mult_rock));
				rock  = (float)((double)rock*(1.0 - (double)steep) + (double)steep);
				cover = clip01((up - lo)*c.steep_mult_grass);
				w[LT_ROCK] = add_dbl(w[LT_ROCK], (1.0 - (double)cover)*(double)rock);
				w[LT_DIRT] = add_dbl(w[LT_DIRT], (1.0 - (double)cover)*(1.0 - (double)rock));
			}
			else {
				cover = clip01(2.0f*(up - lo)*c.steep_mult_snow);
				w[LT_ROCK] = add_dbl(w[LT_ROCK], 1.0 - (double)cover);
			}
		}
	}
	w[lb] += cover*t;
	w[la] = add_dbl(w[la], (double)cover*(1.0 - (double)t));
	biome_to_sand(c, bio, w, grass, lowest, x, y, flags);
	return pack_weights(w);
}
// one texel (x, y) of the 129 x 129 weights texture -> RGBA packed little-endian.  flags: bit 0 = grass (has_any_grass), bit 1 = contributes to its grass block
TERRA_HD uint32_t weights_texel_v(landscape_consts_t const &c, corner_heights_t const &h, biome_corners_t const &bio, float rand_val, unsigned x, unsigned y, unsigned &flags) {
	float const lowest = min4_std(h), highest = max4_std(h), jitter = c.noise_scale*rand_val;
	float const rel_lo = c.relh_adj_tex + (lowest - c.zmin)*c.dz_inv + jitter, rel_hi = c.relh_adj_tex + (highest - c.zmin)*c.dz_inv + jitter;
	int lo_a, lo_b, hi_a, hi_b;
	get_tids(c, rel_lo, lo_a, lo_b, nullptr);
	get_tids(c, rel_hi, hi_a, hi_b, nullptr);
	if (lo_a == hi_b) { // one layer from the lowest corner's lower candidate to the highest corner's upper one
		int const layer = lo_a;
		if (layer != LT_GROUND && layer != LT_SNOW) { // one-hot: bytes directly
			flags = 0u;
			if (layer == LT_DIRT && c.vegetation > 0.0f) {
				float const keep = biome_at(bio, 2, (float)x*(1.0f/128.0f), (float)y*(1.0f/128.0f));
				if (keep < 1.0f) {float const sand = add_dbl(0.0f, (1.0 - (double)keep)*1.0), dirt = 1.0f*keep; return weight_byte(sand) | (weight_byte(dirt) << 8);}
			}
			return 255u << (8*layer);
		}
		// (a shortcut for grass on a SURELY gentle slope -- the square root and the division replaced by a comparison of squares with a margin -- was measured and lost:
		// 445 -> 490 us for the 64 x 64 batch.  A wave executes every path one of its 64 texels takes; land rows mix gentle and steep cells, so the shortcut was added work.)
		return blend_texel(c, h, bio, jitter, lowest, layer, layer, 0.0f, x, y, flags);
	}
	// several layers under the cell: the corner (x, y) decides, without the jitter, and may sit in a blend zone
	float t = 0.0f;
	int la, lb;
	get_tids(c, c.relh_adj_tex + (h.z00 - c.zmin)*c.dz_inv, la, lb, &t);
	return blend_texel(c, h, bio, jitter, lowest, la, lb, t, x, y, flags);
}
TERRA_HD uint32_t weights_texel(landscape_consts_t const &c, float const *zvals /*130x130*/, float const *prm /*12*/, float rand_val, unsigned x, unsigned y, unsigned &flags) {
	unsigned const ix = y*WT_ZV + x;
	biome_corners_t bio;
	for (int k = 0; k < 12; ++k) {bio.v[k] = prm[k];}
	return weights_texel_v(c, corner_heights_t{zvals[ix], zvals[ix+1], zvals[ix+WT_ZV], zvals[ix+WT_ZV+1]}, bio, rand_val, x, y, flags);
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
