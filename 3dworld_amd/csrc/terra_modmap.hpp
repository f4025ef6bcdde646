// Height-edit formats either side of the heightmap texture: tex_mod_map_manager_t's brushes, mod map and .mod file (src/heightmap.h:38-108,
// src/heightmap.cpp:27-58,216-308) and the per-pixel update heightmap_t::modify_heightmap_value (src/heightmap.cpp:99-115).
#pragma once
#include "terra_common.hpp"
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>
#include <stdio.h>

namespace terra {

enum {BSHAPE_CONST_SQ = 0, BSHAPE_CNST_CIR, BSHAPE_LINEAR, BSHAPE_QUADRATIC, BSHAPE_COSINE, BSHAPE_SINE, BSHAPE_FLAT_SQ, BSHAPE_FLAT_CIR, NUM_BSHAPES}; // src/heightmap.h:11
struct hmap_brush_pod_t {int32_t x, y; uint32_t radius; int32_t delta; int16_t shape;}; // tex_mod_map_manager_t::hmap_brush_t (src/heightmap.h:71-76), 20 bytes in the file
struct hmap_mod_pod_t {uint16_t x, y; int32_t delta;};                                  // mod_elem_t (src/heightmap.h:59-64), 8 bytes
static_assert(sizeof(hmap_brush_pod_t) == 20 && sizeof(hmap_mod_pod_t) == 8, "mod file record layout");

// heightmap_t::modify_heightmap_value on an image other threads edit too: a compare-and-swap on the containing 32-bit word.  Within one brush every
// delta has the sign of the brush's delta (the weights are >= 0) and saturating adds of same-signed values commute, flatten brushes store one value,
// and the mod map has one entry per texel -- so the result does not depend on the order the threads arrive in.
// The word is the naturally aligned one around the texel's ABSOLUTE address (a caller may hand in any 2-byte aligned sub-buffer), so the
// compare-and-swap is always aligned; the bytes of the word that belong to other texels (or, for the first / last word of an image that does
// not start / end on a 4-byte boundary, to whatever the caller keeps next to it) are written back unchanged.
TERRA_HD void modify_pixel(uint8_t *pix, int ncolors, size_t ix, int val, bool is_delta) {
	size_t const byte = (ncolors == 2) ? (ix << 1) : ix;
	int const vmax = (ncolors == 2) ? 65535 : 255;
#if defined(__HIP_DEVICE_COMPILE__)
	uintptr_t const a = (uintptr_t)(pix + byte);
	unsigned const shift = (unsigned)(a & 3u)*8u, mask = (ncolors == 2) ? 0xFFFFu : 0xFFu;
	uint32_t *w = (uint32_t *)(a & ~(uintptr_t)3);
	uint32_t old = *w;
	for (;;) {
		int v = val;
		if (is_delta) {v += (int)((old >> shift) & mask);}
		uint32_t const nv = (uint32_t)imax(0, imin(vmax, v));
		uint32_t const want = (old & ~(mask << shift)) | (nv << shift);
		uint32_t const seen = atomicCAS(w, old, want);
		if (seen == old) break;
		old = seen;
	}
#else
	// host emulation (tests): one thread, the texel's own bytes only
	int v = val;
	if (is_delta) {v += (ncolors == 2) ? ((int)pix[byte] | ((int)pix[byte + 1] << 8)) : (int)pix[byte];}
	uint32_t const nv = (uint32_t)imax(0, imin(vmax, v));
	pix[byte] = (uint8_t)(nv & 0xFFu);
	if (ncolors == 2) {pix[byte + 1] = (uint8_t)(nv >> 8);}
#endif
}

// tex_mod_map_t::add for a list: one entry per texel in map order (x, then y), deltas summed (src/heightmap.h:44-49,66-69)
inline std::vector<hmap_mod_pod_t> combine_mods(hmap_mod_pod_t const *mods, size_t n) {
	std::vector<hmap_mod_pod_t> m(mods, mods + n), out;
	std::stable_sort(m.begin(), m.end(), [](hmap_mod_pod_t const &a, hmap_mod_pod_t const &b) {return (a.x == b.x) ? (a.y < b.y) : (a.x < b.x);});
	for (hmap_mod_pod_t const &e : m) {
		if (!out.empty() && out.back().x == e.x && out.back().y == e.y) {out.back().delta += e.delta;} else {out.push_back(e);}
	}
	return out;
}

constexpr uint32_t MOD_HEADER_SIG = 0xdeadbeefu, MOD_TRAILER_SIG = 0xbeefdeadu; // src/heightmap.cpp:240-241
// tex_mod_map_manager_t::write_mod (src/heightmap.cpp:283-308): header, count, combined mods in map order, brush count, brushes, trailer
inline void write_mod_file(char const *fn, hmap_mod_pod_t const *mods, size_t n, hmap_brush_pod_t const *brushes, size_t nb) {
	std::vector<hmap_mod_pod_t> const m = combine_mods(mods, n);
	FILE *fp = fopen(fn, "wb");
	if (!fp) throw std::runtime_error(std::string("cannot open terrain height mod map for write: ") + fn);
	uint32_t const hs = MOD_HEADER_SIG, ts = MOD_TRAILER_SIG, cnt = (uint32_t)m.size(), bcnt = (uint32_t)nb;
	std::vector<hmap_brush_pod_t> b(nb);
	for (size_t i = 0; i < nb; ++i) {memset(&b[i], 0, sizeof(b[i])); b[i].x = brushes[i].x; b[i].y = brushes[i].y; b[i].radius = brushes[i].radius; b[i].delta = brushes[i].delta; b[i].shape = brushes[i].shape;} // padding written as zeros
	bool ok = fwrite(&hs, 4, 1, fp) == 1 && fwrite(&cnt, 4, 1, fp) == 1 && fwrite(m.data(), sizeof(hmap_mod_pod_t), m.size(), fp) == m.size();
	ok = ok && fwrite(&bcnt, 4, 1, fp) == 1 && fwrite(b.data(), sizeof(hmap_brush_pod_t), nb, fp) == nb && fwrite(&ts, 4, 1, fp) == 1;
	ok = (fclose(fp) == 0) && ok;
	if (!ok) throw std::runtime_error(std::string("error writing terrain height mod map ") + fn);
}
// tex_mod_map_manager_t::read_mod (src/heightmap.cpp:243-281); where the reference asserts on a short file this throws
inline void read_mod_file(char const *fn, std::vector<hmap_mod_pod_t> &mods, std::vector<hmap_brush_pod_t> &brushes) {
	FILE *fp = fopen(fn, "rb");
	if (!fp) throw std::runtime_error(std::string("cannot open terrain height mod map for read: ") + fn);
	uint32_t v = 0, sz = 0, bsz = 0;
	std::vector<hmap_mod_pod_t> raw;
	auto fail = [&](char const *what) {fclose(fp); throw std::runtime_error(std::string(what) + " in terrain height mod map " + fn);};
	if (fread(&v, 4, 1, fp) != 1 || v != MOD_HEADER_SIG) fail("incorrect header");
	if (fread(&sz, 4, 1, fp) != 1) fail("truncated");
	long const pos = ftell(fp); fseek(fp, 0, SEEK_END); long const end = ftell(fp); fseek(fp, pos, SEEK_SET);
	if ((uint64_t)sz*sizeof(hmap_mod_pod_t) > (uint64_t)(end - pos)) fail("truncated");
	raw.resize(sz);
	if (sz && fread(raw.data(), sizeof(hmap_mod_pod_t), sz, fp) != sz) fail("truncated");
	if (fread(&bsz, 4, 1, fp) != 1 || (uint64_t)bsz*sizeof(hmap_brush_pod_t) > (uint64_t)(end - ftell(fp))) fail("truncated");
	brushes.resize(bsz);
	if (bsz && fread(brushes.data(), sizeof(hmap_brush_pod_t), bsz, fp) != bsz) fail("truncated");
	if (fread(&v, 4, 1, fp) != 1 || v != MOD_TRAILER_SIG) fail("incorrect trailer");
	fclose(fp);
	mods = combine_mods(raw.data(), raw.size());
}

} // namespace terra
