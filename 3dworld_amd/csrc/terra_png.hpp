// terra_png.hpp -- the on-disk format either side of the heightmap path: 8- / 16-bit grayscale PNG, host code.
//
// What the reference does with libpng (src/image_io.cpp:493-605, heightmap_t / terrain_hmap_manager_t::write_png src/heightmap.cpp:366-378):
//   write   rows in memory order (row 0 first); 16-bit pixels are {fraction, integer} byte pairs = little-endian uint16 in memory, big-endian in the file
//   read    file row i lands in memory row height-1-i (texture_t::load_png flips), 16-bit samples swapped back to little-endian;
//           allow_two_byte_grayscale keeps 16-bit gray as two bytes per pixel, everything else is reduced to 8 bits per sample
// Written from the PNG specification (chunks + zlib stream + the five scanline filters), zlib for deflate / inflate / crc32.  Only what heightmaps use:
// colour type 0 (grayscale), bit depth 8 or 16, no interlace; other files are refused with an error instead of being converted.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <stdexcept>
#include <zlib.h>

namespace terra {

inline void png_put32(std::vector<uint8_t> &v, uint32_t x) {v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);}
inline uint32_t png_get32(uint8_t const *p) {return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];}
inline void png_chunk(std::vector<uint8_t> &out, char const type[4], uint8_t const *data, size_t n) {
	png_put32(out, (uint32_t)n);
	size_t const start = out.size();
	out.insert(out.end(), type, type + 4);
	if (n) out.insert(out.end(), data, data + n);
	png_put32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4)));
}

// pixels: height rows of width samples, 1 byte (ncolors 1) or 2 bytes {lo, hi} (ncolors 2) each; written top row first, like texture_t::write_to_png
inline void png_write_gray(std::string const &path, uint8_t const *pixels, uint32_t width, uint32_t height, int ncolors) {
	if (!pixels || width == 0 || height == 0 || (ncolors != 1 && ncolors != 2)) throw std::invalid_argument("png_write_gray: bad image");
	size_t const row = (size_t)width*ncolors;
	std::vector<uint8_t> raw((row + 1)*height);
	for (uint32_t y = 0; y < height; ++y) {
		uint8_t *dst = raw.data() + (row + 1)*y;
		uint8_t const *src = pixels + row*y;
		*dst++ = 0; // filter type 0 (None): any valid filtering decodes to the same samples
		if (ncolors == 1) {memcpy(dst, src, row);}
		else {for (uint32_t x = 0; x < width; ++x) {dst[2*x] = src[2*x + 1]; dst[2*x + 1] = src[2*x];}} // big-endian samples in the file (png_set_swap)
	}
	uLongf zlen = compressBound((uLong)raw.size());
	std::vector<uint8_t> z(zlen);
	if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("png_write_gray: deflate failed");
	std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	std::vector<uint8_t> ihdr;
	png_put32(ihdr, width); png_put32(ihdr, height);
	ihdr.push_back((uint8_t)(ncolors == 2 ? 16 : 8)); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); // bit depth, colour type 0, deflate, adaptive filtering, no interlace
	png_chunk(out, "IHDR", ihdr.data(), ihdr.size());
	png_chunk(out, "IDAT", z.data(), zlen);
	png_chunk(out, "IEND", nullptr, 0);
	FILE *fp = fopen(path.c_str(), "wb");
	if (!fp) throw std::runtime_error("png_write_gray: cannot open " + path + " for write");
	size_t const w = fwrite(out.data(), 1, out.size(), fp);
	fclose(fp);
	if (w != out.size()) throw std::runtime_error("png_write_gray: short write to " + path);
}

inline int png_paeth(int a, int b, int c) {int const p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : ((pb <= pc) ? b : c);}

// returns pixels in the reference's memory layout: bottom file row first (load_png's flip), 16-bit samples as {lo, hi}.  two_byte = allow_two_byte_grayscale:
// a 16-bit file read without it is reduced to its high bytes (png_set_strip_16)
inline std::vector<uint8_t> png_read_gray(std::string const &path, uint32_t &width, uint32_t &height, int &ncolors, bool two_byte = true) {
	FILE *fp = fopen(path.c_str(), "rb");
	if (!fp) throw std::runtime_error("png_read_gray: cannot open " + path);
	std::vector<uint8_t> f;
	uint8_t buf[65536];
	for (size_t n; (n = fread(buf, 1, sizeof(buf), fp)) > 0;) {f.insert(f.end(), buf, buf + n);}
	fclose(fp);
	static uint8_t const sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	if (f.size() < 8 || memcmp(f.data(), sig, 8) != 0) throw std::runtime_error("png_read_gray: " + path + " is not a PNG file");
	std::vector<uint8_t> idat;
	int bit_depth = 0; bool have_ihdr = false, have_iend = false;
	for (size_t pos = 8; pos + 12 <= f.size();) {
		uint32_t const len = png_get32(&f[pos]);
		if (pos + 12 + (size_t)len > f.size()) throw std::runtime_error("png_read_gray: truncated chunk in " + path);
		uint8_t const *type = &f[pos + 4], *data = &f[pos + 8];
		if (png_get32(&f[pos + 8 + len]) != (uint32_t)crc32(0L, type, (uInt)(len + 4))) throw std::runtime_error("png_read_gray: CRC mismatch in " + path);
		if (memcmp(type, "IHDR", 4) == 0) {
			if (len != 13) throw std::runtime_error("png_read_gray: bad IHDR");
			width = png_get32(data); height = png_get32(data + 4); bit_depth = data[8];
			if (data[9] != 0 || (bit_depth != 8 && bit_depth != 16) || data[10] != 0 || data[11] != 0 || data[12] != 0 || width == 0 || height == 0) {
				throw std::runtime_error("png_read_gray: only non-interlaced 8- or 16-bit grayscale PNGs are supported (" + path + ")");
			}
			// libpng refuses dimensions above its user limits (PNG_USER_WIDTH_MAX / HEIGHT_MAX = 1 000 000) and anything above 2^31 - 1; a heightmap is at most
			// 65536 texels per side (max_tex_ix(), src/heightmap.h:42).  With that bound (row + 1)*height and width*height*2 stay far below 2^63.
			if (width > 65536u || height > 65536u) throw std::runtime_error("png_read_gray: image dimensions above 65536 (" + path + ")");
			have_ihdr = true;
		}
		else if (memcmp(type, "IDAT", 4) == 0) {idat.insert(idat.end(), data, data + len);}
		else if (memcmp(type, "IEND", 4) == 0) {have_iend = true; break;}
		pos += 12 + (size_t)len;
	}
	if (!have_ihdr || !have_iend || idat.empty()) throw std::runtime_error("png_read_gray: incomplete PNG " + path);
	int const bpp = bit_depth/8;
	size_t const row = (size_t)width*bpp;
	std::vector<uint8_t> raw((row + 1)*height);
	uLongf rlen = (uLongf)raw.size();
	if (uncompress(raw.data(), &rlen, idat.data(), (uLong)idat.size()) != Z_OK || rlen != raw.size()) throw std::runtime_error("png_read_gray: inflate failed for " + path);
	std::vector<uint8_t> prev(row, 0), cur(row);
	ncolors = (bit_depth == 16 && two_byte) ? 2 : 1;
	std::vector<uint8_t> out((size_t)width*height*ncolors);
	for (uint32_t y = 0; y < height; ++y) {
		uint8_t const *src = raw.data() + (row + 1)*y;
		int const ft = src[0];
		++src;
		for (size_t i = 0; i < row; ++i) {
			int const a = (i >= (size_t)bpp) ? cur[i - bpp] : 0, b = prev[i], c = (i >= (size_t)bpp) ? prev[i - bpp] : 0;
			int v = src[i];
			switch (ft) {
			case 0: break;
			case 1: v += a; break;
			case 2: v += b; break;
			case 3: v += (a + b)/2; break;
			case 4: v += png_paeth(a, b, c); break;
			default: throw std::runtime_error("png_read_gray: bad filter type in " + path);
			}
			cur[i] = (uint8_t)v;
		}
		uint8_t *dst = out.data() + (size_t)(height - 1 - y)*width*ncolors; // rows[i] = data + (height - i - 1)*scanline_size
		if (bit_depth == 8) {memcpy(dst, cur.data(), row);}
		else if (ncolors == 2) {for (uint32_t x = 0; x < width; ++x) {dst[2*x] = cur[2*x + 1]; dst[2*x + 1] = cur[2*x];}} // big endian -> {lo, hi}
		else {for (uint32_t x = 0; x < width; ++x) {dst[x] = cur[2*x];}} // png_set_strip_16: keep the most significant byte
		prev.swap(cur);
	}
	return out;
}

} // namespace terra
