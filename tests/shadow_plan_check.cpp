// tests/shadow_plan_check.cpp -- TEST INFRASTRUCTURE (built and run by tests/test_shadow_plan.py, no GPU): the host side of the mesh-shadow LDS kernels.
//   * shadow_path_zones (3dworld_amd/csrc/terra_driver.hpp): the closed form of where a sweep's Bresenham walk (src/visibility.cpp:436-486) leaves its first column / row and
//     reaches its last, against the walk itself -- for the abstract (longest, shortest) pairs AND for the real sweeps of tiles under a set of light directions;
//   * shadow_lane_order: every sweep exactly once, idle lanes marked, waves sorted by length and dealt so that the waves sharing a SIMD (i, i + 4, i + 8) carry equal work.
#include "../3dworld_amd/csrc/terra_driver.hpp"
#include <cstdio>
#include <cmath>
#include <set>
using namespace terra;

static shadow_consts_t consts(float lx, float ly, float lz, int zv) {
	shadow_consts_t c; memset(&c, 0, sizeof(c));
	float const mag = sqrtf(lx*lx + ly*ly + lz*lz);
	c.X_SCENE_SIZE = 4.0f; c.Y_SCENE_SIZE = 4.0f; c.DX_VAL = 8.0f/128.0f; c.DY_VAL = 8.0f/128.0f; c.DX_VAL_INV = 1.0f/c.DX_VAL; c.DY_VAL_INV = 1.0f/c.DY_VAL;
	c.zmin = -10.0f; c.zmax = 10.0f; c.dirx = -lx/mag; c.diry = -ly/mag; c.dirz = -lz/mag;
	c.dist = (float)(2.0*256.0/(double)sqrtf(c.dirx*c.dirx + c.diry*c.diry));
	c.xsize = zv; c.ysize = zv; c.mask_fill = 0;
	return c;
}

int main() {
	int bad = 0;
	if (!shadow_path_zones_hold()) {printf("shadow_path_zones does not hold for some (longest, shortest)\n"); ++bad;}
	float const lights[][3] = {{0.6f, 0.5f, 0.4f}, {-0.8f, 0.3f, 0.25f}, {0.2f, -0.9f, 0.15f}, {-0.5f, -0.5f, 0.8f}, {1.0f, 0.0f, 0.3f}, {0.0f, -1.0f, 0.2f}, {0.05f, 0.9f, 0.02f}, {0.7f, 0.4f, 0.3f}, {-0.01f, 0.3f, 0.9f}};
	for (auto const &L : lights) {
		for (int zv : {130, 34}) {
			shadow_consts_t const c = consts(L[0], L[1], L[2], zv);
			uint32_t const npaths = 4u*(uint32_t)zv;
			std::vector<int> len(npaths, 0);
			for (uint32_t p = 0; p < npaths; ++p) { // the zones of the real sweeps against their walks
				shadow_path_t w;
				if (!shadow_path_setup(c, p, w)) continue;
				len[p] = w.longest + 1;
				int fe, lb; shadow_path_zones(w.longest, w.shortest, fe, lb);
				int x = w.xa, y = w.ya, numerator = w.longest >> 1;
				for (int i = 0; i <= w.longest; ++i) {
					bool const on_first = (x == w.xa || y == w.ya), on_last = (x == w.xb || y == w.yb);
					if (on_first != (i < fe) || on_last != (i >= lb)) {if (bad < 10) printf("light %g %g %g zv %d sweep %u step %d: zones [0,%d) [%d,%d] but first %d last %d\n", L[0], L[1], L[2], zv, p, i, fe, lb, w.longest, on_first, on_last); ++bad;}
					if (i > 0 && i < w.longest && !on_first && !on_last && !((unsigned)x < (unsigned)c.xsize && (unsigned)y < (unsigned)c.ysize)) {if (bad < 10) printf("sweep %u step %d: a middle cell outside the tile (%d, %d)\n", p, i, x, y); ++bad;}
					numerator += w.shortest;
					if (numerator >= w.longest) {numerator -= w.longest; x += w.dx1; y += w.dy1;} else {x += w.dx2; y += w.dy2;}
				}
			}
			uint32_t const lanes = (npaths + 63u)/64u*64u + 64u; // one spare wave, as the kernels have (520 sweeps on 576 lanes)
			std::vector<uint16_t> order(lanes, 0);
			if (!shadow_lane_order(c, npaths, lanes, order.data())) {printf("shadow_lane_order refused %u sweeps on %u lanes\n", npaths, lanes); ++bad; continue;}
			std::set<uint32_t> seen; uint32_t idle = 0;
			for (uint32_t l = 0; l < lanes; ++l) {if (order[l] == 0xFFFF) {++idle;} else {if (order[l] >= npaths || !seen.insert(order[l]).second) {printf("lane %u: sweep %u twice or out of range\n", l, order[l]); ++bad;}}}
			if (seen.size() != npaths || idle != lanes - npaths) {printf("lane order is not a permutation: %zu sweeps, %u idle of %u lanes\n", seen.size(), idle, lanes); ++bad;}
			// a wave costs its longest sweep; the waves of a SIMD (i, i + 4, ...) should carry about equal sums, and the longest sweeps sit in wave 0
			uint32_t const nw = lanes/64; std::vector<int> wave_len(nw, 0); int simd[4] = {0, 0, 0, 0}, total = 0, longest_any = 0;
			for (uint32_t wv = 0; wv < nw; ++wv) {for (uint32_t l = 0; l < 64; ++l) {uint16_t const p = order[64*wv + l]; if (p != 0xFFFF) wave_len[wv] = std::max(wave_len[wv], len[p]);} simd[wv % 4] += wave_len[wv]; total += wave_len[wv]; longest_any = std::max(longest_any, wave_len[wv]);}
			if (wave_len[0] != longest_any) {printf("the longest sweep is not in wave 0\n"); ++bad;}
			int const hi = std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3]));
			if (nw >= 8 && hi*4 > total + total/4 + 4*longest_any/2) {printf("light %g %g %g zv %d: SIMD sums %d %d %d %d are not balanced\n", L[0], L[1], L[2], zv, simd[0], simd[1], simd[2], simd[3]); ++bad;}
		}
	}
	uint16_t tmp[64];
	if (shadow_lane_order(consts(0.6f, 0.5f, 0.4f, 130), 520, 64, tmp)) {printf("shadow_lane_order accepted more sweeps than lanes\n"); ++bad;}
	printf("%s (%d problems)\n", bad ? "FAILED" : "ok", bad);
	return bad ? 1 : 0;
}
