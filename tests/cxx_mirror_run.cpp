// tests/cxx_mirror_run.cpp -- include/terra_cxx.hpp EXECUTED: the engine-side C++ mirror of the reference's call surface, driven exactly as an engine caller would
// (heightmap_t::proc_gen's pattern, src/heightmap.cpp:135-143,169; tile_t::create_zvals; eval_mesh_sin_terms, eval_mesh_sin_terms_scaled, get_exact_zval), linked against libterra_emul.so on the CPU box and
// against libterra_hip.so on the GPU box (tests/test_engine_in_the_loop.py compares the floats it writes with the oracle).
#include "terra_cxx.hpp"
#include <cstring>
#include <cstdio>
#include <vector>
using namespace terra_cxx;

int main(int argc, char **argv) {
	if (argc < 2) return 2;
	terra_config c; std::memset(&c, 0, sizeof(c)); // the synthetic scene of BASELINE.md section 3 (scene_config/config.txt:56-97), 8 octaves
	c.mesh_x = c.mesh_y = 128; c.scene_x = c.scene_y = c.scene_z = 4.0f; c.mesh_height = 0.7f; c.mesh_scale = 1.0f;
	c.mesh_seed = 1; c.mesh_freq_filter = 1; c.mesh_gen_mode = TERRA_MGEN_SINE; c.mesh_gen_shape = 0; c.glaciate = 1;
	c.hmap[0] = 1000.0f; c.hmap[4] = 1000.0f; c.hmap[9] = 5.0f; c.hmap[10] = 0.001f; c.hmap[11] = -4.0f;
	c.erode_amount = 1.0f; c.start_mag = 0.02f; c.start_freq = 240.0f; c.mag_mult = 2.0f; c.freq_mult = 0.5f;
	check(terra_init_scene(default_ctx(), &c), "terra_init_scene");
	terra_state st; check(terra_get_state(default_ctx(), &st), "terra_get_state");
	unsigned const nx = 70, ny = 50;
	std::vector<float> out;
	{ // heightmap_t::proc_gen's call pattern on the mirror class
		mesh_xy_grid_cache_t height_gen;
		bool const ready = height_gen.build_arrays(-35.0f, -25.0f, st.DX_VAL, st.DY_VAL, nx, ny);
		if (!ready) return 3;
		height_gen.enable_glaciate();
		std::vector<float> vals(nx*ny);
		for (unsigned y = 0; y < ny; ++y) {for (unsigned x = 0; x < nx; ++x) {vals[y*nx + x] = height_gen.eval_index(x, y);}}
		out.insert(out.end(), vals.begin(), vals.end());
		float min_zval = vals[0];
		for (float v : vals) {min_zval = (v < min_zval) ? v : min_zval;}
		apply_erosion(vals.data(), (int)nx, (int)ny, min_zval, 300); // run_erosion: min_zval = min(vals)
		out.insert(out.end(), vals.begin(), vals.end());
		// the no_wait protocol: launched -> 0, same arguments again -> collected
		mesh_xy_grid_cache_t g2;
		bool const r0 = g2.build_arrays(3.0f, 4.0f, st.DX_VAL, st.DY_VAL, 33, 17, false, false, true);
		bool const r1 = g2.build_arrays(3.0f, 4.0f, st.DX_VAL, st.DY_VAL, 33, 17, false, false, false);
		if (r0 || !r1) return 4;
		g2.clear_context(); g2.free_cshader();
	}
	float const pts[3][2] = {{0.5f, 0.5f}, {-10.25f, 3.0f}, {100.0f, -77.5f}};
	for (auto const &p : pts) {out.push_back(eval_mesh_sin_terms(p[0], p[1]));}
	{
		int const txy[2] = {2, -1};
		std::vector<float> z(130*130); terra_tile_stats ts;
		std::vector<unsigned char> nm(129*129*4), nm2(129*129*4); float mnz = 0.0f, mnz2 = 0.0f; terra_tile_stats ts2;
		tiles_create_zvals(txy, 1, 80, z.data(), &ts, nm.data(), &mnz);
		out.insert(out.end(), z.begin(), z.end());
		tiles_upload_normal_texture(txy, 1, z.data(), &ts2, nm2.data(), &mnz2); // the post pass alone over those zvals: the same stats, normals and min_normal_z
		if (std::memcmp(&ts, &ts2, sizeof(ts)) != 0 || nm != nm2 || std::memcmp(&mnz, &mnz2, sizeof(float)) != 0) return 6;
	}
	{ // the all-modes point queries: one point and a batch, in index space (scaled) and in world space (exact, with a scroll offset)
		float const q[4][2] = {{0.5f, 0.5f}, {-10.25f, 3.0f}, {100.0f, -77.5f}, {3.75f, 12.5f}};
		float zs[4], ze[4];
		eval_mesh_sin_terms_scaled(&q[0][0], 4, 0.5f, zs);
		get_exact_zval(&q[0][0], 4, ze, false, 3, -2);
		out.insert(out.end(), zs, zs + 4); out.insert(out.end(), ze, ze + 4);
		out.push_back(eval_mesh_sin_terms_scaled(q[1][0], q[1][1], 0.5f)); out.push_back(get_exact_zval(q[2][0], q[2][1], false, 3, -2));
	}
	FILE *f = std::fopen(argv[1], "wb");
	if (!f) return 5;
	std::fwrite(out.data(), sizeof(float), out.size(), f);
	std::fclose(f);
	std::printf("cxx mirror ok: %zu floats\n", out.size());
	return 0;
}
