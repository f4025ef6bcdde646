/* tools/bench_native_multi.c -- all GPUs of a node from ONE C process through include/terra.h (terra_multi_*): what a 3DWorld build would do, no Python,
 * no torch.distributed.  One context per device (or --same-device: G contexts on device 0, for a box with one GPU), one host thread each inside a call.
 *   (a) headline: one independent 16384^2 heightmap region per GPU (noise + glaciate + fused min, 1000-droplet erosion), P in flight per GPU  -> weak scaling
 *   (b) BASELINE config 4: the 64 x 64 tile batch block-partitioned over the GPUs (0 and 1000 droplets per tile)                               -> strong scaling
 *   (c) ONE 16384^2 heightmap as row strips, min of the whole map folded on the host                                                          -> strong scaling
 *   (d) BASELINE config 5: ONE 512^3 voxel field as y slabs                                                                                   -> strong scaling
 *   (e) mesh shadows of the 64 x 64 terrain: column strips, rows pipelined, border edges by hipMemcpyPeerAsync
 *   gcc -O2 -std=c99 -Iinclude tools/bench_native_multi.c -L3dworld_amd -lterra_hip -lpthread -Wl,-rpath,$PWD/3dworld_amd -o tools/_bin/bench_native_multi
 *   tools/_bin/bench_native_multi [gpus=all] [steps=16] [size=16384] [--same-device]                                                                       */
#define _POSIX_C_SOURCE 199309L
#include "terra.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXG 64
#define PIPES 4
static double now(void) {struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9*(double)t.tv_nsec;}
#define CK(x) do {if ((x) != 0) {fprintf(stderr, "%s failed: %s\n", #x, terra_last_error()); exit(1);}} while (0)

/* (a): every GPU runs PIPES pipelines (own context each, like bench.py); the contexts of all GPUs form one terra_multi so one foreach starts them all */
typedef struct {float *z; int n, steps, droplets, gpu_slot; terra_state st;} region_t;
static region_t regions[MAXG*PIPES];
static int region_steps(terra_ctx *ctx, uint32_t index, void *user) {
	region_t *r = &regions[index];
	(void)user;
	for (int s = 0; s < r->steps; ++s) {
		float mn = 0.0f, mx = 0.0f;
		int rc = terra_gen_grid_minmax_dev(ctx, -0.5f*(float)r->n + (float)r->gpu_slot*(float)r->n, -0.5f*(float)r->n, r->st.DX_VAL, r->st.DY_VAL, (uint32_t)r->n, (uint32_t)r->n, TERRA_GEN_GLACIATE, 0, r->z, &mn, &mx);
		if (rc == 0) {rc = terra_apply_erosion_dev(ctx, r->z, r->n, r->n, mn, (uint32_t)r->droplets, TERRA_ERODE_MINZ_IS_MIN);}
		if (rc) return rc;
	}
	return terra_synchronize(ctx);
}

int main(int argc, char **argv) {
	int same = 0, pos = 0, G = 0, steps = 16, n = 16384;
	for (int i = 1; i < argc; ++i) {
		if (strcmp(argv[i], "--same-device") == 0) {same = 1; continue;}
		int const v = atoi(argv[i]);
		if (pos == 0) {G = v;} else if (pos == 1) {steps = v;} else if (pos == 2) {n = v;}
		++pos;
	}
	int const ndev = terra_device_count();
	if (ndev < 1) {fprintf(stderr, "no HIP device (there is no CPU fall-back)\n"); return 3;}
	if (G <= 0) {G = ndev;}
	if (G > MAXG || (!same && G > ndev) || steps < 1 || n < 256) {fprintf(stderr, "bad arguments (%d GPUs present)\n", ndev); return 2;}
	terra_config c; memset(&c, 0, sizeof(c)); /* the synthetic scene of BASELINE.md section 3 (scene_config/config.txt:56-97), 8 octaves */
	c.mesh_x = c.mesh_y = 128; c.scene_x = c.scene_y = c.scene_z = 4.0f; c.mesh_height = 0.7f; c.mesh_scale = 1.0f;
	c.mesh_seed = 1; c.mesh_freq_filter = 1; c.mesh_gen_mode = TERRA_MGEN_SINE; c.mesh_gen_shape = 0; c.glaciate = 1;
	c.hmap[0] = 1000.0f; c.hmap[4] = 1000.0f; c.hmap[9] = 5.0f; c.hmap[10] = 0.001f; c.hmap[11] = -4.0f;
	c.erode_amount = 1.0f; c.start_mag = 0.02f; c.start_freq = 240.0f; c.mag_mult = 2.0f; c.freq_mult = 0.5f;

	/* ---- (a) one region per GPU, PIPES heightmaps in flight on each */
	int devs[MAXG*PIPES];
	for (int g = 0; g < G; ++g) {for (int p = 0; p < PIPES; ++p) {devs[g*PIPES + p] = same ? 0 : g;}}
	terra_multi *mp = NULL;
	CK(terra_multi_create(&mp, devs, (uint32_t)(G*PIPES)));
	CK(terra_multi_init_scene(mp, &c));
	for (int i = 0; i < G*PIPES; ++i) {
		region_t *r = &regions[i];
		r->n = n; r->droplets = 1000; r->gpu_slot = i/PIPES;
		CK(terra_get_state(terra_multi_ctx(mp, (uint32_t)i), &r->st));
		CK(terra_malloc(terra_multi_ctx(mp, (uint32_t)i), (void **)&r->z, (size_t)n*(size_t)n*sizeof(float)));
		r->steps = 2;
	}
	CK(terra_multi_foreach(mp, region_steps, NULL)); /* warm-up: scratch allocation, graph capture */
	int const per_pipe = (steps + PIPES - 1)/PIPES;
	for (int i = 0; i < G*PIPES; ++i) {regions[i].steps = per_pipe;}
	double t0 = now();
	CK(terra_multi_foreach(mp, region_steps, NULL));
	double dt = now() - t0;
	printf("{\"what\": \"heightmap regions\", \"gpus\": %d, \"same_device\": %d, \"grid\": %d, \"maps\": %d, \"ms_per_map_per_gpu\": %.4f, \"gcells_per_s\": %.2f, \"scaling\": \"weak\"}\n",
		G, same, n, G*PIPES*per_pipe, 1e3*dt/(PIPES*per_pipe), (double)n*(double)n*(double)(G*PIPES*per_pipe)/dt/1e9);
	for (int i = 0; i < G*PIPES; ++i) {terra_free(terra_multi_ctx(mp, (uint32_t)i), regions[i].z);}
	terra_multi_destroy(mp);

	/* ---- (b) .. (e): one context per GPU */
	for (int g = 0; g < G; ++g) {devs[g] = same ? 0 : g;}
	terra_multi *m = NULL;
	CK(terra_multi_create(&m, devs, (uint32_t)G));
	CK(terra_multi_init_scene(m, &c));
	uint32_t const NT = 64*64;
	int32_t *tiles = (int32_t *)malloc(NT*2*sizeof(int32_t));
	for (int ty = -32, k = 0; ty < 32; ++ty) {for (int tx = -32; tx < 32; ++tx, ++k) {tiles[2*k] = tx; tiles[2*k+1] = ty;}}
	float *dz[MAXG]; terra_tile_stats *dst[MAXG]; uint8_t *dnm[MAXG]; float *dmn[MAXG];
	for (int g = 0; g < G; ++g) {
		uint32_t first, cnt; terra_multi_partition(NT, (uint32_t)G, (uint32_t)g, &first, &cnt);
		terra_ctx *ctx = terra_multi_ctx(m, (uint32_t)g);
		size_t const k = cnt ? cnt : 1;
		CK(terra_malloc(ctx, (void **)&dz[g], k*130*130*4)); CK(terra_malloc(ctx, (void **)&dst[g], k*sizeof(terra_tile_stats)));
		CK(terra_malloc(ctx, (void **)&dnm[g], k*129*129*4)); CK(terra_malloc(ctx, (void **)&dmn[g], k*4));
	}
	for (int droplets = 0; droplets <= 1000; droplets += 1000) {
		int const reps = droplets ? 2 : 8;
		CK(terra_multi_tiles_create_zvals_dev(m, tiles, NT, (uint32_t)droplets, dz, dst, dnm, dmn)); CK(terra_multi_synchronize(m));
		t0 = now();
		for (int r = 0; r < reps; ++r) {CK(terra_multi_tiles_create_zvals_dev(m, tiles, NT, (uint32_t)droplets, dz, dst, dnm, dmn));}
		CK(terra_multi_synchronize(m));
		dt = now() - t0;
		printf("{\"what\": \"64x64 tile batch\", \"gpus\": %d, \"droplets_per_tile\": %d, \"ms_per_batch\": %.3f, \"gcells_per_s\": %.3f, \"scaling\": \"strong\"}\n",
			G, droplets, 1e3*dt/reps, (double)NT*130.0*130.0*reps/dt/1e9);
	}
	/* (e) mesh shadows over the whole terrain (host zvals in, host masks out: the call uploads each strip and returns the masks) */
	{
		float *hz = (float *)malloc((size_t)NT*130*130*4); uint8_t *hs = (uint8_t *)malloc((size_t)NT*130*130);
		for (int g = 0; g < G; ++g) {
			uint32_t first, cnt; terra_multi_partition(NT, (uint32_t)G, (uint32_t)g, &first, &cnt);
			if (cnt) {CK(terra_memcpy_d2h(terra_multi_ctx(m, (uint32_t)g), hz + (size_t)first*130*130, dz[g], (size_t)cnt*130*130*4));}
		}
		float const light[3] = {0.6f, 0.5f, 0.4f};
		CK(terra_multi_tiles_mesh_shadows(m, tiles, NT, hz, light, hs));
		t0 = now();
		CK(terra_multi_tiles_mesh_shadows(m, tiles, NT, hz, light, hs));
		dt = now() - t0;
		size_t shadowed = 0; for (size_t i = 0; i < (size_t)NT*130*130; ++i) {shadowed += hs[i] != 0;}
		printf("{\"what\": \"mesh shadows 64x64 tiles, host zvals in / host masks out (277 MB up, 69 MB down)\", \"gpus\": %d, \"ms\": %.2f, \"shadowed_cells\": %zu}\n", G, 1e3*dt, shadowed);
		/* the device-resident form: the terrain lies on the GPUs as strips of tile columns (terra_multi_shadow_layout), only the border edges move */
		uint32_t *own = (uint32_t *)malloc(NT*4), *pos = (uint32_t *)malloc(NT*4), per[MAXG];
		CK(terra_multi_shadow_layout(m, tiles, NT, light, own, pos, per));
		float *sz[MAXG]; uint8_t *ssm[MAXG];
		for (int g = 0; g < G; ++g) {
			terra_ctx *ctx = terra_multi_ctx(m, (uint32_t)g);
			size_t const k = per[g] ? per[g] : 1;
			CK(terra_malloc(ctx, (void **)&sz[g], k*130*130*4)); CK(terra_malloc(ctx, (void **)&ssm[g], k*130*130));
		}
		for (uint32_t i = 0; i < NT; ++i) {CK(terra_memcpy_h2d(terra_multi_ctx(m, own[i]), sz[own[i]] + (size_t)pos[i]*130*130, hz + (size_t)i*130*130, (size_t)130*130*4));}
		CK(terra_multi_tiles_mesh_shadows_dev(m, tiles, NT, sz, light, ssm)); CK(terra_multi_synchronize(m));
		int const sreps = 4;
		t0 = now();
		for (int r = 0; r < sreps; ++r) {CK(terra_multi_tiles_mesh_shadows_dev(m, tiles, NT, sz, light, ssm));}
		CK(terra_multi_synchronize(m));
		dt = (now() - t0)/sreps;
		size_t shadowed2 = 0;
		{uint8_t *tmp = (uint8_t *)malloc((size_t)130*130);
		 for (uint32_t i = 0; i < NT; ++i) {CK(terra_memcpy_d2h(terra_multi_ctx(m, own[i]), tmp, ssm[own[i]] + (size_t)pos[i]*130*130, (size_t)130*130)); for (int j = 0; j < 130*130; ++j) {shadowed2 += tmp[j] != 0;}}
		 free(tmp);}
		printf("{\"what\": \"mesh shadows 64x64 tiles, device resident strips\", \"gpus\": %d, \"ms\": %.2f, \"shadowed_cells\": %zu, \"same_as_host_form\": %d, \"exchange\": \"event + one gather launch per chunk over peer-mapped edge buffers\"}\n", G, 1e3*dt, shadowed2, shadowed2 == shadowed);
		{ /* one context, the same terrain (terra_tiles_mesh_shadows_dev): what not sharding costs */
			terra_ctx *c0 = terra_multi_ctx(m, 0); float *z1; uint8_t *s1;
			CK(terra_malloc(c0, (void **)&z1, (size_t)NT*130*130*4)); CK(terra_malloc(c0, (void **)&s1, (size_t)NT*130*130));
			CK(terra_memcpy_h2d(c0, z1, hz, (size_t)NT*130*130*4));
			CK(terra_tiles_mesh_shadows_dev(c0, tiles, NT, z1, light, s1)); CK(terra_synchronize(c0));
			t0 = now();
			for (int r = 0; r < sreps; ++r) {CK(terra_tiles_mesh_shadows_dev(c0, tiles, NT, z1, light, s1));}
			CK(terra_synchronize(c0));
			printf("{\"what\": \"mesh shadows 64x64 tiles, ONE context\", \"ms\": %.2f}\n", 1e3*(now() - t0)/sreps);
			terra_free(c0, z1); terra_free(c0, s1);
		}
		for (int g = 0; g < G; ++g) {terra_free(terra_multi_ctx(m, (uint32_t)g), sz[g]); terra_free(terra_multi_ctx(m, (uint32_t)g), ssm[g]);}
		free(own); free(pos);
		free(hz); free(hs);
	}
	for (int g = 0; g < G; ++g) {terra_ctx *ctx = terra_multi_ctx(m, (uint32_t)g); terra_free(ctx, dz[g]); terra_free(ctx, dst[g]); terra_free(ctx, dnm[g]); terra_free(ctx, dmn[g]);}
	free(tiles);
	/* (c) one heightmap as row strips */
	{
		float *strip[MAXG];
		terra_state st; CK(terra_get_state(terra_multi_ctx(m, 0), &st));
		for (int g = 0; g < G; ++g) {uint32_t first, cnt; terra_multi_partition((uint32_t)n, (uint32_t)G, (uint32_t)g, &first, &cnt); CK(terra_malloc(terra_multi_ctx(m, (uint32_t)g), (void **)&strip[g], (size_t)(cnt ? cnt : 1)*(size_t)n*4));}
		float mn = 0.0f, mx = 0.0f;
		CK(terra_multi_gen_grid_rows_dev(m, -0.5f*(float)n, -0.5f*(float)n, st.DX_VAL, st.DY_VAL, (uint32_t)n, (uint32_t)n, TERRA_GEN_GLACIATE, 0, strip, &mn, &mx));
		t0 = now();
		for (int r = 0; r < steps; ++r) {CK(terra_multi_gen_grid_rows_dev(m, -0.5f*(float)n, -0.5f*(float)n, st.DX_VAL, st.DY_VAL, (uint32_t)n, (uint32_t)n, TERRA_GEN_GLACIATE, 0, strip, &mn, &mx));}
		dt = now() - t0;
		printf("{\"what\": \"one heightmap as row strips\", \"gpus\": %d, \"grid\": %d, \"ms\": %.4f, \"gcells_per_s\": %.2f, \"min\": %.9g, \"max\": %.9g, \"scaling\": \"strong\"}\n", G, n, 1e3*dt/steps, (double)n*(double)n*steps/dt/1e9, mn, mx);
		for (int g = 0; g < G; ++g) {terra_free(terra_multi_ctx(m, (uint32_t)g), strip[g]);}
	}
	/* (c2) ONE heightmap with its erosion: the row strips are physical allocations on their GPUs mapped back to back (terra_multi_dgrid_create), every context fills its
	 * rows, context (step mod G) erodes the whole map through the one pointer -- remote rows over xGMI */
	{
		terra_state st; CK(terra_get_state(terra_multi_ctx(m, 0), &st));
		size_t const gran = terra_dgrid_granularity(terra_multi_ctx(m, 0));
		size_t sb[MAXG]; uint32_t r0[MAXG], rn[MAXG];
		int ok = 1;
		for (int g = 0; g < G; ++g) {terra_multi_partition((uint32_t)n, (uint32_t)G, (uint32_t)g, &r0[g], &rn[g]); sb[g] = (size_t)rn[g]*(size_t)n*4; if (sb[g] == 0 || sb[g] % gran) ok = 0;}
		if (ok) {
			terra_dgrid *dg = NULL; void *base = NULL;
			CK(terra_multi_dgrid_create(m, sb, &dg, &base));
			float *grid = (float *)base;
			double tsum = 0.0;
			for (int r = 0; r < steps + 2; ++r) {
				double const ta = now();
				float mn = 1e30f;
				for (int g = 0; g < G; ++g) { /* (a 3DWorld process would drive these from its worker threads; here one after another: the timing is the eroded map's, not the strips') */
					float a = 0.0f, b = 0.0f;
					CK(terra_gen_grid_rows_minmax_dev(terra_multi_ctx(m, (uint32_t)g), -0.5f*(float)n, -0.5f*(float)n, st.DX_VAL, st.DY_VAL, (uint32_t)n, (uint32_t)n, TERRA_GEN_GLACIATE, 0, r0[g], rn[g], grid + (size_t)r0[g]*(size_t)n, &a, &b));
					if (a < mn) mn = a;
				}
				terra_ctx *e = terra_multi_ctx(m, (uint32_t)(r % G));
				CK(terra_apply_erosion_dev(e, grid, n, n, mn, 1000, TERRA_ERODE_MINZ_IS_MIN)); CK(terra_synchronize(e));
				if (r >= 2) tsum += now() - ta;
			}
			printf("{\"what\": \"one heightmap on a distributed grid, noise strips + whole-map erosion by one context, serial\", \"gpus\": %d, \"grid\": %d, \"ms\": %.4f, \"gcells_per_s\": %.2f}\n", G, n, 1e3*tsum/steps, (double)n*(double)n*steps/tsum/1e9);
			terra_dgrid_destroy(dg);
		}
	}
	/* (d) one 512^3 voxel field as y slabs */
	{
		uint32_t const VN = 512;
		float *slab[MAXG];
		for (int g = 0; g < G; ++g) {uint32_t first, cnt; terra_multi_partition(VN, (uint32_t)G, (uint32_t)g, &first, &cnt); CK(terra_malloc(terra_multi_ctx(m, (uint32_t)g), (void **)&slab[g], (size_t)(cnt ? cnt : 1)*VN*VN*4));}
		float const lo[3] = {-1.0f, -1.0f, -0.25f}, vsz[3] = {2.0f/VN, 2.0f/VN, 0.5f/VN}, off[3] = {0.0f, 0.0f, 0.0f};
		CK(terra_multi_voxel_fill_dev(m, slab, VN, VN, VN, lo, vsz, off, 1.0f, 1.0f, 123, 456, TERRA_MGEN_SINE, 0.0f, 1)); CK(terra_multi_synchronize(m));
		t0 = now();
		for (int r = 0; r < steps; ++r) {CK(terra_multi_voxel_fill_dev(m, slab, VN, VN, VN, lo, vsz, off, 1.0f, 1.0f, 123, 456, TERRA_MGEN_SINE, 0.0f, 1));}
		CK(terra_multi_synchronize(m));
		dt = now() - t0;
		printf("{\"what\": \"512^3 voxel field as y slabs\", \"gpus\": %d, \"ms\": %.4f, \"gvoxels_per_s\": %.2f, \"scaling\": \"strong\"}\n", G, 1e3*dt/steps, (double)VN*VN*VN*steps/dt/1e9);
		for (int g = 0; g < G; ++g) {terra_free(terra_multi_ctx(m, (uint32_t)g), slab[g]);}
	}
	terra_multi_destroy(m);
	return 0;
}
