#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs 2-5) on one GPU: device-resident throughput with HIP-event timing.
Not the headline metric (bench.py); used to find the next kernel to work on.  Prints one JSON object."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("3dworld_amd")
t = pkg.Terra(0)
out = {}

def timed(fn, reps=3, warm=1, spin_ms=100.0):
    # warm calls for at least spin_ms: a cold or briefly idle chip runs 15-20 % below its sustained clock for the first ~30 ms of load (profiles/r04_clock_ramp.txt)
    t_s = time.perf_counter()
    for _ in range(warm): fn()
    while (time.perf_counter() - t_s) * 1e3 < spin_ms: fn()
    t.synchronize(); t.timer_start()
    for _ in range(reps): fn()
    return t.timer_stop() / reps

# C2: 4096^2 single heightmap, every mode, 9 octaves (reference default) and 8
N = 4096
z = t.alloc(N * N * 4)
for name, mode in (("sine", 0), ("simplex", 1), ("perlin", 2), ("dwarp", 4)):
    for octv in (9, 8):
        st = t.init_scene(pkg.make_config(mesh_gen_mode=mode, mesh_freq_filter=9 - octv))
        ms = timed(lambda: t.gen_grid_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE))
        out[f"C2_{name}_{octv}oct_4096_gcells_s"] = round(N * N / ms / 1e6, 2)
# C3: 4096^2 + erosion, 1e3 / 1e5 / 1e6 droplets
st = t.init_scene(pkg.make_config(mesh_gen_mode=0))
for D in (1000, 100000, 1000000):
    if D == 1000:  # first call of the context: scratch allocation and hipGraph capture, not part of the steady state
        mn, mx = t.gen_grid_minmax_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        t.apply_erosion_dev(z.ptr, N, N, mn, D, pkg.ERODE_MINZ_IS_MIN)
    mn, mx = t.gen_grid_minmax_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
    t.synchronize(); t0 = time.perf_counter()
    t.apply_erosion_dev(z.ptr, N, N, mn, D, pkg.ERODE_MINZ_IS_MIN); t.synchronize()
    dt = time.perf_counter() - t0
    r = t.erosion_report().as_dict()
    out[f"C3_erosion_4096_{D}_droplets"] = {"ms": round(dt * 1e3, 2), "droplets_per_s": round(D / dt), "steps_per_s": round(r["steps"] / dt), "rounds": r["rounds"], "windows": r["windows"], "traces": r["traces"], "fallbacks": r["serial_fallbacks"]}
z.free()
# C3, literally: the heightmap_island_eroded preset -- heights loaded from heightmaps/heightmap_island_1k.png (`mh_filename ... 180.3 -18.75`, scene_config/
# config_heightmap.txt:84) through heightmap_t::postprocess_height (pixels -> floats -> whole-image erosion -> pixels), 10^5 and 10^6 droplets (config_heightmap.txt:78)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc_
png = os.path.join(ROOT, "tests", "golden", "heightmap_island_1k.png")
if os.path.exists(png):
    pix = pkg.terra.read_png(png, lib=t.lib)
    hh, ww = pix.shape
    pc_.island_setup(t, pc_.island_cfg(pkg.make_config))
    dp = t.alloc(pix.nbytes); dv = t.alloc(ww * hh * 4)
    dp.upload(pix); t.heightmap_to_floats_dev(dp.ptr, ww, hh, 1, dv.ptr)
    mn_, mx_ = t.minmax_dev(dv.ptr, ww * hh)
    pc_.island_setup(t, pc_.island_cfg(pkg.make_config), (mn_, mx_))
    for D in (100000, 1000000):
        for rep in range(2):
            dp.upload(pix); t.synchronize(); t0 = time.perf_counter()
            bad = t.heightmap_postprocess_dev(dp.ptr, ww, hh, 1, D, dv.ptr); t.synchronize()
            dt = time.perf_counter() - t0
        r = t.erosion_report().as_dict()
        out[f"C3_island_1k_postprocess_height_{D}_droplets"] = {"ms": round(dt * 1e3, 2), "droplets_per_s": round(D / dt), "steps": r["steps"], "rounds": r["rounds"], "out_of_range_pixels": bad}
    dp.free(); dv.free()
# C4: 64x64 tiles of 128^2
tiles = np.array([(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)], np.int32)
n = len(tiles)
zt = t.alloc(n * 130 * 130 * 4); stt = t.alloc(n * 160); nm = t.alloc(n * 129 * 129 * 4); mz = t.alloc(n * 4)
for name, mode in (("sine", 0), ("simplex", 1)):
    t.init_scene(pkg.make_config(mesh_gen_mode=mode))
    for it in (0, 1000):
        if mode == 1 and it: continue
        ms = timed(lambda: t.tiles_create_zvals_dev(tiles, it, zt.ptr, stt.ptr, nm.ptr, mz.ptr), reps=2)
        out[f"C4_tiles64x64_{name}_{it}iters"] = {"ms": round(ms, 2), "tiles_per_s": round(n / ms * 1e3), "gcells_s": round(n * 130 * 130 / ms / 1e6, 3)}
# the same batch with NO ocean (water plane far below the terrain): every droplet of every tile walks until it deposits -- the worst case for a kernel that holds two tiles per CU
t.init_scene(pkg.make_config(mesh_gen_mode=0))
t.set_water_plane_z(-1.0e3)
ms = timed(lambda: t.tiles_create_zvals_dev(tiles, 1000, zt.ptr, stt.ptr, nm.ptr, mz.ptr), reps=2, spin_ms=0.0)
out["C4_tiles64x64_sine_1000iters_all_land"] = {"ms": round(ms, 2), "tiles_per_s": round(n / ms * 1e3), "gcells_s": round(n * 130 * 130 / ms / 1e6, 3)}
# row f1: AO lighting for the same 64x64 tiles (201^2 context per tile + 8 x 8 ray march per texel)
t.init_scene(pkg.make_config(mesh_gen_mode=0))
t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr)
ao = t.alloc(n * 129 * 129)
ms = timed(lambda: t.tiles_ao_lighting_dev(tiles, zt.ptr, ao.ptr), reps=2)
out["F1_tile_ao_64x64_sine"] = {"ms": round(ms, 2), "tiles_per_s": round(n / ms * 1e3)}
# row f3: landscape weights texture of the same batch (second noise field + biome parameters + per-texel blend + grass blocks)
t.set_landscape(pkg.make_landscape(grass_density=100))
wt = t.alloc(n * 129 * 129 * 4); gbk = t.alloc(n * 32 * 32 * 12); hgr = t.alloc(n)
ms = timed(lambda: t.tiles_create_weights_dev(tiles, zt.ptr, wt.ptr, gbk.ptr, hgr.ptr), reps=2)
out["F3_tile_weights_64x64_sine"] = {"ms": round(ms, 2), "tiles_per_s": round(n / ms * 1e3)}
t.set_landscape(pkg.make_landscape())
# row f2: mesh shadows of the 64x64 batch for a low sun (127 dependency levels along the anti-diagonals)
t.init_scene(pkg.make_config(mesh_gen_mode=0))
t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr)
sm = t.alloc(n * 130 * 130)
ms = timed(lambda: t.tiles_mesh_shadows_dev(tiles, zt.ptr, (0.6, 0.5, 0.4), sm.ptr), reps=2)
out["F2_tile_mesh_shadows_64x64"] = {"ms": round(ms, 2), "tiles_per_s": round(n / ms * 1e3)}
sm.free()
# tiles served from a heightmap texture: proc_gen a 4096^2 map (noise + 1000-droplet erosion + 16-bit quantise) on the device, then sample the tile batch from it
H = 4096
hv = t.alloc(H * H * 4); hp = t.alloc(H * H * 2)
rng = t.heightmap_proc_gen_dev(hv.ptr, H, H, 1000, hp.ptr)
if rng is not None:
    t.hmap_set_dev(hp.ptr, H, H, 2, float(rng[0]), float(np.float32(np.float64(rng[1]) / 255.0)))
    ms = timed(lambda: t.tiles_create_zvals_dev(tiles, 0, zt.ptr, stt.ptr, nm.ptr, mz.ptr), reps=2)
    out["C4h_tiles64x64_from_heightmap_texture"] = {"ms": round(ms, 2), "tiles_per_s": round(n / ms * 1e3)}
    t.hmap_set_dev(None)
hv.free(); hp.free()
ao.free()
for b in (zt, stt, nm, mz): b.free()
# C5: voxels
t.init_scene(pkg.make_config(mesh_gen_mode=0))
lo, vsz, off = (-15.9, -15.9, -1.0), (0.0622, 0.0622, 0.0625), (0.0, 0.0, 0.0)
for dims in ((512, 512, 64), (512, 512, 512)):
    v = t.alloc(dims[0] * dims[1] * dims[2] * 4)
    for name, mode in (("sines", 0), ("simplex", 1), ("perlin", 2)):
        ms = timed(lambda: t.voxel_fill_dev(v.ptr, dims[0], dims[1], dims[2], lo, vsz, off, 1.0, 1.0, 123, 456, mode, 0.01, 1), reps=2)
        out[f"C5_voxels_{dims[0]}x{dims[1]}x{dims[2]}_{name}_gvoxels_s"] = round(dims[0] * dims[1] * dims[2] / ms / 1e6, 2)
    v.free()
# ---- the reference's CPU path beside each number: bounded samples (seconds each) on this host's cores.  kind = "reference" when oracle/_ref
# (the reference's own TUs) travelled with the repo, else "port" (oracle/terra_oracle.c).  Erosion: 1 thread is the only deterministic order.
if "--no-cpu" not in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    orclib.build_oracle()
    kind = "reference" if orclib.ref_available() else "port"
    ck = orclib.Checker("ref" if kind == "reference" else "orc")
    cores = ck.num_threads()
    cpu = {"kind": kind, "cores": cores}
    def wall(fn):
        t0 = time.perf_counter(); r = fn(); return time.perf_counter() - t0, r
    for name, mode, n in (("sine", 0, 4096), ("simplex", 1, 2048), ("perlin", 2, 2048), ("dwarp", 4, 1024)):
        s_ = ck.init(orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
        dt, _ = wall(lambda: ck.gen_grid(-n / 2, -n / 2, s_.DX_VAL, s_.DY_VAL, n, n, 1))
        cpu[f"C2_{name}_8oct_{n}_gcells_s"] = round(n * n / dt / 1e9, 5)
    s_ = ck.init(orclib.make_config(mesh_gen_mode=0))
    n = 4096
    g0 = ck.gen_grid(-n / 2, -n / 2, s_.DX_VAL, s_.DY_VAL, n, n, 1)
    for D, thr in ((1000, cores), (100000, cores), (100000, 1), (1000000, 1)):
        ck.set_num_threads(thr)
        g = g0.copy()
        dt, _ = wall(lambda: ck.apply_erosion(g, float(g0.min()), D))
        cpu[f"C3_erosion_4096_{D}_droplets_{thr}thr"] = {"ms": round(dt * 1e3, 1), "droplets_per_s": round(D / dt), "deterministic": thr == 1}
    if os.path.exists(png):  # the reference's own heightmap_t::postprocess_height on the same image, one thread (the only deterministic droplet order)
        ck.set_num_threads(1)
        pc_.island_setup(ck, pc_.island_cfg(orclib.make_config))
        v_ = ck.heightmap_to_floats(pix)
        pc_.island_setup(ck, pc_.island_cfg(orclib.make_config), (v_.min(), v_.max()))
        for D in (100000, 1000000):
            dt, _ = wall(lambda: ck.heightmap_postprocess(pix, D))
            cpu[f"C3_island_1k_postprocess_height_{D}_droplets_1thr"] = {"ms": round(dt * 1e3, 1), "droplets_per_s": round(D / dt)}
        ck.set_mesh_file_scale(1.0, 0.0)
        s_ = ck.init(orclib.make_config(mesh_gen_mode=0))
    ck.set_num_threads(8)  # 130 rows per tile: more threads only add fork/join cost (256 threads: 8 tiles/s)
    nt = 64
    dt, _ = wall(lambda: [ck.tile_create_zvals(tx, ty, 0) for ty in range(-4, 4) for tx in range(-4, 4)])
    cpu["C4_tiles_sine_0iters_tiles_per_s_8thr"] = round(nt / dt)
    dt, _ = wall(lambda: [ck.tile_create_zvals(tx, ty, 1000) for ty in range(-2, 2) for tx in range(-2, 2)])
    cpu["C4_tiles_sine_1000iters_tiles_per_s_8thr"] = round(16 / dt, 1)
    ck.set_num_threads(4)  # calc_mesh_ao_lighting: "#pragma omp parallel num_threads(4)" (src/tiled_mesh.cpp:614-616)
    zs_ = [ck.tile_create_zvals(tx, ty, 0)[0] for ty in range(-2, 2) for tx in range(-2, 2)]
    dt, _ = wall(lambda: [ck.tile_ao_lighting(tx, ty, zs_[(ty + 2) * 4 + (tx + 2)]) for ty in range(-2, 2) for tx in range(-2, 2)])
    cpu["F1_tile_ao_tiles_per_s_4thr"] = round(16 / dt, 1)
    ck.set_landscape(orclib.make_landscape(grass_density=100))  # create_texture: 2 threads for the noise field, the blend is serial (src/tiled_mesh.cpp:1111,1117)
    dt, _ = wall(lambda: [ck.tile_create_weights(tx, ty, zs_[(ty + 2) * 4 + (tx + 2)]) for ty in range(-2, 2) for tx in range(-2, 2)])
    cpu["F3_tile_weights_tiles_per_s_1thr"] = round(16 / dt, 1)
    ck.set_landscape(orclib.make_landscape())
    tl_ = [(tx, ty) for ty in range(-4, 4) for tx in range(-4, 4)]
    dt, _ = wall(lambda: ck.tiles_mesh_shadows(tl_, np.stack([ck.tile_create_zvals(tx, ty, 0)[0] for tx, ty in tl_]), (0.6, 0.5, 0.4)))
    cpu["F2_tile_mesh_shadows_tiles_per_s_incl_zvals"] = round(len(tl_) / dt, 1)
    # ---- the same rows with ALL host cores: the engine makes tiles one after another, each with an OpenMP loop over its 130 rows (src/tiled_mesh.cpp:495, 2408-2413) -- that is
    # the "_8thr" numbers above (more threads only add fork/join cost).  The best a CPU port could do is a tile per core: a pool of host threads, one tile each, OpenMP off.
    from concurrent.futures import ThreadPoolExecutor
    pool_n = max(1, min(len(os.sched_getaffinity(0)), 128))
    def pool_map(fn, items):
        def run(it):
            ck.set_num_threads(1)  # omp_set_num_threads is per calling thread
            return fn(it)
        with ThreadPoolExecutor(pool_n) as ex:
            return list(ex.map(run, items))
    cpu["allthr_pool_threads"] = pool_n
    tl2 = [(tx, ty) for ty in range(-8, 8) for tx in range(-16, 16)]  # 512 tiles around the origin (land and ocean mixed, like the GPU batch)
    dt, zs2 = wall(lambda: pool_map(lambda xy: ck.tile_create_zvals(xy[0], xy[1], 0)[0], tl2))
    cpu["C4_tiles_sine_0iters_tiles_per_s_allthr"] = round(len(tl2) / dt)
    dt, _ = wall(lambda: pool_map(lambda xy: ck.tile_create_zvals(xy[0], xy[1], 1000), tl2))
    cpu["C4_tiles_sine_1000iters_tiles_per_s_allthr"] = round(len(tl2) / dt, 1)
    dt, _ = wall(lambda: pool_map(lambda i: ck.tile_ao_lighting(tl2[i][0], tl2[i][1], zs2[i]), range(len(tl2))))
    cpu["F1_tile_ao_tiles_per_s_allthr"] = round(len(tl2) / dt, 1)
    ck.set_landscape(orclib.make_landscape(grass_density=100))
    dt, _ = wall(lambda: pool_map(lambda i: ck.tile_create_weights(tl2[i][0], tl2[i][1], zs2[i]), range(len(tl2))))
    cpu["F3_tile_weights_tiles_per_s_allthr"] = round(len(tl2) / dt, 1)
    ck.set_landscape(orclib.make_landscape())
    ck.set_num_threads(cores)
    dt, _ = wall(lambda: ck.tiles_mesh_shadows(tl2, np.stack(zs2), (0.6, 0.5, 0.4)))  # the sweeps chain from tile to tile (sh_out -> sh_in, src/tiled_mesh.cpp:664-692): serial in the reference too
    cpu["F2_tile_mesh_shadows_tiles_per_s_zvals_given"] = round(len(tl2) / dt, 1)
    for name, mode, nn in (("simplex", 1, 4096), ("perlin", 2, 4096), ("dwarp", 4, 2048)):  # (the fBm rows above are all-core already: OpenMP over the rows of one grid; larger grids so that 256 threads have rows to share)
        s_ = ck.init(orclib.make_config(mesh_gen_mode=mode, mesh_freq_filter=1))
        dt, _ = wall(lambda: ck.gen_grid(-nn / 2, -nn / 2, s_.DX_VAL, s_.DY_VAL, nn, nn, 1))
        cpu[f"C2_{name}_8oct_{nn}_gcells_s_allthr"] = round(nn * nn / dt / 1e9, 5)
    s_ = ck.init(orclib.make_config(mesh_gen_mode=0))
    ck.set_num_threads(cores)
    dims = (256, 256, 64)
    dt, _ = wall(lambda: ck.voxel_fill(dims[0], dims[1], dims[2], lo, vsz, off, 1.0, 1.0, 123, 456, 0, 0.01, 1))
    cpu["C5_voxels_256x256x64_sines_gvoxels_s"] = round(dims[0] * dims[1] * dims[2] / dt / 1e9, 4)
    dims = (512, 512, 64)  # (OpenMP over y, src/voxels.cpp:312: 512 rows for the host's threads)
    dt, _ = wall(lambda: ck.voxel_fill(dims[0], dims[1], dims[2], lo, vsz, off, 1.0, 1.0, 123, 456, 0, 0.01, 1))
    cpu["C5_voxels_512x512x64_sines_gvoxels_s_allthr"] = round(dims[0] * dims[1] * dims[2] / dt / 1e9, 4)
    for name, mode in (("simplex", 1), ("perlin", 2)):  # the lattice fields (glm, src/voxels.cpp:328-338): the reference on all host threads
        dims = (256, 256, 64)
        dt, _ = wall(lambda: ck.voxel_fill(dims[0], dims[1], dims[2], lo, vsz, off, 1.0, 1.0, 123, 456, mode, 0.01, 1))
        cpu[f"C5_voxels_256x256x64_{name}_gvoxels_s_allthr"] = round(dims[0] * dims[1] * dims[2] / dt / 1e9, 4)
    out["cpu"] = cpu
print(json.dumps(out))
