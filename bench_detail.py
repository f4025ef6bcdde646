"""bench_detail.py -- the measurements bench.py reports under `detail` beside the headline (same run, same contexts): the sharded workloads (strips / tiles / voxels),
one heightmap in flight, the other noise modes, dense erosion, the END-TO-END rates with the z grid delivered to host memory, and the per-rank step floor of the
one-grid line at a simulated world size.  `env` is bench.py's namespace of the run (contexts, grids, the timed() helper ...)."""
import os
import time

FP32_PEAK_TFLOPS = 157.3


def flops_per_cell(mode, octaves):
    """sine 2 flop per term (10 terms per octave); fBm 70 flop per octave and evaluation, domain warp = 5 evaluations."""
    return 20.0 * octaves if mode == 0 else 70.0 * octaves * (5 if mode == 4 else 1)


def strips_steps_fn(env):
    """strong scaling of ONE grid, the noise + min half: row strips + all_reduce(min) of one float"""
    pkg, t, z, st, N, dist = env.pkg, env.t, env.z, env.st, env.N, env.dist
    r0, r1 = env.dmod.strip_rows(N, env.rank, env.world)
    red = env.torch.zeros(1, dtype=env.torch.float32, device=env.coll_dev)

    def fn(k):
        for _ in range(k):
            mn, _ = t.gen_grid_rows_minmax_dev(z.data_ptr(), -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, r0, r1 - r0, pkg.GEN_GLACIATE)
            if env.have_group:
                red[0] = mn
                dist.all_reduce(red, op=dist.ReduceOp.MIN)  # min(vals) of the whole map: what run_erosion / from_floats need next
                mn = float(red.item())
    return fn, r1 - r0


def tiles_steps_fn(env, droplets):
    """strong scaling of BASELINE config 4: 64 x 64 tiles block-partitioned, no collective"""
    torch, t, dev = env.torch, env.t, env.dev
    my_tiles, bufs = env.my_tiles, env.tile_bufs
    nt = len(my_tiles)
    import numpy as np
    tiles_np = np.ascontiguousarray(my_tiles, np.int32).reshape(-1, 2)  # once: a Python list of 4096 tuples takes ~0.6-1.1 ms to convert PER CALL, more than the batch takes on the GPU

    def fn(k):
        if nt == 0:
            return
        if not bufs:
            bufs["z"] = torch.empty(nt * 130 * 130, dtype=torch.float32, device=dev)
            bufs["st"] = torch.empty(nt * 39 * 4, dtype=torch.uint8, device=dev)
            bufs["nm"] = torch.empty(nt * 129 * 129 * 4, dtype=torch.uint8, device=dev)
            bufs["mnz"] = torch.empty(nt, dtype=torch.float32, device=dev)
        for _ in range(k):
            t.tiles_create_zvals_dev(tiles_np, droplets, bufs["z"].data_ptr(), bufs["st"].data_ptr(), bufs["nm"].data_ptr(), bufs["mnz"].data_ptr())
        t.synchronize()
    return fn


def sharded(env, detail, run_steps):
    """single / strips / tiles / voxels: all ranks take part"""
    args, K, P, world, cells = env.args, env.args.steps, env.P, env.world, env.cells
    ke = max(4, min(K, 16))
    if args.workload != "heightmap" or P > 1:
        d1 = env.timed(lambda k: run_steps(k, 1), ke, 2, "single")
        detail["single"] = {"pipelines": 1, "steps": ke, "latency_ms_single": round(d1 / ke * 1e3, 4), "gcells_s": round(world * cells * ke / d1 / 1e9, 3), "scaling": "weak",
                            "note": "one heightmap in flight per GPU: noise and erosion of a map do not overlap with another map's"}
    if args.workload != "strips":
        fn, rows = strips_steps_fn(env)
        ds = env.timed(fn, ke, 2, "strips")
        coll = ("all_reduce(min) of one float per step over " + env.backend_name + (" (one-rank group)" if world == 1 else "")) if env.have_group else "none (no process group)"
        detail["strips"] = {"steps": ke, "ms_per_step": round(ds / ke * 1e3, 4), "gcells_s": round(cells * ke / ds / 1e9, 3), "scaling": "strong", "rows_per_rank": rows, "collective": coll,
                            "erosion": "excluded: one shared grid in serial droplet order does not shard (replicas only)"}
    if args.workload != "tiles":
        dt0 = env.timed(tiles_steps_fn(env, 0), ke, 2, "tiles_0")
        kt = max(2, min(K, 3))
        dt1 = env.timed(tiles_steps_fn(env, 1000), kt, 1, "tiles_1000")
        ntile = len(env.all_tiles)
        tc = ntile * 130 * 130
        detail["tiles"] = {"tiles": ntile, "tiles_per_rank": len(env.my_tiles), "scaling": "strong", "collective": "none",
                           "erosion_0": {"steps": ke, "ms_per_batch": round(dt0 / ke * 1e3, 4), "gcells_s": round(tc * ke / dt0 / 1e9, 3), "mtiles_s": round(ntile * ke / dt0 / 1e6, 3)},
                           "erosion_1000": {"steps": kt, "ms_per_batch": round(dt1 / kt * 1e3, 3), "gcells_s": round(tc * kt / dt1 / 1e9, 4), "ktiles_s": round(ntile * kt / dt1 / 1e3, 2)}}
    # BASELINE config 5: ONE 512^3 voxel field (voxel_manager::create_procedural, sine mode) as y slabs, no collective
    VN = 512
    v0, v1 = env.dmod.strip_rows(VN, env.rank, world)
    buf = {}

    def voxel_steps(k):
        if v1 <= v0:
            return
        if not buf:
            buf["v"] = env.torch.empty((v1 - v0) * VN * VN, dtype=env.torch.float32, device=env.dev)
        for _ in range(k):
            env.t.voxel_fill_slab_dev(buf["v"].data_ptr(), VN, VN, VN, (-1.0, -1.0, -0.25), (2.0 / VN, 2.0 / VN, 0.5 / VN), (0.0, 0.0, 0.0), 1.0, 1.0, 123, 456, 0, 0.0, 1, v0, v1 - v0)
        env.t.synchronize()
    dv = env.timed(voxel_steps, ke, 2, "voxels")
    detail["voxels"] = {"grid": f"{VN}^3", "steps": ke, "ms_per_field": round(dv / ke * 1e3, 4), "gvoxels_s": round(VN ** 3 * ke / dv / 1e9, 2), "scaling": "strong", "y_rows_per_rank": v1 - v0, "collective": "none"}
    # the same entry point's lattice generators (glm simplex / Perlin fBm, src/voxels.cpp:328-338) at the reference's own field size (scene_config/config_voxel_params.txt:1-3), rank 0 only
    if env.rank == 0:
        nzr = 64
        vb = env.torch.empty(VN * VN * nzr, dtype=env.torch.float32, device=env.dev)
        fb = {}
        for name, gm in (("sines", 0), ("simplex", 1), ("perlin", 2)):
            call = lambda gm=gm: env.t.voxel_fill_dev(vb.data_ptr(), VN, VN, nzr, (-1.0, -1.0, -0.25), (2.0 / VN, 2.0 / VN, 0.5 / nzr), (0.0, 0.0, 0.0), 1.0, 1.0, 123, 456, gm, 0.0, 1)  # noqa: E731
            for _ in range(3):
                call()
            env.t.synchronize(); env.t.timer_start()
            for _ in range(8):
                call()
            ms = env.t.timer_stop() / 8
            fb[name] = {"ms_per_field": round(ms, 4), "gvoxels_s": round(VN * VN * nzr / ms / 1e6, 1)}
        detail["voxels_512x512x64"] = fb
        del vb


def modes_and_dense(env, detail, ms_gen, ms_ero):
    """rank 0: the same 16384^2 step in the other noise modes (BASELINE config 2 names Perlin + domain warp); dense whole-map erosion (config_heightmap.txt:78, BASELINE config 3)"""
    pkg, t, z, st, N, cells, args, torch = env.pkg, env.t, env.z, env.st, env.N, env.cells, env.args, env.torch
    x0, y0 = env.x0, env.y0
    md = {}
    for name, m in env.MODES.items():
        if m == env.mode:
            md[name] = {"ms_noise": round(ms_gen, 4), "ms_erosion": round(ms_ero, 4), "gcells_s": round(cells / (ms_gen + ms_ero) / 1e6, 3), "gcells_s_noise_only": round(cells / ms_gen / 1e6, 3)}
            continue
        t.init_scene(pkg.make_config(mesh_gen_mode=m, mesh_freq_filter=9 - args.octaves))
        mnm, _ = t.gen_grid_minmax_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < 0.5 * args.clock_warmup_ms:  # init_scene above left the chip idle for a few ms
            t.gen_grid_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        rr = 4
        t.timer_start()
        for _ in range(rr):
            mnm, _ = t.gen_grid_minmax_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        msn = t.timer_stop() / rr
        t.timer_start()
        t.apply_erosion_dev(z.data_ptr(), N, N, mnm, args.droplets, pkg.ERODE_MINZ_IS_MIN)
        mse = t.timer_stop()
        fl = flops_per_cell(m, args.octaves)
        md[name] = {"ms_noise": round(msn, 4), "ms_erosion": round(mse, 4), "gcells_s": round(cells / (msn + mse) / 1e6, 3), "gcells_s_noise_only": round(cells / msn / 1e6, 3),
                    "tflops_8d": round(fl * cells / (msn * 1e-3) / 1e12, 2), "frac_fp32_peak": round(fl * cells / (msn * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}
    t.init_scene(pkg.make_config(mesh_gen_mode=env.mode, mesh_freq_filter=9 - args.octaves))
    detail["modes"] = md
    de = {}
    for nn, dd in ((N, 1000000), (4096, 1000000), (4096, 100000)):
        zz = z[:nn * nn] if nn * nn <= cells else torch.empty(nn * nn, dtype=torch.float32, device=env.dev)  # (a bench grid smaller than config 3's 4096^2 map)
        for _pass in range(2):  # the first run of a shape allocates the scheduler's buffers (GBs for the 16384^2 ring): time the second
            mnd, _ = t.gen_grid_minmax_dev(zz.data_ptr(), -nn / 2, -nn / 2, st.DX_VAL, st.DY_VAL, nn, nn, pkg.GEN_GLACIATE)
            t.synchronize()
            t0 = time.perf_counter()
            t.apply_erosion_dev(zz.data_ptr(), nn, nn, mnd, dd, pkg.ERODE_MINZ_IS_MIN)
            t.synchronize()
            sec = time.perf_counter() - t0
        de[f"{nn}x{nn}_{dd}_droplets"] = {"ms": round(sec * 1e3, 2), "mdroplets_s": round(dd / sec / 1e6, 3), "rounds": t.erosion_report().rounds}
    detail["dense_erosion"] = de


def fused_modes(env, detail):
    """rank 0: the TOLERANCE modes beside the bit-exact default -- TERRA_GEN_FUSED (one rounding per multiply-add: the sine sums on the f32 matrix
    pipe, bit-equal to the restated mode; simplex / Perlin with contraction allowed) and TERRA_GEN_FAST (the sine sums on the half-precision matrix pipe with split operands),
    both within 1e-5 * zmax_est of the reference.  The 16384^2 noise-only grid of the headline, the noise-only fBm grids, the 512^3 and 512 x 512 x 64 voxel fields.
    `parity` names the -m gpu tests that hold every number's configuration to its bar; `eroded_cells_beyond_tol` is why the headline `value` stays on the exact path."""
    import os
    pkg, t, z, st, N, cells, args, torch = env.pkg, env.t, env.z, env.st, env.N, env.cells, env.args, env.torch
    x0, y0 = env.x0, env.y0
    out = {}

    def time_grid(flags, reps=6):
        t.gen_grid_minmax_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, flags)
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < 0.5 * args.clock_warmup_ms:
            t.gen_grid_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, flags)
        t.timer_start()
        for _ in range(reps):
            t.gen_grid_minmax_dev(z.data_ptr(), x0, y0, st.DX_VAL, st.DY_VAL, N, N, flags)
        return t.timer_stop() / reps

    fl = flops_per_cell(env.mode, args.octaves)
    sine = {}
    for name, fb in (() if env.mode != 0 else (("exact", 0), ("fused", pkg.GEN_FUSED), ("fast", pkg.GEN_FAST))):
        ms = time_grid(pkg.GEN_GLACIATE | fb)
        tf, wb = fl * cells / (ms * 1e-3) / 1e12, cells * 4 / (ms * 1e-3) / 1e9
        sine[name] = {"ms_noise_call": round(ms, 4), "gcells_s_noise_only": round(cells / ms / 1e6, 2), "tflops_8d": round(tf, 2),
                      "frac_fp32_peak": round(tf / FP32_PEAK_TFLOPS, 4), "write_tb_s": round(wb / 1e3, 3),
                      # exact: separate multiplies and adds on the vector ALU; fused: the f32 matrix pipe (same 157.3 TFLOP/s datapath); fast: the half-precision matrix pipe has
                      # 16x that, the kernel is bound by writing its 4 B per cell (k_sine_grid_h3)
                      "roofline": ({"bound": "hbm", "achieved": round(wb, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(wb / 8000.0, 4)} if name == "fast" else
                                   {"bound": "mfma" if name == "fused" else "valu", "achieved": round(tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 4)})}
    sine["note"] = ("ms_noise_call = table launches + grid kernel + fused min / max, HIP events; fast: bound by writing the grid (write_tb_s of the ~8 TB/s HBM peak), "
                    "its effective flop rate is above the fp32 vector peak because the work runs on the half-precision matrix pipe")
    out["sine_16384"] = sine
    fb_modes = {}
    for name, m in env.MODES.items():
        if m in (0, 4):
            continue  # (the domain warp has no fused kernel: beyond the bar, see terra_driver.hpp fused_kernel_exists)
        t.init_scene(pkg.make_config(mesh_gen_mode=m, mesh_freq_filter=9 - args.octaves))
        e, f = time_grid(pkg.GEN_GLACIATE, 3), time_grid(pkg.GEN_GLACIATE | pkg.GEN_FUSED, 3)
        fb_modes[name] = {"ms_exact": round(e, 4), "ms_fused": round(f, 4), "speedup": round(e / f, 3), "gcells_s_fused": round(cells / f / 1e6, 2)}
    t.init_scene(pkg.make_config(mesh_gen_mode=env.mode, mesh_freq_filter=9 - args.octaves))
    out["fbm_16384"] = fb_modes
    vox = {}
    VN = 512
    for nz in (512, 64):
        buf = torch.empty(VN * VN * nz, dtype=torch.float32, device=env.dev)
        row = {}
        for name, lvl in (("exact", "0"), ("fused", "1"), ("fast", "2")):
            t.set_option("gen.fused", lvl)
            try:
                call = lambda: t.voxel_fill_dev(buf.data_ptr(), VN, VN, nz, (-1.0, -1.0, -0.25), (2.0 / VN, 2.0 / VN, 0.5 / nz), (0.0, 0.0, 0.0), 1.0, 1.0, 123, 456, 0, 0.0, 1)  # noqa: E731
                for _ in range(4):
                    call()
                t.timer_start()
                for _ in range(10):
                    call()
                ms = t.timer_stop() / 10
            finally:
                t.set_option("gen.fused", "0")
            row[name] = {"ms_per_field": round(ms, 4), "gvoxels_s": round(VN * VN * nz / ms / 1e6, 1)}
        vox[f"{VN}x{VN}x{nz}"] = row
        del buf
    out["voxels"] = vox
    beyond = None
    try:
        beyond = int(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_fused_eroded_beyond.txt")).read().split()[0])
    except (OSError, ValueError, IndexError):
        pass
    out["eroded_cells_beyond_tol"] = {"count": beyond, "of": 16384 * 16384, "droplets": 1000, "source": "tests/test_gpu_fused.py::test_fused_headline_grid_16384_every_cell (recorded: profiles/r06_fused_eroded_beyond.txt)",
                                      "consequence": "the droplet paths amplify last-bit differences (the count is not 0): the headline value stays on the bit-exact path"}
    out["parity"] = "tests/test_gpu_fused.py: fused = bit-equal to the restated mode AND <= 1e-5*zmax_est of the reference (whole 16384^2 grid, 4096 tiles, 512^3 / 512x512x64 fields); fast and fBm-fused = the tolerance bar only"
    detail["fused"] = out


def end_to_end(env, detail):
    """SURVEY 8(d) "(ii) end-to-end incl. D2H z": the z grid of every heightmap delivered to HOST memory.  (a) the library: noise + min + erosion on the device, map i
    on the PCIe link (terra_download_async: bands on four streams, csrc/terra_xfer.hpp) while map i + 1 is computed -- into a pinned array (terra_host_alloc) and into an ordinary
    one; the link's own speed (one hipMemcpyAsync into pinned memory) beside it.  (b) the reference's OWN caller, heightmap_t::proc_gen (src/heightmap.cpp:130-151), from the
    patched engine build (oracle/_ref/libengine_hip.so: build_arrays and apply_erosion go through include/terra.h, everything else -- the eval_index loop, from_floats -- is the
    reference's CPU code) against the unpatched reference (oracle/_ref/liboracle_ref.so), same process, all host cores.  Rank 0, N = 1."""
    import numpy as np
    pkg, t, st, N, torch, dev, args = env.pkg, env.t, env.st, env.N, env.torch, env.dev, env.args
    nbytes = N * N * 4
    out = {}
    g = env.z
    hp = torch.empty(N * N, dtype=torch.float32, pin_memory=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(dev); t0 = time.perf_counter(); hp.copy_(g, non_blocking=True); torch.cuda.synchronize(dev); best = min(best, time.perf_counter() - t0)
    out["hipMemcpy_pinned_d2h_gbs"] = round(nbytes / best / 1e9, 2)
    del hp
    zs = [env.zs[0], env.zs[1 % len(env.zs)]] if len(env.zs) > 1 else [env.zs[0], torch.empty(N * N, dtype=torch.float32, device=dev)]
    K = max(4, min(args.steps, 8))

    def run(dsts, k):
        for i in range(k):
            s = i & 1
            mn, _ = t.gen_grid_minmax_dev(zs[s].data_ptr(), env.x0, env.y0, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
            t.apply_erosion_dev(zs[s].data_ptr(), N, N, mn, args.droplets, pkg.ERODE_MINZ_IS_MIN)
            t.download_wait()                      # map i - 1 has landed: its copy ran beside this map's kernels
            t.download_async(zs[s].data_ptr(), dsts[s])
        t.download_wait()
    pins = [t.pinned((N, N)) for _ in range(2)]
    pags = [np.zeros((N, N), np.float32) for _ in range(2)]
    for name, dsts in (("pinned", [p.array for p in pins]), ("pageable", pags)):
        run(dsts, 2)
        t0 = time.perf_counter(); run(dsts, K); dt = time.perf_counter() - t0
        out[name] = {"steps": K, "ms_per_map": round(dt / K * 1e3, 3), "gcells_s": round(N * N * K / dt / 1e9, 3), "link_gbs": round(nbytes * K / dt / 1e9, 2),
                     "frac_of_hipMemcpy_pinned": round(nbytes * K / dt / 1e9 / out["hipMemcpy_pinned_d2h_gbs"], 3)}
    same = bool((pins[(K - 1) & 1].array == pags[(K - 1) & 1]).all())
    out["pinned_equals_pageable"] = same
    for p in pins:
        p.free()
    del pags
    out["bound"] = "PCIe: 4 B per cell over the host link; the device side of a map (noise + erosion, ~1.5 ms at 16384^2) hides under the previous map's copy (~19 ms)"
    # (b) engine in the loop
    try:
        import orclib
        lib = orclib.engine_lib("hip")
        if lib is None or not orclib.ref_available():
            out["engine_proc_gen"] = {"unavailable": "oracle/_ref/libengine_hip.so / liboracle_ref.so did not travel with the repo"}
        else:
            eng, ref = orclib.Checker("ref", lib), orclib.Checker("ref")
            cfg = orclib.make_config(mesh_gen_mode=env.mode, mesh_freq_filter=9 - args.octaves)
            res = {"cores": ref.num_threads(), "droplets": args.droplets}
            for n in (4096, N):
                row = {}
                for name, ck, hip in (("reference_cpu", ref, -1), ("engine_hip_piecewise", eng, 0), ("engine_hip_whole", eng, 1)):
                    if hip >= 0:
                        ck.set_use_hip_terrain(1)
                        ck.set_use_hip_proc_gen(hip)
                    ck.init(cfg)
                    if hip >= 0:
                        ck.heightmap_proc_gen(256, 256, 10)  # first use: library context, scratch
                    best = 1e9
                    for _rep in range(2 if (hip == 1 or n <= 4096) else 1):  # (the fast path twice: its first map of a size allocates device scratch)
                        t0 = time.perf_counter()
                        pix, _, _ = ck.heightmap_proc_gen(n, n, args.droplets)
                        best = min(best, time.perf_counter() - t0)
                        del pix
                    row[name] = {"s": round(best, 4), "gcells_s": round(n * n / best / 1e9, 4)}
                row["speedup_piecewise"] = round(row["reference_cpu"]["s"] / row["engine_hip_piecewise"]["s"], 2)
                row["speedup_whole"] = round(row["reference_cpu"]["s"] / row["engine_hip_whole"]["s"], 2)
                res[f"{n}x{n}"] = row
            res["note"] = ("heightmap_t::proc_gen as the engine calls it (the harness's copy of the pixels included): reference_cpu = the unpatched build, all host cores; "
                           "engine_hip_piecewise = INTEGRATION.md sections 2 + 3 only: build_arrays and apply_erosion on the GPU, the grid crosses PCIe three times and the eval_index "
                           "copy loop, the z range and from_floats stay the reference's CPU loops; engine_hip_whole = section 3e: the body as one call, 2 bytes per cell cross the link")
            out["engine_proc_gen"] = res
    except Exception as e:  # noqa: BLE001 -- reported, never fatal for the line
        out["engine_proc_gen"] = {"error": repr(e)[:300]}
    detail["end_to_end"] = out


def onegrid_rank_floor(env, detail, sim_world):
    """What ONE rank of the one-grid line does per step at world size `sim_world`, measured on this GPU: its 1/sim_world row strip of noise (table launch + grid kernel), the
    all_reduce(min) of one float over the (one-rank) group and the host read-back -- and, every sim_world-th step, a whole-grid erosion on an eroder context beside it.  The
    slowest of the two is the floor of the N = sim_world step: the predicted one-grid value at that world size is cells / floor (xGMI window traffic of remote droplets not
    included: a few thousand 4 KiB loads per erosion)."""
    pkg, N, st, torch, dist = env.pkg, env.N, env.st, env.torch, env.dist
    t = env.t
    rows = -(-N // sim_world)
    z = env.z
    K = max(16, env.args.steps)
    red = torch.zeros(1, dtype=torch.float32, device=env.coll_dev)

    def noise_steps(k, erode_every=0, ectx=None):
        import threading
        th = None
        for s in range(k):
            mn, _ = t.gen_grid_rows_minmax_dev(z.data_ptr(), -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 0, rows, pkg.GEN_GLACIATE)
            if env.have_group:
                red[0] = mn
                dist.all_reduce(red, op=dist.ReduceOp.MIN)
                mn = float(red.item())
            if erode_every and s % erode_every == 0:
                if th is not None:
                    th.join()
                th = threading.Thread(target=lambda m=env.full_min: ectx.apply_erosion_dev(env.ez.data_ptr(), N, N, m, env.args.droplets, pkg.ERODE_MINZ_IS_MIN))
                th.start()
        if th is not None:
            th.join()
    t.synchronize()
    noise_steps(8)
    t0 = time.perf_counter(); noise_steps(K); t.synchronize(); d_noise = (time.perf_counter() - t0) / K
    t.timer_start()
    for _ in range(K):
        t.gen_grid_rows_minmax_dev(z.data_ptr(), -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 0, rows, pkg.GEN_GLACIATE)
    ms_dev = t.timer_stop() / K
    own = len(env.ctxs) < 2  # (one context = one thread at a time: the eroding thread needs a context that is not the noise loop's)
    ectx = env.ctxs[-1]
    if own:
        ectx = pkg.Terra(env.local_rank); ectx.init_scene(pkg.make_config(mesh_gen_mode=env.mode, mesh_freq_filter=9 - env.args.octaves))
    env.ez = env.zs[-1] if len(env.zs) > 1 else torch.empty(N * N, dtype=torch.float32, device=env.dev)
    env.full_min, _ = ectx.gen_grid_minmax_dev(env.ez.data_ptr(), -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)  # a real heightmap for the erosions
    noise_steps(sim_world, sim_world, ectx)
    t0 = time.perf_counter(); noise_steps(K, sim_world, ectx); t.synchronize(); ectx.synchronize(); d_both = (time.perf_counter() - t0) / K
    # the same step as OneHeightmapPipeline enqueues it under RCCL (dist.py::_run_device_paced, tools/bench_native_onegrid.c): nothing read back -- the strip's min stays in
    # HBM, the all_reduce works on it on the noise stream, the eroding context waits for an event behind it
    d_dev = d_dev_both = None
    if str(env.coll_dev).startswith("cuda"):
        stream = torch.cuda.Stream(device=env.dev)
        mm = torch.zeros(2, dtype=torch.float32, device=env.dev)
        ev = t.event_create()

        def dev_steps(k, erode_every=0):
            import threading
            th = None
            with torch.cuda.stream(stream):
                for s in range(k):
                    t.gen_grid_rows_minmax_async_dev(z.data_ptr(), -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 0, rows, mm.data_ptr(), pkg.GEN_GLACIATE)
                    if env.have_group:
                        dist.all_reduce(mm[0:1], op=dist.ReduceOp.MIN)
                    t.event_record(ev)
                    if erode_every and s % erode_every == 0:
                        if th is not None:
                            th.join()

                        def job(m=env.full_min):
                            ectx.event_wait(ev)
                            ectx.apply_erosion_dev(env.ez.data_ptr(), N, N, m, env.args.droplets, pkg.ERODE_MINZ_IS_MIN)
                        th = threading.Thread(target=job)
                        th.start()
            if th is not None:
                th.join()
        t.synchronize(); t.set_stream(stream.cuda_stream)
        try:
            dev_steps(8)
            t.synchronize()
            t0 = time.perf_counter(); dev_steps(K); t.synchronize(); d_dev = (time.perf_counter() - t0) / K
            dev_steps(sim_world, sim_world); t.synchronize(); ectx.synchronize()
            t0 = time.perf_counter(); dev_steps(K, sim_world); t.synchronize(); ectx.synchronize(); d_dev_both = (time.perf_counter() - t0) / K
            # the same with the sparse scheduler's read-only phases sharded by strip owner (terra_erosion_shard_*, dist.py shard_traces): EVERY step this rank probes / traces the
            # droplets that start in its strip on a tracer context (own stream, behind the step's all_reduce), and every sim_world-th step its eroder gathers the traces of
            # all sim_world arenas and checks / commits (the other ranks' arenas: traced once, up front -- what the gather would find there)
            try:
                D = env.args.droplets
                tctx = pkg.Terra(env.local_rank); tctx.init_scene(pkg.make_config(mesh_gen_mode=env.mode, mesh_freq_filter=9 - env.args.octaves))
                stride = -(-tctx.erosion_shard_arena_bytes(D) // 4096) * 4096
                arena = torch.empty(stride * sim_world, dtype=torch.uint8, device=env.dev)
                row_end = [min((r + 1) * rows, N) for r in range(sim_world)]
                for r in range(sim_world):
                    r0 = min(r * rows, N)
                    tctx.erosion_shard_trace_dev(env.ez.data_ptr(), N, N, D, r0, row_end[r] - r0, arena.data_ptr() + r * stride)
                tctx.synchronize()
                NT = int(__import__("os").environ.get("TERRA_BENCH_TRACERS", "3"))  # tracer contexts: consecutive steps' traces overlap (each is a latency chain longer than a strip's noise), as in OneHeightmapPipeline
                tcs = [tctx] + [pkg.Terra(env.local_rank) for _ in range(NT - 1)]
                for c in tcs[1:]:
                    c.init_scene(pkg.make_config(mesh_gen_mode=env.mode, mesh_freq_filter=9 - env.args.octaves))
                streams2 = [torch.cuda.Stream(device=env.dev) for _ in tcs]
                for c, s2 in zip(tcs, streams2):
                    c.set_stream(s2.cuda_stream)
                ev2 = tctx.event_create()
                mmz = torch.tensor([env.full_min, 0.0], dtype=torch.float32, device=env.dev)
                flag = torch.zeros(1, dtype=torch.float32, device=env.dev)
                pg2 = None
                if env.have_group:  # "all traces made" in a communicator of its own (as in OneHeightmapPipeline): in the step's group it would queue in front of the next step's all_reduce(min)
                    import os
                    import sys
                    sys.stdout.flush()
                    saved = os.dup(1)
                    os.dup2(2, 1)  # (a new communicator prints a banner on the C stdout)
                    try:
                        pg2 = dist.new_group()
                        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg2)
                        torch.cuda.synchronize(env.dev)
                    finally:
                        import ctypes
                        try:
                            ctypes.CDLL(None).fflush(None)
                        finally:
                            os.dup2(saved, 1)
                            os.close(saved)
                torch.cuda.current_stream(env.dev).synchronize()

                host_parts = {}

                def shard_steps(k, erode_every=0):
                    import threading
                    th = None
                    pc_ = time.perf_counter
                    for s in range(k):
                        a0 = pc_()
                        with torch.cuda.stream(stream):
                            t.gen_grid_rows_minmax_async_dev(z.data_ptr(), -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 0, rows, mm.data_ptr(), pkg.GEN_GLACIATE)
                            a1 = pc_()
                            if env.have_group:
                                dist.all_reduce(mm[0:1], op=dist.ReduceOp.MIN)
                            a2 = pc_()
                            t.event_record(ev)
                        a3 = pc_()
                        tc = tcs[s % NT]
                        with torch.cuda.stream(streams2[s % NT]):
                            tc.event_wait(ev)
                            a4 = pc_()
                            tc.erosion_shard_trace_dev(env.ez.data_ptr(), N, N, D, 0, rows, arena.data_ptr())
                            a5 = pc_()
                            if env.have_group:
                                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg2)  # "all traces made"
                            a6 = pc_()
                            tc.event_record(ev2)
                        a7 = pc_()
                        for nm, dv in (("noise", a1 - a0), ("allreduce_min", a2 - a1), ("event_record", a3 - a2), ("event_wait", a4 - a3), ("trace_call", a5 - a4), ("allreduce_flag", a6 - a5), ("event_record2", a7 - a6)):
                            host_parts[nm] = host_parts.get(nm, 0.0) + dv
                        if erode_every and s % erode_every == 0:
                            if th is not None:
                                th.join()

                            def job():
                                ectx.event_wait(ev2)
                                ectx.erosion_shard_finish_dev(env.ez.data_ptr(), N, N, mmz.data_ptr(), D, pkg.ERODE_MINZ_IS_MIN, sim_world, 0, row_end, arena.data_ptr(), stride)
                            th = threading.Thread(target=job)
                            th.start()
                    if th is not None:
                        th.join()
                try:
                    def sync_all():
                        t.synchronize(); ectx.synchronize()
                        for c in tcs:
                            c.synchronize()
                    shard_steps(sim_world, sim_world); sync_all()
                    host_parts.clear()
                    t0 = time.perf_counter(); shard_steps(K, sim_world); d_shard_host = (time.perf_counter() - t0) / K; sync_all(); d_shard = (time.perf_counter() - t0) / K
                    tctx.timer_start()
                    for _ in range(8):
                        tctx.erosion_shard_trace_dev(env.ez.data_ptr(), N, N, D, 0, rows, arena.data_ptr())
                    ms_trace = tctx.timer_stop() / 8
                    ectx.synchronize(); ectx.timer_start()
                    for _ in range(4):
                        ectx.erosion_shard_finish_dev(env.ez.data_ptr(), N, N, mmz.data_ptr(), D, pkg.ERODE_MINZ_IS_MIN, sim_world, 0, row_end, arena.data_ptr(), stride)
                    ms_finish = ectx.timer_stop() / 4
                    ectx.timer_start()
                    for _ in range(4):
                        ectx.apply_erosion_dev(env.ez.data_ptr(), N, N, env.full_min, D, pkg.ERODE_MINZ_IS_MIN)
                    ms_whole = ectx.timer_stop() / 4
                    sharded = {"ms_step_enqueue_only_sharded_traces_with_every_%dth_finish" % sim_world: round(d_shard * 1e3, 4), "ms_step_host_enqueue": round(d_shard_host * 1e3, 4), "host_us_per_step_by_call": {k_: round(v_ / K * 1e6, 1) for k_, v_ in host_parts.items()}, "ms_trace_own_strip": round(ms_trace, 4), "tracer_contexts": NT,
                               "ms_finish_gather_check_commit": round(ms_finish, 4), "ms_whole_erosion_one_context": round(ms_whole, 4),
                               "note": "per step: strip noise + all_reduce, this rank's 1/%d of the traces on a tracer context, a second collective; every %dth step the eroder gathers all arenas and "
                                       "commits (instead of tracing all droplets itself, most of them over xGMI)" % (sim_world, sim_world)}
                finally:
                    for c in tcs:
                        c.synchronize(); c.set_stream(None)
                    tctx.event_destroy(ev2)
                    for c in tcs:
                        c.close()
            except Exception as e:  # noqa: BLE001 -- an extra, never the reason a bench line is missing
                sharded = {"error": repr(e)[:300]}
        finally:
            t.synchronize(); t.set_stream(None); t.event_destroy(ev)
    if own:
        ectx.close()
    if d_dev is None:
        sharded = None
    floor = max(d_both, d_noise) if d_dev is None else max(d_dev, d_dev_both)
    detail["onegrid_rank_floor"] = {"simulated_world": sim_world, "rows_per_rank": rows, "steps": K,
                                    "ms_strip_noise_device": round(ms_dev, 4), "ms_step_noise_allreduce_item": round(d_noise * 1e3, 4),
                                    "host_and_collective_share": round(1.0 - ms_dev / (d_noise * 1e3), 3),
                                    "ms_step_with_every_%dth_erosion" % sim_world: round(d_both * 1e3, 4),
                                    "ms_step_enqueue_only": None if d_dev is None else round(d_dev * 1e3, 4),
                                    "ms_step_enqueue_only_with_every_%dth_erosion" % sim_world: None if d_dev_both is None else round(d_dev_both * 1e3, 4),
                                    "sharded_traces": sharded,
                                    "predicted_gcells_s_at_that_world": round(N * N / floor / 1e9, 1),
                                    "predicted_from": "the read-back step (gloo / no device collective)" if d_dev is None else "the enqueue-only step (what the pipeline runs under RCCL)",
                                    "collective": ("all_reduce(min) over " + env.backend_name + " (one-rank group on this box)") if env.have_group else "none (no process group)",
                                    "note": "a prediction from measured parts, not a measurement of N GPUs: remote window traffic over xGMI and the slowest-rank effect are not in it"}
