"""CPU tests of the HOST LOGIC of libterra_hip (scene start-up, kernel sequencing, speculative erosion rounds, tile batching,
C-ABI argument checking) through tests/emul/libterra_emul.so -- the same driver and kernel bodies with a host-loop backend.
This is test infrastructure, not a product path; the product (libterra_hip.so) is exercised by tests/test_gpu_parity.py."""
import numpy as np
import pytest

import parity_cases as pc


@pytest.mark.parametrize("mode", [0, 1, 2, 4])
def test_scene_and_grids_golden(pkg, emul, mode):
    pc.case_scene_and_grids(pkg, emul, mode)


def test_shapes_golden(pkg, emul):
    pc.case_shapes(pkg, emul)


@pytest.mark.parametrize("mode,n,mss,force", [(0, 260, 0, False), (0, 140, 50, False), (1, 140, 0, True), (3, 96, 0, False)])
def test_grid_vs_oracle(pkg, emul, orc, mode, n, mss, force):
    pc.case_grid_vs_oracle(pkg, emul, orc, mode, n, mss, force)


def test_division_free_lattice_helpers(emul_lib):
    """glm::mod(x, 289) on lattice coordinates and h/41 in glm::perlin are IEEE divisions in the reference; terra_noise.hpp computes the same
    bits without dividing (exactness argument in the header): 5.2e6 arguments incl. both sides of the fall-back threshold, inf and NaN."""
    import ctypes
    lib = ctypes.CDLL(emul_lib)
    lib.terra_emul_noise_helper_mismatches.restype = ctypes.c_ulonglong
    assert lib.terra_emul_noise_helper_mismatches() == 0


def test_two_cells_per_lane_noise_equals_one_cell(emul_lib):
    """k_noise_grid evaluates two cells per lane on register pairs (terra_noise.hpp templates over the arithmetic type): same bits as one cell."""
    import ctypes
    lib = ctypes.CDLL(emul_lib)
    lib.terra_emul_noise_x2_mismatches.restype = ctypes.c_ulonglong
    lib.terra_emul_noise_x2_mismatches.argtypes = [ctypes.c_uint, ctypes.c_uint32]
    assert lib.terra_emul_noise_x2_mismatches(300000, 7) == 0


def test_lattice_table_noise_equals_direct_evaluation(emul_lib):
    """the grid kernels read the hashed-lattice-point part of glm's simplex / Perlin from a table (terra_noise.hpp: noise_lut_fill, simplex2_lut,
    perlin2_lut): same bits as the direct evaluation on random positions, next to the mod-289 wrap columns, beyond the 2^22 switch, through fBm / domain warp"""
    import ctypes
    lib = ctypes.CDLL(emul_lib)
    lib.terra_emul_noise_lut_mismatches.restype = ctypes.c_ulonglong
    lib.terra_emul_noise_lut_mismatches.argtypes = [ctypes.c_uint, ctypes.c_uint32]
    assert lib.terra_emul_noise_lut_mismatches(400000, 11) == 0


def test_3d_lattice_table_noise_equals_direct_evaluation(emul_lib):
    """k_voxel_noise reads glm's perlin(vec3) / simplex(vec3) hashes and gradients from a 3-D table (terra_noise.hpp: noise3_lut_fill, perlin3_lut_z2, simplex3_lut), two
    voxels per lane: same bits as the direct evaluation on random positions, next to the mod-289 wrap planes, beyond the 2^22 switch, on every (x, y) lattice column"""
    import ctypes
    lib = ctypes.CDLL(emul_lib)
    lib.terra_emul_noise3_lut_mismatches.restype = ctypes.c_ulonglong
    lib.terra_emul_noise3_lut_mismatches.argtypes = [ctypes.c_uint, ctypes.c_uint32]
    assert lib.terra_emul_noise3_lut_mismatches(300000, 13) == 0


def test_block_records_of_regular_fbm_sums_equal_direct_evaluation(emul_lib):
    """regular fBm sums read their gradient terms from per-block records (one per lattice cell and octave, terra_noise.hpp: noise_blocktab_build / fbm2_bt):
    every cell of random 128 x 16 patches (random origin incl. the mod-289 wrap columns, spacing, octave count, shape) gives the bits of the direct sum"""
    import ctypes
    lib = ctypes.CDLL(emul_lib)
    lib.terra_emul_noise_blocktab_mismatches.restype = ctypes.c_ulonglong
    lib.terra_emul_noise_blocktab_mismatches.argtypes = [ctypes.c_uint, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint)]
    fb = ctypes.c_uint(0)
    assert lib.terra_emul_noise_blocktab_mismatches(400, 5, ctypes.byref(fb)) == 0
    assert fb.value < 400  # most patches fit the records; the coarse ones exercise the fall-back decision


def test_powf_restatement_matches_libm(emul_lib):
    """glaciate's pow(relh, custom_glaciate_exp) is libm powf in the reference; 3dworld_amd/csrc/terra_powf.hpp restates glibc's algorithm so the
    device gets the same bits (ocml powf does not).  4*10^6 arguments here, 5*10^7 when the header was written."""
    import ctypes
    lib = ctypes.CDLL(emul_lib)
    lib.terra_emul_powf_mismatches.restype = ctypes.c_ulonglong
    lib.terra_emul_powf_mismatches.argtypes = [ctypes.c_ulonglong, ctypes.c_uint32]
    assert lib.terra_emul_powf_mismatches(4000000, 2024) == 0


def test_sine_epilogue_variants(pkg, emul, orc):
    pc.case_sine_epilogue_variants(pkg, emul, orc)


def test_erosion_golden(pkg, emul):
    r = pc.case_erosion_golden(pkg, emul)
    assert r.rounds >= 2  # dense case: the fixed point needs re-traces


@pytest.mark.parametrize("n,iters", [(96, 50), (256, 300), (512, 1000), (1024, 400)])
def test_speculative_erosion_equals_serial(pkg, emul, orc, n, iters):
    pc.case_erosion_vs_oracle(pkg, emul, orc, n, iters)


def test_erosion_serial_flag_and_overflow_fallback(pkg, emul, orc):
    pc.case_erosion_vs_oracle(pkg, emul, orc, 128, 120, flags=pkg.ERODE_SERIAL)
    pc.case_erosion_vs_oracle(pkg, emul, orc, 128, 120, flags=pkg.ERODE_SERIAL_WAVE)
    emul.set_erosion_tuning(window=64, block_list_capacity=16)  # tiny block lists + small windows: long droplets overflow and run alone, in order
    r, _ = pc.case_erosion_vs_oracle(pkg, emul, orc, 512, 400)
    assert r.serial_fallbacks >= 1 and r.windows >= 7


@pytest.mark.parametrize("n,iters,window,slice_steps,blk_cap,near", [(128, 2000, 32, 4, 0, 0), (256, 3000, 64, 16, 0, 8), (96, 1500, 16, 1, 0, 0), (256, 1200, 48, 8, 12, 0), (192, 900, 7, 3, 0, 2), (256, 2500, 64, 16, 0, 100000)])
def test_erosion_sliding_ring(pkg, emul, orc, n, iters, window, slice_steps, blk_cap, near):
    r, _ = pc.case_erosion_sliding_ring(pkg, emul, orc, n, iters, window, slice_steps, blk_cap, near=near)
    assert r.rounds > r.windows
    if blk_cap:
        assert r.serial_fallbacks >= 1


@pytest.mark.parametrize("fuse", ["0", "4", "7"])
def test_erosion_round_launch_fusion_knob(pkg, emul, orc, fuse):
    """"ero.fuse": the round's bookkeeping passes folded into fewer launches (a batch of rounds as one graph, marks + unlink + publish in one wave, commit + resume + hand-over
    + end of round in one wave closed by the wave that finishes last) -- every combination gives the serial result (the default, 3, runs everywhere else)"""
    import os
    old = os.environ.get("TERRA_ERO_FUSE")
    os.environ["TERRA_ERO_FUSE"] = fuse
    try:
        r, _ = pc.case_erosion_sliding_ring(pkg, emul, orc, 160, 2500, 48, 5, near=8, ck="2:16")
        assert r.checkpoint_resumes > 0 and r.rounds > r.windows
    finally:
        if old is None:
            os.environ.pop("TERRA_ERO_FUSE", None)
        else:
            os.environ["TERRA_ERO_FUSE"] = old
        emul.apply_env_options()


@pytest.mark.parametrize("ck,near,window,slice_steps", [("1:16", 0, 64, 16), ("2:16", 8, 48, 5), ("4:16", 100000, 64, 16), ("7:3", 0, 32, 9), ("32:0", 4, 64, 16), ("3:16", 2, 200, 64)])
def test_erosion_checkpointed_retraces(pkg, emul, orc, ck, near, window, slice_steps):
    """re-traces that resume from a checkpoint of the droplet's previous trace: dense droplets on a small map (every droplet conflicts with its neighbours in the ring),
    checkpoints every few steps so that the copy-from-the-published-version and the roll-back-in-place paths, the undo log (with one step per checkpoint nearly every
    write-back rewrites a cell) and checkpoint exhaustion all occur -- still the serial result, bit for bit"""
    r, _ = pc.case_erosion_sliding_ring(pkg, emul, orc, 160, 2500, window, slice_steps, near=near, ck=ck)
    if not ck.endswith(":0"):
        assert r.checkpoint_resumes > 0 and r.checkpoint_steps_saved > 0
    else:
        assert r.checkpoint_resumes == 0


def test_grid_degenerate_shapes(pkg, emul, orc):
    pc.case_grid_degenerate_shapes(pkg, emul, orc)


def test_tile_batch_shapes(pkg, emul, orc):
    pc.case_tile_batch_shapes(pkg, emul, orc)


def test_hmap_edits_and_export(pkg, emul, orc, tmp_path):
    pc.case_hmap_edits_and_export(pkg, emul, orc, tmp_path)


def test_tile_weights_texture(pkg, emul, orc):
    pc.case_tile_weights(pkg, emul, orc)


def test_tile_ao_lighting(pkg, emul, orc):
    pc.case_tile_ao(pkg, emul, orc)


def test_tile_mesh_shadows(pkg, emul, orc):
    pc.case_tile_mesh_shadows(pkg, emul, orc)


def test_tile_mesh_shadows_halo_interface(pkg, emul, orc):
    pc.case_tile_mesh_shadows_halo(pkg, emul, orc)


def test_tiles_post_pass_on_adversarial_zvals(pkg, emul, orc):
    pc.case_tiles_post_adversarial(pkg, emul, orc)


def test_tiles_from_heightmap_texture(pkg, emul, orc):
    pc.case_tiles_from_heightmap(pkg, emul, orc)


@pytest.mark.parametrize("n,iters,retraces,flags", [(512, 400, 100000, 0), (384, 300, 100000, 2), (256, 500, 3, 0), (128, 120, 0, 2), (1024, 300, 2, 2), (1024, 60, None, 0)])
def test_sparse_erosion_scheduler(pkg, emul, orc, n, iters, retraces, flags):
    r, _ = pc.case_erosion_sparse(pkg, emul, orc, n, iters, "1", retraces, flags)
    if retraces == 100000:
        assert r.sparse_droplets == iters and r.sparse_retraces > 0 and r.rounds == 1 + r.sparse_retraces  # every conflict resolved by a re-trace on the grid
    elif retraces is not None:
        assert 0 < r.sparse_droplets < iters and r.sparse_retraces <= retraces                              # a committed prefix, the rest by the multi-version scheduler


@pytest.mark.parametrize("n,iters,world,eroder,force,retraces", [
    (1024, 60, 2, 0, None, None),      # the sparse scheduler by its own choice, two strips
    (1024, 60, 3, 1, None, None),      # three strips (one empty), the eroder in the middle
    (512, 400, 3, 2, "1", 100000),     # forced onto a map where droplets of different strips meet: every conflict resolved by a re-trace on the eroding rank
    (256, 500, 2, 1, "1", 3),          # ... and with a small allowance: a committed prefix, then the general scheduler
    (128, 120, 2, 0, "1", 0),          # ... gives up at once
    (96, 200, 2, 1, None, None),       # a dense run: the sparse scheduler is not tried, the trace calls do nothing
    (640, 250, 1, 0, "1", 100000),     # one rank: the calls compose to the ordinary erosion
])
def test_sharded_sparse_erosion(pkg, emul_lib, orc, n, iters, world, eroder, force, retraces):
    """SURVEY 8e rows 2-3 / VERDICT r05 item 6: the sparse scheduler's read-only phases sharded by strip owner (terra_erosion_shard_*), the ranks as contexts of one process"""
    rep = pc.case_erosion_sharded(pkg, lambda: pkg.Terra(0, emul_lib), orc, n, iters, world, eroder, force, retraces)
    if retraces == 100000:
        assert rep.sparse_droplets == iters and rep.sparse_retraces > 0
    elif retraces == 3:
        assert 0 < rep.sparse_droplets < iters
    elif force is None and iters == 200:
        assert rep.sparse_droplets == 0


def test_sharded_erosion_argument_errors(pkg, emul):
    """terra_erosion_shard_*: rows outside the grid, a world beyond 16, self outside the world, a row_end that does not end at ysize, a stride smaller than an arena"""
    import ctypes as C
    emul.init_scene(pkg.make_config(mesh_gen_mode=0))
    n, d = 64, 10
    g = emul.alloc(n * n * 4); a = emul.alloc(emul.erosion_shard_arena_bytes(d) * 2); m = emul.alloc(8)
    with pytest.raises(pkg.TerraError):
        emul.erosion_shard_trace_dev(g.ptr, n, n, d, 60, 10, a.ptr)
    for world, self_rank, ends, stride in [(17, 0, [n] * 17, 1 << 30), (2, 2, [32, n], 1 << 30), (2, 0, [32, n - 1], 1 << 30), (2, 0, [40, 32], 1 << 30), (2, 0, [32, n], 4096)]:
        with pytest.raises(pkg.TerraError):
            emul.erosion_shard_finish_dev(g.ptr, n, n, m.ptr, d, 0, world, self_rank, ends, a.ptr, stride)
    for b in (g, a, m):
        b.free()


def test_sparse_erosion_edge_cases_and_probe_pass(pkg, emul, orc):
    pc.case_erosion_edge_sparse(pkg, emul, orc)


def test_sparse_erosion_is_chosen_by_density_and_can_be_switched_off(pkg, emul, orc):
    r, _ = pc.case_erosion_sparse(pkg, emul, orc, 2048, 40, None)     # 40^2 <= 2 * 257^2 blocks: tried
    assert r.sparse_droplets > 0
    r, _ = pc.case_erosion_sparse(pkg, emul, orc, 2048, 40, "0")      # switched off
    assert r.sparse_droplets == 0
    r, _ = pc.case_erosion_sparse(pkg, emul, orc, 256, 300, None)     # dense: not tried
    assert r.sparse_droplets == 0


def test_download_api_host_logic(pkg, emul, orc):
    pc.case_big_transfers(pkg, emul, orc, sizes=((700, 300), (64, 33)))


def test_erosion_context_reuse(pkg, emul, orc):
    pc.case_erosion_context_reuse(pkg, emul, orc)


def test_erosion_edge_cases(pkg, emul, orc):
    pc.case_erosion_edge(pkg, emul, orc)


@pytest.mark.parametrize("mode,iters", [(0, 0), (0, 120), (1, 60)])
def test_tiles(pkg, emul, orc, mode, iters):
    pc.case_tiles(pkg, emul, orc, mode, iters, tiles=((0, 0), (-3, 7), (5, -2)))


def test_tile_golden(pkg, emul):
    pc.case_tile_golden(pkg, emul)


def test_voxels_random_shapes(pkg, emul, orc):
    pc.case_voxels_random(pkg, emul, orc, 5, 10, big=False)


def test_voxels(pkg, emul, orc):
    pc.case_voxels_golden(pkg, emul)
    pc.case_voxels_vs_oracle(pkg, emul, orc, 0, (33, 17, 20))


def test_proc_gen_and_quantize(pkg, emul, orc):
    pc.case_proc_gen(pkg, emul, orc, 160, 120)
    pc.case_quantize_golden(pkg, emul)


@pytest.mark.parametrize("mode,n", [(0, 150), (1, 90)])
def test_gen_grid_minmax(pkg, emul, orc, mode, n):
    pc.case_gen_grid_minmax(pkg, emul, orc, mode, n)


def test_ground_mesh_and_point_queries(pkg, emul, orc):
    pc.case_ground_mesh_and_point_queries(pkg, emul, orc)


def test_eval_points_all_modes(pkg, emul, orc):
    pc.case_eval_points(pkg, emul, orc, n=200)


def test_mesh_text_file_read_write(pkg, emul, orc, tmp_path):
    pc.case_mesh_text_file(pkg, emul, orc, tmp_path)


@pytest.mark.parametrize("mode,nx,ny,nstrips", [(0, 300, 200, 3), (1, 90, 70, 4), (4, 40, 33, 2)])
def test_grid_row_strips(pkg, emul, orc, mode, nx, ny, nstrips):
    pc.case_grid_row_strips(pkg, emul, orc, mode, nx, ny, nstrips)


@pytest.mark.parametrize("gen_mode,shape,nslabs", [(0, (20, 13, 17), 3), (1, (8, 9, 6), 2)])
def test_voxel_slabs(pkg, emul, orc, gen_mode, shape, nslabs):
    pc.case_voxel_slabs(pkg, emul, orc, gen_mode, shape, nslabs)


def test_generator_protocol(pkg, emul, orc):
    pc.case_generator_protocol(pkg, emul, orc)


def test_inject_engine_state(pkg, emul, orc):
    pc.case_inject_engine_state(pkg, emul, orc)


def test_api_errors(pkg, emul):
    pc.case_api_errors(pkg, emul)


def test_random_configs(pkg, emul, orc):
    pc.case_random_configs(pkg, emul, orc, range(6))


def test_random_heightmap_textures(pkg, emul, orc):
    pc.case_random_heightmap_textures(pkg, emul, orc)


def test_heightmap_postprocess_golden(pkg, emul):
    """rest of row a12: to_floats / from_floats / postprocess_height through the host logic, against the reference's own members (golden)"""
    pc.case_heightmap_postprocess_golden(pkg, emul)


def test_heightmap_postprocess_vs_oracle(pkg, emul, orc):
    import numpy as np
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:70, 0:53]  # odd sizes: the 4-pixel fast path ends in a scalar tail
    hgt = 120 + 60 * np.sin(xx * 0.13) * np.cos(yy * 0.1 + 0.5) + rng.uniform(-3, 3, xx.shape)
    pix8 = np.ascontiguousarray(hgt.astype(np.uint8))
    pix16 = np.ascontiguousarray(np.stack([((hgt % 1.0) * 256).astype(np.uint8), hgt.astype(np.uint8)], axis=-1))
    for pix in (pix8, pix16):
        changed, rep = pc.case_heightmap_postprocess_vs_oracle(pkg, emul, orc, pix, 700)
        assert changed > 0 and rep.droplets == 700


def test_multi_contexts_in_one_process(pkg, emul_lib, orc):
    """terra_multi_*: three contexts driven by three host threads -- tiles, row strips, voxel slabs, the row-pipelined mesh shadows with the edge hand-over,
    one region per context -- the union equals the oracle"""
    pc.case_multi_contexts(pkg, emul_lib, orc, 3)


def test_hot_sqrt_selftest_entry_point(emul):
    """the self test's host plumbing (on the host sqrt_rn IS sqrtf; the device sequence is checked by tests/test_gpu_parity.py::test_hot_sqrt_equals_sqrtf)"""
    assert emul.selftest_hot_sqrt(65521) == 0


def test_mesh_seed_zero_static_generator_continues_emul(pkg, emul):
    """mesh_seed 0 (src/mesh_gen.cpp:213-216,238-239): the host logic of terra_init_scene against a fresh oracle process (the GPU form: tests/test_gpu_timed_sizes.py)"""
    from test_gpu_timed_sizes import check_seed0_sequence
    check_seed0_sequence(pkg, emul)


def test_streamed_pipeline_device_min_and_events_emul(pkg, emul_lib, orc):
    """terra_gen_grid_minmax_async_dev + terra_event_* + terra_apply_erosion_devmin_dev through the C ABI (bench.py's streamed schedule), host logic"""
    pc.case_streamed_pipeline(pkg, lambda: pkg.Terra(0, emul_lib), orc, N=200, maps=5, P=2, droplets=(150, 0, 600))


def test_pipeline_step_with_event_handoff_emul(pkg, emul_lib, orc):
    """3dworld_amd/pipeline.py (what bench.py times): three contexts on three threads, the noise turn handed over by events, min(vals) in device memory; host logic"""
    import importlib
    import threading
    import orclib
    pmod = importlib.import_module("3dworld_amd.pipeline")
    N, P, steps, droplets = 160, 3, 3, 200
    ctxs = [pkg.Terra(0, emul_lib) for _ in range(P)]
    try:
        cfg = pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1)
        st = [c.init_scene(cfg) for c in ctxs][0]
        turns = pmod.NoiseTurns()
        evs = [c.event_create() for c in ctxs]
        z = [[c.alloc(N * N * 4) for _ in range(steps)] for c in ctxs]
        mm = [[c.alloc(8) for _ in range(steps)] for c in ctxs]
        errs = []

        def worker(p):
            try:
                for s in range(steps):
                    pmod.proc_gen_step(pkg, ctxs[p], turns, evs[p], z[p][s].ptr, mm[p][s].ptr, -N / 2 + (p + P * s) * N, -N / 2, st.DX_VAL, st.DY_VAL, N, N, droplets)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=worker, args=(p,)) for p in range(P)]
        [x.start() for x in th]; [x.join() for x in th]
        assert not errs, errs
        orc.init(orclib.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
        for p in range(P):
            for s in range(steps):
                ref = orc.gen_grid(-N / 2 + (p + P * s) * N, -N / 2, st.DX_VAL, st.DY_VAL, N, N, 1)
                mn, mx = mm[p][s].download(np.float32, (2,))
                assert (mn, mx) == (ref.min(), ref.max())
                orc.apply_erosion(ref, float(ref.min()), droplets)
                orclib.assert_bit_equal(ref, z[p][s].download(np.float32, (N, N)), f"pipeline {p} step {s}")
        for c, e in zip(ctxs, evs):
            c.event_destroy(e)
    finally:
        for c in ctxs:
            c.close()


def test_erosion_ring_is_capped_by_free_memory_and_scratch_can_be_released(pkg, emul_lib, orc, monkeypatch):
    """the speculation ring shrinks to what the device has free (TERRA_ERO_MEM_BUDGET pretends a small device) -- more ring generations, the same result; and
    terra_release_scratch between calls changes nothing either"""
    monkeypatch.setenv("TERRA_ERO_MEM_BUDGET", str(96 << 20))
    t = pkg.Terra(0, emul_lib)
    try:
        pc.case_erosion_vs_oracle(pkg, t, orc, 160, 3000)
        rep = t.erosion_report().as_dict()
        assert rep["windows"] >= 8, rep  # 3000 droplets through a ring of at most ~360 slots (96 MiB / 266 KiB)
        t.release_scratch()
        pc.case_erosion_vs_oracle(pkg, t, orc, 160, 700)
        z, st, nm, mnz = t.tiles_create_zvals([(0, 0), (1, -1)], 50)
        t.release_scratch()
        z2, _, _, _ = t.tiles_create_zvals([(0, 0), (1, -1)], 50)
        assert (z.view(np.uint32) == z2.view(np.uint32)).all()
    finally:
        t.close()


def test_sparse_scheduler_hands_over_when_its_scratch_does_not_fit(pkg, emul, orc):
    """the sparse scheduler's scratch (~137 KB per droplet) obeys the same free-memory rule as the ring: forced on with a budget it cannot fit, the run goes to the
    general scheduler -- same result, no sparse droplets in the report"""
    emul.set_option("ero.sparse", "1"); emul.set_option("ero.mem_budget", str(24 << 20))
    try:
        r, _ = pc.case_erosion_vs_oracle(pkg, emul, orc, 200, 400)   # 400 droplets x 137 KB = 55 MB > 24 MB
        assert r.sparse_droplets == 0, r.as_dict()
        emul.set_option("ero.mem_budget", "-1")
        r, _ = pc.case_erosion_vs_oracle(pkg, emul, orc, 200, 400)
        assert r.sparse_droplets > 0, r.as_dict()
    finally:
        emul.set_option("ero.sparse", "auto"); emul.set_option("ero.mem_budget", "-1")


def test_fused_tolerance_mode_host_logic(pkg, emul, orc):
    """TERRA_GEN_FUSED / option "gen.fused" through the emulator (the per-cell form of the mode, sine_cell_fused): the flag plumbing, the `no fused kernel` fall-back,
    the option switch, and both bars -- bit-equal to the restated mode, within 1e-5 * zmax_est of the reference's arithmetic"""
    worst = pc.case_fused_grids(pkg, emul, orc, sizes=((260, 150), (1, 1), (64, 64)))
    assert worst < 2e-6, worst
    pc.case_fused_minmax_and_option(pkg, emul, orc, n=200)
    pc.case_fused_tiles(pkg, emul, orc, tiles=((0, 0), (-3, 7), (5, 5)))
    pc.case_fused_voxels(pkg, emul, orc, shapes=((40, 24, 32), (7, 5, 50), (1, 1, 1)))
    for mode in (1, 2, 4):  # (the emulator's per-cell kernels have one build: the flag must reach them and change nothing)
        assert pc.case_fused_fbm(pkg, emul, orc, mode, 96) == 0.0
    assert pc.case_fused_voxel_fbm(pkg, emul, orc, 1, (12, 10, 16)) == 0.0
    assert pc.case_fast_mode(pkg, emul, orc, sizes=((260, 150), (1, 1)), vox_shapes=((12, 10, 16),)) < 2e-6  # (the emulator answers TERRA_GEN_FAST with the fused form)
