#!/usr/bin/env python3
"""dense whole-map erosion (BASELINE config 3 with config_heightmap.txt's real droplet count): ring size x slice sweep ("W:slice,..."; 0 = the library's default); prints the scheduler's report"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
combos = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[3] if len(sys.argv) > 3 else "2048:1024,8192:64,16384:64,16384:32,16384:128").split(",")]
t = pkg.Terra(0)
st = t.init_scene(pkg.make_config(mesh_gen_mode=0))
z = t.alloc(N * N * 4); zc = t.alloc(N * N * 4)
mn, mx = t.gen_grid_minmax_dev(zc.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
lib = t.lib
for (w, sl) in combos:
    t.set_erosion_tuning(window=w if w else 0xFFFFFFFF, **({'block_list_capacity': int(os.environ['ERO_SWEEP_MAXB'])} if os.environ.get('ERO_SWEEP_MAXB') else {}))
    if sl:  # 0: the library's default
        t.set_erosion_slice_steps(sl)
    for rep in range(2):
        lib.terra_memcpy_h2d  # noqa
        import ctypes
        # device-to-device copy through the library's own stream: re-generate instead (cheap)
        t.gen_grid_dev(z.ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
        t.synchronize()
        t0 = time.perf_counter()
        t.apply_erosion_dev(z.ptr, N, N, mn, D, pkg.ERODE_MINZ_IS_MIN)
        t.synchronize()
        dt = time.perf_counter() - t0
    r = t.erosion_report().as_dict()
    print(f"W {w} slice {sl}: {dt*1e3:.1f} ms rounds {r['rounds']} traces {r['traces']} steps {r['steps']} traced {r['traced_steps']} fallbacks {r['serial_fallbacks']}  us/round {dt*1e6/max(1,r['rounds']):.0f}  shifts {r['window_shifts']}", flush=True)
    tk = 0.01  # us per tick
    print(f"    critical_steps {r['critical_steps']} critical_shifts {r['critical_shifts']} | device us: waves {r['clk_wave']*tk:.0f} init {r['clk_init']*tk:.0f} shifts {r['clk_shift']*tk:.0f} tail {r['clk_tail']*tk:.0f} critical {r['clk_critical']*tk:.0f}"
          f" | per shift {r['clk_shift']*tk/max(1,r['window_shifts']):.2f} us, per step (run - shifts) {(r['clk_wave']-r['clk_init']-r['clk_tail']-r['clk_shift'])*tk/max(1,r['traced_steps']):.3f} us, init/trace {r['clk_init']*tk/max(1,r['traces']):.2f} tail/trace {r['clk_tail']*tk/max(1,r['traces']):.2f}", flush=True)
    ns = max(1, r['window_shifts'])
    print(f"    per window move: write-back {r['clk_shift_flush']*tk/ns:.2f} us, block flags {r['clk_shift_prep']*tk/ns:.2f}, look-ups {r['clk_shift_load']*tk/ns:.2f}, grid loads + fill {(r['clk_shift']-r['clk_shift_flush']-r['clk_shift_prep']-r['clk_shift_load'])*tk/ns:.2f}", flush=True)
    print(f"    longest wave per round, summed: {r['clk_critical']*tk:.0f} us = window moves {r['crit_clk_shift']*tk:.0f} + before/after {r['crit_clk_edge']*tk:.0f} + steps {(r['clk_critical']-r['crit_clk_shift']-r['crit_clk_edge'])*tk:.0f} us for {r['crit_steps_own']} steps; its window moves: write-back {r['crit_clk_flush']*tk:.0f} us, block flags {r['crit_clk_prep']*tk:.0f}, look-ups {r['crit_clk_load']*tk:.0f}, grid loads + fill {(r['crit_clk_shift']-r['crit_clk_flush']-r['crit_clk_load']-r['crit_clk_prep'])*tk:.0f}", flush=True)
