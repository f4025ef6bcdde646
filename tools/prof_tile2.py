"""tools/prof_tile2.py -- the 64 x 64 x 1000 tile batch with one wave per tile (default) and with two (TERRA_TILE_WAVES=2, k_tile_erosion2) on the same box, and the two-wave
kernel's own counters (TERRA_T2_DIAG=1, printed by the library to stderr): steps made as the primary / speculatively, runs put back, droplets done again, waits, spins."""
import importlib, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("3dworld_amd")
    t = pkg.Terra(0)
    t.init_scene(pkg.make_config(mesh_gen_mode=0))
    tiles = [(tx, ty) for ty in range(-32, 32) for tx in range(-32, 32)]
    for rep in range(3):
        t0 = time.perf_counter()
        t.tiles_create_zvals(tiles, 1000, stats=False, normals=False)
        print(f"  {len(tiles)} tiles x 1000 droplets: {(time.perf_counter() - t0)*1e3:.1f} ms (host clock, incl. the download)", flush=True)
    print(f"  tiles redone by the one-wave kernel after a spin time-out: {t.tile_erosion_fallbacks()}")
    import numpy as np
    one = np.array([(24, 28)], np.int32)  # the batch's heaviest tile: all land, 103 756 droplet steps in the serial order
    zt = t.alloc(130 * 130 * 4)
    for rep in range(2):
        t.synchronize(); t0 = time.perf_counter(); t.tiles_create_zvals_dev(one, 0, zt.ptr); t.synchronize(); t1 = time.perf_counter()
        t.tiles_create_zvals_dev(one, 1000, zt.ptr); t.synchronize(); t2 = time.perf_counter()
    print(f"  tile (24, 28) alone, 1000 droplets: {((t2 - t1) - (t1 - t0))*1e3:.1f} ms", flush=True)
    sys.exit(0)
for label, env in (("one wave per tile (default)", {}), ("two waves per tile (TERRA_TILE_WAVES=2)", {"TERRA_TILE_WAVES": "2"}), ("two waves per tile with counters (TERRA_T2_DIAG=1)", {"TERRA_TILE_WAVES": "2", "TERRA_T2_DIAG": "1"})):
    print("==", label, flush=True)
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=e, capture_output=True, text=True, timeout=600)
    print(r.stdout, end="")
    for line in r.stderr.splitlines():
        if "t2" in line.lower() or "tile2" in line.lower() or "primary" in line.lower():
            print("  [library]", line)
