#!/usr/bin/env python3
"""Host <-> device bandwidth of whole grids, and the end-to-end heightmap rate with the z grid delivered to HOST memory (SURVEY 8d "(ii) end-to-end incl. D2H z").

  hipMemcpy pinned   torch: a device tensor copied into a pinned host tensor (one hipMemcpyAsync, HIP events): the reference bandwidth of this box's link
  hipMemcpy pageable torch: the same into an ordinary host tensor (what a plain hipMemcpy to the caller's vector does)
  terra download     terra_download_async + terra_download_wait (csrc/terra_xfer.hpp: 8 MiB bands, 4 streams, pinned staging) into a pinned array / a pageable array
  terra upload       terra_memcpy_h2d from a pageable array (the same engine, reversed)
  end to end         K heightmaps: noise + min + erosion on the device, each map downloaded while the NEXT map's kernels run (two host arrays, alternating)

usage: pcie_probe.py [size=16384] [steps=8]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dworld_amd")


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nbytes = N * N * 4
    dev = torch.device("cuda", 0)
    out = {"grid": N, "bytes": nbytes}
    g = torch.empty(N * N, dtype=torch.float32, device=dev).normal_()
    hp = torch.empty(N * N, dtype=torch.float32, pin_memory=True)
    hq = torch.empty(N * N, dtype=torch.float32); hq.zero_()  # pageable, touched

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best
    out["hipMemcpy_pinned_d2h_gbs"] = round(nbytes / timed(lambda: hp.copy_(g, non_blocking=True)) / 1e9, 2)
    out["hipMemcpy_pinned_h2d_gbs"] = round(nbytes / timed(lambda: g.copy_(hp, non_blocking=True)) / 1e9, 2)
    out["hipMemcpy_pageable_d2h_gbs"] = round(nbytes / timed(lambda: hq.copy_(g)) / 1e9, 2)
    out["hipMemcpy_pageable_h2d_gbs"] = round(nbytes / timed(lambda: g.copy_(hq)) / 1e9, 2)
    a = np.empty(N * N, np.float32); b = np.zeros(N * N, np.float32)
    t0 = time.perf_counter(); a[:] = b; out["host_memcpy_1thread_gbs"] = round(nbytes / (time.perf_counter() - t0) / 1e9, 2)
    del hp, hq

    t = pkg.Terra(0)
    st = t.init_scene(pkg.make_config(mesh_gen_mode=0, mesh_freq_filter=1))
    pin = [t.pinned((N, N)) for _ in range(2)]
    pag = [np.zeros((N, N), np.float32) for _ in range(2)]
    z = [t.alloc(nbytes) for _ in range(2)]
    t.gen_grid_dev(z[0].ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE); t.synchronize()

    def dl(dst):
        t.download_async(z[0].ptr, dst); t.download_wait()
    for name, dst in (("pinned", pin[0].array), ("pageable", pag[0])):
        dl(dst)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); dl(dst); best = min(best, time.perf_counter() - t0)
        out[f"terra_download_{name}_gbs"] = round(nbytes / best / 1e9, 2)
    assert (pin[0].array == pag[0]).all()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); z[1].upload(pag[0]); best = min(best, time.perf_counter() - t0)
    out["terra_upload_pageable_gbs"] = round(nbytes / best / 1e9, 2)

    # ---- end to end: the z grid of every heightmap lands in host memory; map i is on the link while map i + 1 is computed
    def e2e(dsts, overlap):
        def run(k):
            for i in range(k):
                s = i & 1
                mn, _ = t.gen_grid_minmax_dev(z[s].ptr, -N / 2, -N / 2, st.DX_VAL, st.DY_VAL, N, N, pkg.GEN_GLACIATE)
                t.apply_erosion_dev(z[s].ptr, N, N, mn, 1000, pkg.ERODE_MINZ_IS_MIN)
                if overlap:
                    t.download_wait()               # map i - 1 has landed (its copy ran beside this map's kernels)
                    t.download_async(z[s].ptr, dsts[s])
                else:
                    t.download_async(z[s].ptr, dsts[s]); t.download_wait()
            t.download_wait()
        run(2)
        t0 = time.perf_counter(); run(K); dt = time.perf_counter() - t0
        return {"ms_per_map": round(dt / K * 1e3, 3), "gcells_s": round(N * N * K / dt / 1e9, 3), "link_gbs": round(nbytes * K / dt / 1e9, 2)}
    out["end_to_end"] = {"pinned_overlapped": e2e([p.array for p in pin], True), "pinned_serial": e2e([p.array for p in pin], False),
                         "pageable_overlapped": e2e(pag, True), "pageable_serial": e2e(pag, False)}
    for p in pin:
        p.free()
    t.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
