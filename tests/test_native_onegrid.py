"""tools/bench_native_onegrid.c -- ONE heightmap per step on all ranks, one C process per GPU (include/terra.h + rccl.h), the host-language form of
3dworld_amd/dist.py::OneHeightmapPipeline.  CPU: the driver linked against the host-emulation library, 2 and 3 processes, the per-step minimum through its
shared-memory page; GPU: against libterra_hip.so -- one rank with the all-reduce through RCCL on the noise stream, two ranks sharing GPU 0 (RCCL refuses that,
so --coll shm).  `--check` compares the last step's grid byte for byte with the same map made by ONE context alone; the driver exits 4 when they differ."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "bench_native_onegrid.c")
ROCM = "/opt/rocm"


def build(out, libdir, libname):
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(ROCM, "include", "rccl", "rccl.h")):
        pytest.skip("gcc / rccl.h not available")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROCM, "include"), SRC,
           "-L" + libdir, "-l" + libname, "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-lamdhip64", "-lpthread", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def run(exe, *args, timeout=300):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout}\n{r.stderr}"
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def emul_driver(emul_lib, tmp_path_factory):
    return build(str(tmp_path_factory.mktemp("onegrid") / "onegrid_emul"), os.path.dirname(emul_lib), "terra_emul")


@pytest.mark.parametrize("ranks,steps,n,droplets,grids,eroders", [(2, 4, 256, 200, 3, 2), (3, 7, 384, 300, 2, 1)])
def test_c_driver_ranks_on_the_emulator_equal_one_context(emul_driver, ranks, steps, n, droplets, grids, eroders):
    d = run(emul_driver, ranks, steps, n, droplets, "--same-device", "--coll", "shm", "--check", "--grids", grids, "--eroders", eroders, "--warmup", 2)
    assert d["check"] == "bit-equal" and d["ranks"] == ranks and d["steps"] == steps


@pytest.mark.parametrize("ranks,steps,n,droplets,grids,eroders", [(2, 4, 512, 40, 3, 2), (3, 5, 640, 60, 2, 1)])
def test_c_driver_sharded_traces_on_the_emulator_equal_one_context(emul_driver, ranks, steps, n, droplets, grids, eroders):
    """--shard-traces: every rank traces the droplets that start in its rows into its arena (terra_erosion_shard_trace_dev), the step's eroder gathers them through the
    mapping (terra_erosion_shard_finish_dev); few droplets on these maps, so that the sparse scheduler takes the run by its own rule"""
    d = run(emul_driver, ranks, steps, n, droplets, "--same-device", "--coll", "shm", "--check", "--shard-traces", "--grids", grids, "--eroders", eroders, "--warmup", 2)
    assert d["check"] == "bit-equal" and d["ranks"] == ranks and d["shard_traces"] == 1


def test_c_driver_rank_floor_mode_on_the_emulator(emul_driver):
    d = run(emul_driver, 1, 4, 256, 200, "--same-device", "--coll", "shm", "--simulate-world", 4)
    assert d["simulate_world"] == 4 and d["ms_per_step"] > 0


def test_c_driver_refuses_bad_arguments(emul_driver):
    r = subprocess.run([emul_driver, "2", "4", "256", "200", "--simulate-world", "4"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2


def test_c_driver_a_failing_rank_ends_the_job(emul_driver):
    """40 ranks on a 256-row grid: the last ranks own no rows and give up; the launcher takes the others (which would wait for them in a barrier) down with them"""
    r = subprocess.run([emul_driver, "40", "2", "256", "50", "--coll", "shm", "--same-device", "--grids", "2", "--warmup", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no rows for this rank" in r.stderr


@pytest.fixture(scope="module")
def hip_driver(tmp_path_factory):
    return build(os.path.join(ROOT, "tools", "_bin", "bench_native_onegrid"), os.path.join(ROOT, "3dworld_amd"), "terra_hip")


@pytest.mark.gpu
def test_c_driver_one_rank_through_rccl_equals_one_context(hip_driver):
    d = run(hip_driver, 1, 6, 2048, 1000, "--check", "--warmup", 2)
    assert d["coll"] == "rccl" and d["check"] == "bit-equal"


@pytest.mark.gpu
def test_c_driver_sharded_traces_one_rank_rccl_and_two_ranks_on_gpu0(hip_driver):
    d = run(hip_driver, 1, 6, 2048, 300, "--check", "--shard-traces", "--warmup", 2)
    assert d["coll"] == "rccl" and d["check"] == "bit-equal" and d["shard_traces"] == 1
    d = run(hip_driver, 2, 6, 2048, 300, "--same-device", "--check", "--shard-traces", "--warmup", 2, "--grids", 3)
    assert d["coll"] == "shm" and d["check"] == "bit-equal" and d["ranks"] == 2


@pytest.mark.gpu
def test_c_driver_two_ranks_sharing_gpu0_equal_one_context(hip_driver):
    d = run(hip_driver, 2, 6, 2048, 1000, "--same-device", "--check", "--warmup", 2, "--grids", 3)
    assert d["coll"] == "shm" and d["check"] == "bit-equal" and d["ranks"] == 2
