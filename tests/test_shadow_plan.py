"""the host side of the mesh-shadow LDS kernels (no GPU): the closed form of a sweep's first / last zone against the Bresenham walk, and the lane order (tests/shadow_plan_check.cpp)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shadow_zones_and_lane_order(tmp_path):
    exe = str(tmp_path / "shadow_plan_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", exe, os.path.join(ROOT, "tests", "shadow_plan_check.cpp"), "-lz"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok (0 problems)"), r.stdout + r.stderr
