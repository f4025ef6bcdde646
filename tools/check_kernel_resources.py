#!/usr/bin/env python3
"""Compile terra_hip.hip to gfx950 assembly and print / check the register, LDS and scratch use of the kernels whose speed depends on it.
The droplet trace kernel is large (one 64-lane wave per droplet, window + version bookkeeping inlined); small source changes have tipped the register allocator
into spilling inside its step loop (1.3-1.7x slower on the GPU) -- so the build is checked: no scratch in the trace kernel, the sine grid kernel and the fBm kernels.
usage: check_kernel_resources.py [--check]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
MUST_NOT_SPILL = [("speculative_erosion", "wave_scratch_t"),  # k_waves<trace lambda>: the only speculative_erosion kernel that takes the LDS scratch
                  ("k_sine_grid", "Lb0ELb0E"), ("k_noise_grid", ""), ("k_tile_erosion", "")]


def kernels():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "terra.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "3dworld_amd", "csrc"), "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "3dworld_amd", "csrc", "terra_hip.hip")]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s = open(out).read()
    res = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        g = lambda k: int(re.search(k + r" (\S+)", m.group(2)).group(1))
        res.append((m.group(1), g(".amdhsa_next_free_vgpr"), g(".amdhsa_group_segment_fixed_size"), g(".amdhsa_private_segment_fixed_size")))
    return res


def main():
    bad = []
    for name, vgpr, lds, scratch in kernels():
        hot = any(a in name and b in name for a, b in MUST_NOT_SPILL)
        if hot or "--all" in sys.argv:
            print(f"{name[:40]}..{name[-48:]}  vgpr {vgpr}  lds {lds}  scratch {scratch}{'  <-- hot' if hot else ''}")
        if hot and scratch:
            bad.append(name)
    if "--check" in sys.argv and bad:
        raise SystemExit("scratch (register spills) in: " + ", ".join(b[-60:] for b in bad))


if __name__ == "__main__":
    main()
