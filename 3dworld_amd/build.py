"""hipcc build recipe for libterra_hip.so (gfx950 only; cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libterra_hip.so")
SOURCES = ["terra_hip.hip", "terra_fz.hip"]  # terra_fz.hip: the contraction-allowed build of the noise kernels (the tolerance mode)
HEADERS = ["terra_multi.hpp", "terra_common.hpp", "terra_sincosf.hpp", "terra_powf.hpp", "terra_png.hpp", "terra_landscape.hpp", "terra_modmap.hpp", "terra_noise.hpp", "terra_erosion.hpp", "terra_driver.hpp", "terra_simple_paths.hpp", "terra_api_impl.hpp", "terra_kernels.hpp"]
# -ffp-contract=off: the reference CPU path has no FMA; parity is bit-exact only without contraction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-unused-result"]


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


HASH_FILE = LIB + ".srchash"


def source_hash():
    """sha256 over every source the library is made of (all of csrc/, include/terra.h) and the compiler flags: what the shipped .so must have been built from"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + [os.path.join(os.path.dirname(HERE), "include", "terra.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def needs_build():
    """the library is missing, or was built from other sources than the tree holds now (keyed on content, not on mtimes: a checkout or a copy resets those)"""
    if not os.path.exists(LIB) or not os.path.exists(HASH_FILE):
        return True
    return open(HASH_FILE).read().strip() != source_hash()


def build_library(force=False, verbose=False):
    """Compile 3dworld_amd/csrc/*.hip -> 3dworld_amd/libterra_hip.so (in tree, so it travels with the repo snapshot)."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB, "-lz"]  # zlib: the PNG heightmap files (terra_png.hpp)
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    if verbose and r.stderr:
        print(r.stderr)
    with open(HASH_FILE, "w") as f:
        f.write(source_hash() + "\n")
    return LIB
