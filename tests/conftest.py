import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


def _has_gpu():
    try:
        import ctypes
        lib = os.path.join(ROOT, "3dworld_amd", "libterra_hip.so")
        if not os.path.exists(lib):
            return False
        return ctypes.CDLL(lib).terra_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with -m gpu on the GPU box; when someone runs the whole suite on a box without a GPU, skip them loudly
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device / libterra_hip.so (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("3dworld_amd")


@pytest.fixture(scope="session")
def orc():
    import orclib
    orclib.build_oracle()
    return orclib.Checker("orc")


@pytest.fixture(scope="session")
def ref():
    import orclib
    orclib.build_oracle()
    if not orclib.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return orclib.Checker("ref")


@pytest.fixture(scope="session")
def emul_lib():
    """tests/emul/libterra_emul.so: host emulation of the kernel bodies (test infrastructure, see tests/emul/terra_emul.cpp)."""
    if os.environ.get("TERRA_EMUL_LIB"):  # a pre-built variant, e.g. the -fsanitize=address,undefined build (run pytest with LD_PRELOAD=libasan.so:libubsan.so)
        return os.environ["TERRA_EMUL_LIB"]
    src = os.path.join(ROOT, "tests", "emul", "terra_emul.cpp")
    out = os.path.join(ROOT, "tests", "emul", "libterra_emul.so")
    csrc = os.path.join(ROOT, "3dworld_amd", "csrc")
    deps = [src, os.path.join(ROOT, "include", "terra.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    stale = lambda: not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)  # noqa: E731
    if stale():  # several xdist workers may find it stale at once: one builds (to a temporary file, renamed when complete), the others wait for the lock and find it fresh
        import fcntl
        with open(out + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                tmp = f"{out}.{os.getpid()}.tmp"
                subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", tmp, src, "-lz"], check=True)
                os.replace(tmp, out)
    return out


@pytest.fixture()
def emul(pkg, emul_lib):
    t = pkg.Terra(0, emul_lib)
    yield t
    t.close()


@pytest.fixture()
def gpu(pkg):
    """The product: libterra_hip.so on cuda:0 / hip:0.  Fails (does not skip) when the library is missing on a GPU box."""
    t = pkg.Terra(0)
    yield t
    t.close()
