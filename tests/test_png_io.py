"""Heightmap PNG files (row f4): our from-scratch reader / writer against libpng -- the library the reference itself calls
(src/image_io.cpp:493-605) -- through its simplified API, plus the reference's row-order / byte-order conventions and a round trip
through the tile-from-texture path.  CPU only (host code); skipped when libpng16 is not installed."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PNG_IMAGE_VERSION = 1
PNG_FORMAT_GRAY = 0
PNG_FORMAT_LINEAR_Y = 4  # PNG_FORMAT_FLAG_LINEAR: 16-bit linear grayscale, native-endian uint16


class PngImage(C.Structure):  # png_image (png.h, libpng 1.6)
    _fields_ = [("opaque", C.c_void_p), ("version", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32),
                ("flags", C.c_uint32), ("colormap_entries", C.c_uint32), ("warning_or_error", C.c_uint32), ("message", C.c_char * 64)]


@pytest.fixture(scope="module")
def libpng():
    for name in ("libpng16.so.16", "libpng16.so"):
        try:
            lib = C.CDLL(name)
            break
        except OSError:
            lib = None
    if lib is None:
        pytest.skip("libpng16 not installed")
    lib.png_image_begin_read_from_file.argtypes = [C.POINTER(PngImage), C.c_char_p]
    lib.png_image_finish_read.argtypes = [C.POINTER(PngImage), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.png_image_write_to_file.argtypes = [C.POINTER(PngImage), C.c_char_p, C.c_int, C.c_void_p, C.c_int32, C.c_void_p]
    lib.png_image_free.argtypes = [C.POINTER(PngImage)]
    return lib


def libpng_read(lib, path, sixteen):
    img = PngImage(); img.version = PNG_IMAGE_VERSION
    assert lib.png_image_begin_read_from_file(C.byref(img), str(path).encode()), img.message
    img.format = PNG_FORMAT_LINEAR_Y if sixteen else PNG_FORMAT_GRAY
    out = np.empty((img.height, img.width), np.uint16 if sixteen else np.uint8)
    assert lib.png_image_finish_read(C.byref(img), None, out.ctypes.data, 0, None), img.message
    return out


def libpng_write(lib, path, arr):
    img = PngImage(); img.version = PNG_IMAGE_VERSION
    img.height, img.width = arr.shape
    img.format = PNG_FORMAT_LINEAR_Y if arr.dtype == np.uint16 else PNG_FORMAT_GRAY
    assert lib.png_image_write_to_file(C.byref(img), str(path).encode(), 0, np.ascontiguousarray(arr).ctypes.data, 0, None), img.message


@pytest.fixture(scope="module")
def emul_t(emul_lib):
    pkg = importlib.import_module("3dworld_amd")
    t = pkg.Terra(0, emul_lib)
    yield pkg, t
    t.close()


def test_written_files_decode_identically_with_libpng(libpng, emul_t, tmp_path):
    _, t = emul_t
    rng = np.random.default_rng(3)
    for shape in ((7, 5), (64, 64), (129, 257)):
        v16 = rng.integers(0, 65536, shape, dtype=np.uint16)
        pix = np.stack([(v16 & 255).astype(np.uint8), (v16 >> 8).astype(np.uint8)], axis=-1)  # {fraction, integer}: little-endian uint16 in memory
        t.heightmap_write_png(tmp_path / "a16.png", pix)
        assert (libpng_read(libpng, tmp_path / "a16.png", True) == v16).all()  # row 0 first, big-endian samples in the file
        v8 = rng.integers(0, 256, shape, dtype=np.uint8)
        t.heightmap_write_png(tmp_path / "a8.png", v8)
        assert (libpng_read(libpng, tmp_path / "a8.png", False) == v8).all()


def test_reads_libpng_files_with_the_references_conventions(libpng, emul_t, tmp_path):
    """libpng's encoder uses every filter type; texture_t::load_png flips the rows and turns big-endian samples into {lo, hi} pairs"""
    _, t = emul_t
    rng = np.random.default_rng(4)
    for shape in ((9, 3), (100, 130), (257, 64)):
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        smooth = (np.sin(xx * 0.07) * np.cos(yy * 0.05) * 20000 + 30000 + rng.integers(-40, 40, shape)).astype(np.uint16)  # smooth: the Sub / Up / Average / Paeth filters get chosen
        libpng_write(libpng, tmp_path / "b16.png", smooth)
        got = t.heightmap_read_png(tmp_path / "b16.png", True)
        assert got.shape == shape + (2,)
        v = got[:, :, 0].astype(np.uint16) | (got[:, :, 1].astype(np.uint16) << 8)
        assert (v == smooth[::-1]).all()  # rows[i] = data + (height - i - 1)*scanline_size
        hi = t.heightmap_read_png(tmp_path / "b16.png", False)  # without allow_two_byte_grayscale: png_set_strip_16
        assert hi.shape == shape and (hi == (smooth[::-1] >> 8).astype(np.uint8)).all()
        s8 = (smooth >> 8).astype(np.uint8)
        libpng_write(libpng, tmp_path / "b8.png", s8)
        assert (t.heightmap_read_png(tmp_path / "b8.png", True) == s8[::-1]).all()


def test_round_trip_and_errors(emul_t, tmp_path):
    pkg, t = emul_t
    rng = np.random.default_rng(5)
    pix = rng.integers(0, 256, (33, 17, 2), dtype=np.uint8)
    t.heightmap_write_png(tmp_path / "c.png", pix)
    back = t.heightmap_read_png(tmp_path / "c.png", True)
    assert (back == pix[::-1]).all()  # write does not flip, read does (the reference's own asymmetry)
    (tmp_path / "bad.png").write_bytes(b"not a png at all")
    with pytest.raises(pkg.TerraError):
        t.heightmap_read_png(tmp_path / "bad.png")
    data = bytearray((tmp_path / "c.png").read_bytes()); data[40] ^= 0xFF
    (tmp_path / "crc.png").write_bytes(bytes(data))
    with pytest.raises(pkg.TerraError):
        t.heightmap_read_png(tmp_path / "crc.png")
    with pytest.raises(pkg.TerraError):
        t.heightmap_read_png(tmp_path / "missing.png")
    # a well-formed file (valid CRCs) whose IHDR declares dimensions that would wrap the size computations: must be refused before any allocation
    import struct, zlib
    def chunk(tp, body):
        return struct.pack(">I", len(body)) + tp + body + struct.pack(">I", zlib.crc32(tp + body) & 0xFFFFFFFF)
    for w, h in ((0xFFFFFFFF, 0x80000001), (65537, 1), (1, 0x7FFFFFFF), (0x10000000, 0x10000000)):
        ihdr = struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)
        (tmp_path / "huge.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""))
        with pytest.raises(pkg.TerraError):
            t.heightmap_read_png(tmp_path / "huge.png")


def test_file_parsers_survive_mutations(emul_t, tmp_path):
    """PNG and .mod files with flipped bytes, truncations and inflated counts: the parsers either return an image / a record list or fail with a
    TerraError -- they never crash, hang or allocate by an untrusted count (run in-process: a crash would take the test session down)"""
    import orclib
    pkg, t = emul_t
    rng = np.random.default_rng(11)
    pix = rng.integers(0, 256, (21, 19, 2), dtype=np.uint8)
    t.heightmap_write_png(tmp_path / "ok.png", pix)
    good_png = (tmp_path / "ok.png").read_bytes()
    mods = orclib.make_mods([(3, 4, 100), (7, 1, -30), (0, 0, 5)]); brs = orclib.make_brushes([(1, 2, 3, 400, 4), (-7, 8, 5, -600, 2)])
    t.hmap_write_mod(tmp_path / "ok.mod", mods, brs)
    good_mod = (tmp_path / "ok.mod").read_bytes()
    outcomes = {"png_ok": 0, "png_err": 0, "mod_ok": 0, "mod_err": 0}
    for k in range(400):
        for kind, good, reader in (("png", good_png, lambda f: t.heightmap_read_png(f, True)), ("mod", good_mod, lambda f: t.hmap_read_mod(f))):
            data = bytearray(good)
            op = k % 4
            if op == 0:    # flip a few bytes
                for _ in range(int(rng.integers(1, 4))):
                    data[int(rng.integers(0, len(data)))] ^= int(rng.integers(1, 256))
            elif op == 1:  # truncate
                data = data[:int(rng.integers(0, len(data)))]
            elif op == 2:  # overwrite a 4-byte field with a huge value (chunk length / record count)
                o = int(rng.integers(0, max(1, len(data) - 4))); data[o:o + 4] = (0xFFFFFFF0 + int(rng.integers(0, 15))).to_bytes(4, "big" if kind == "png" else "little")
            else:          # append garbage
                data += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
            f = tmp_path / f"m.{kind}"
            f.write_bytes(bytes(data))
            try:
                reader(f)
                outcomes[kind + "_ok"] += 1
            except pkg.TerraError:
                outcomes[kind + "_err"] += 1
    assert outcomes["png_err"] > 100 and outcomes["mod_err"] > 50, outcomes  # most mutations are detected (CRC / signature / length checks)
