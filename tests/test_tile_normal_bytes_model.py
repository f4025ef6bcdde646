"""k_tile_post's byte test, modelled in numpy (3dworld_amd/csrc/terra_kernels.hpp: tp_word_fast).  The kernel takes a normal's bytes floor(127*(n_i/|n| + 1)) from
t_i = fma(127, n_i*rsq(|n|^2), 127) whenever every t_i is further than TP_EPS = 2^-14 from an integer, and falls back to the reference's statements otherwise.  Its
proof needs |rsq(s)*sqrt(s) - 1| <= 2^-23 (checked on the device over every fp32 input: terra_selftest_hot_sqrt).  Here the claim itself is tested: for random and
adversarial slopes and EVERY reciprocal square root the bound allows (the exact one moved by -2^-23 .. +2^-23, relative), a "sure" t_i truncates to the byte the
reference's statement gives (src/tiled_mesh.cpp:865-880: float division by the rounded magnitude, double-precision byte conversion).  Pure arithmetic: no library involved."""
import numpy as np

F = np.float32
TP_EPS = F(2.0 ** -14)


def reference_bytes(n, s):
    """(unsigned char)(127.0*(norm[i] + 1.0)) with norm = n / sqrtf(s) in float (pointT::get_norm), the conversion in double"""
    with np.errstate(all="ignore"):
        mag = np.sqrt(s)                      # float32 sqrt: correctly rounded
        q = (n / mag).astype(F)               # float32 division: correctly rounded
        return np.floor(127.0 * (q.astype(np.float64) + 1.0))


def fast_t(n, s, rel):
    """t = fma(127, n*r, 127) in float32 with r = (1/sqrt(s))*(1 + rel) rounded to float32: any value the hardware instruction may return within the bound"""
    r = ((1.0 / np.sqrt(s.astype(np.float64))) * (1.0 + rel)).astype(F)
    a = (n * r).astype(F)                                        # one float32 multiply
    return (127.0 * a.astype(np.float64) + 127.0).astype(F)      # 127*a is exact in double (8 + 24 bits), the sum too: one rounding, like the fma


def check(n0, n1, dxy):
    c2 = F(dxy * dxy)
    s = ((n0 * n0).astype(F) + (n1 * n1).astype(F)).astype(F) + c2   # the reference's sum, in its order
    s = s.astype(F)
    nz = np.full_like(n0, dxy)
    ok_rows = np.isfinite(s) & (s > 0)
    bad = 0
    sure_total = 0
    for comp in (n0, n1, nz):
        want = reference_bytes(comp, s)
        for rel in (-2.0 ** -23, -2.0 ** -24, 0.0, 2.0 ** -24, 2.0 ** -23):
            t = fast_t(comp, s, rel)
            with np.errstate(all="ignore"):
                frac = (t - np.floor(t)).astype(F)
                sure = ok_rows & (np.abs(frac - F(0.5)) < (F(0.5) - TP_EPS))   # false for NaN
                got = np.floor(t)
            bad += int((sure & (got != want)).sum())
            sure_total += int(sure.sum())
    return bad, sure_total


def test_a_sure_byte_is_the_references_byte_for_every_rsq_within_the_bound():
    rng = np.random.default_rng(11)
    n = 1_000_000
    dxy = F(0.0009765625 * 1.7)
    bad = sure = 0
    for scale in (1e-6, 1e-4, 1e-3, 1e-2, 1.0, 1e3, 1e12):   # slopes from far below to far above dxdy
        n0 = (rng.standard_normal(n) * scale * float(dxy)).astype(F)
        n1 = (rng.standard_normal(n) * scale * float(dxy)).astype(F)
        b, s_ = check(n0, n1, dxy)
        bad += b; sure += s_
    assert bad == 0
    assert sure > 0.9 * 7 * 15 * n * 0.5   # the short path decides the large majority of the bytes (else the test tests nothing)


def test_components_on_byte_boundaries_are_never_sure_and_wrong():
    """slopes built so that n_x/|n| sits next to k/127 - 1 (where the byte changes): the byte test must either refuse or be right"""
    dxy = F(0.002)
    k = np.arange(1, 254, dtype=np.float64)
    target = k / 127.0 - 1.0                                   # the component value at which byte k begins
    target = target[np.abs(target) < 0.999]
    reps = 4000
    rng = np.random.default_rng(12)
    x = np.repeat(target, reps) + rng.uniform(-3e-7, 3e-7, target.size * reps)   # within a few float ulps of the boundary
    n0 = (x / np.sqrt(1.0 - x * x) * float(dxy)).astype(F)     # n1 = 0: n0/|n| = x up to rounding
    n1 = np.zeros_like(n0)
    bad, _ = check(n0, n1, dxy)
    assert bad == 0


def test_flat_and_special_cells_are_left_to_the_exact_path_or_right():
    dxy = F(0.0017)
    n0 = np.array([0.0, -0.0, 1e-30, np.inf, -np.inf, np.nan, 1e19, 3e38], F)
    n1 = np.array([0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1e19, 3e38], F)
    bad, _ = check(n0, n1, dxy)
    assert bad == 0


def test_the_model_has_teeth(monkeypatch):
    """with a margin of 2^-18 instead of 2^-14 the same check DOES find sure-and-wrong bytes (first at 2^-17): the margin in force is ~8x what the model can break"""
    import sys
    rng = np.random.default_rng(13)
    dxy = F(0.0009765625 * 1.7)
    n0 = (rng.standard_normal(500_000) * 1e-2 * float(dxy)).astype(F)
    n1 = (rng.standard_normal(500_000) * 1e-2 * float(dxy)).astype(F)
    monkeypatch.setattr(sys.modules[__name__], "TP_EPS", F(2.0 ** -18))
    bad, _ = check(n0, n1, dxy)
    assert bad > 0
    monkeypatch.setattr(sys.modules[__name__], "TP_EPS", F(2.0 ** -16))
    bad, _ = check(n0, n1, dxy)
    assert bad == 0
