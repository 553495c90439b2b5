"""csrc/ebvo_math.h (the atan2 / sincos / exp both the oracle's portable mode and the HIP kernels use)
against glibc: identical except where glibc itself is not correctly rounded (<= 1 ulp, rare)."""
import numpy as np

from tests import oracle as orc


def test_atan2_agrees_with_libm():
    rng = np.random.default_rng(0)
    n = 400_000
    ang = rng.uniform(-np.pi, np.pi, n)
    y, x = np.sin(ang), np.cos(ang)
    a, b = orc.atan2_v(y, x, orc.PORTABLE), orc.atan2_v(y, x, orc.LIBM)
    diff = a != b
    assert diff.mean() < 2e-3
    assert (np.abs(a - b)[diff] <= np.spacing(np.abs(b[diff]))).all()


def test_atan2_special_values():
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-300, -1e-300, 0.5, 3.0])
    Y, X = [v.ravel().copy() for v in np.meshgrid(sp, sp)]
    a, b = orc.atan2_v(Y, X, orc.PORTABLE), orc.atan2_v(Y, X, orc.LIBM)
    assert (a == b).all()
    assert (np.signbit(a) == np.signbit(b)).all()
    assert np.isnan(orc.atan2_v(np.array([np.nan, 1.0]), np.array([1.0, np.nan]), orc.PORTABLE)).all()


def test_sincos_agrees_with_libm():
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.uniform(-np.pi, np.pi, 300_000), rng.uniform(-50, 50, 100_000),
                        np.array([0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, np.pi / 4, 1e-9, -1e-9])])
    s, c = orc.sincos_v(t, orc.PORTABLE)
    sm, cm = orc.sincos_v(t, orc.LIBM)
    for a, b in ((s, sm), (c, cm)):
        diff = a != b
        assert diff.mean() < 4e-3
        assert (np.abs(a - b)[diff] <= np.spacing(np.abs(b[diff]))).all()


def test_exp_agrees_with_libm_and_mpmath():
    import mpmath as mp
    rng = np.random.default_rng(2)
    x = np.concatenate([-rng.uniform(0, 90, 200_000), rng.uniform(-2, 2, 100_000), -rng.exponential(1e-3, 50_000),
                        np.array([0.0, -0.0, -1e-300, -745.0, -746.0, 709.0, 710.0, -np.inf, np.inf, 1e-17, -0.5])])
    a, b = orc.exp_v(x, orc.PORTABLE), orc.exp_v(x, orc.LIBM)
    diff = a != b
    assert diff.mean() < 2e-3
    assert (np.abs(a - b)[diff] <= np.spacing(np.abs(b[diff]))).all()
    assert np.isnan(orc.exp_v(np.array([np.nan]), orc.PORTABLE)).all()
    # where the two disagree the shared routine is the correctly rounded one
    mp.mp.prec = 200
    for xv, av, bv in list(zip(x[diff], a[diff], b[diff]))[:200]:
        exact = mp.exp(mp.mpf(float(xv)))
        assert abs(mp.mpf(float(av)) - exact) <= abs(mp.mpf(float(bv)) - exact)


def _is_nearest(mp, value, exact):
    """`value` (a double) is the double closest to the arbitrary-precision `exact`"""
    d = abs(mp.mpf(float(value)) - exact)
    lo, hi = np.nextafter(value, -np.inf), np.nextafter(value, np.inf)
    return d <= abs(mp.mpf(float(lo)) - exact) and d <= abs(mp.mpf(float(hi)) - exact)


def test_atan2_sincos_are_correctly_rounded_against_mpmath():
    """An anchor that does not go through csrc/ebvo_math.h: the oracle's portable mode and the kernels share that header,
    so their agreement on theta / sin / cos says nothing about the routine itself.  Here every result is compared with the
    200-bit value of mpmath: the shared routine returns the nearest double on all sampled inputs, including every input
    on which glibc returns the other neighbour."""
    import mpmath as mp
    mp.mp.prec = 200
    rng = np.random.default_rng(3)
    # atan2 as the detector calls it: a unit vector (cpu_toed.cpp:226-229), plus general magnitudes
    ang = rng.uniform(-np.pi, np.pi, 6000)
    y = np.concatenate([np.sin(ang), rng.normal(0, 50, 2000), np.array([1e-8, -1e-8, 1.0, -1.0, 3.0])])
    x = np.concatenate([np.cos(ang), rng.normal(0, 50, 2000), np.array([1.0, -1.0, 1e-8, -1e-8, -4.0])])
    a, b = orc.atan2_v(y, x, orc.PORTABLE), orc.atan2_v(y, x, orc.LIBM)
    for yv, xv, av in zip(y, x, a):
        assert _is_nearest(mp, av, mp.atan2(mp.mpf(float(yv)), mp.mpf(float(xv)))), (yv, xv, av)
    # the inputs on which glibc disagrees, from a larger sample: the shared routine is the nearest one there too
    ang = rng.uniform(-np.pi, np.pi, 300_000)
    y, x = np.sin(ang), np.cos(ang)
    a, b = orc.atan2_v(y, x, orc.PORTABLE), orc.atan2_v(y, x, orc.LIBM)
    dis = np.flatnonzero(a != b)
    assert len(dis) > 20
    for k in dis[:300]:
        assert _is_nearest(mp, a[k], mp.atan2(mp.mpf(float(y[k])), mp.mpf(float(x[k]))))
    # sin / cos on the orientation range and beyond
    t = np.concatenate([rng.uniform(-np.pi, np.pi, 6000), rng.uniform(-50, 50, 1500), np.array([1e-9, -1e-9, np.pi / 2, np.pi])])
    s, c = orc.sincos_v(t, orc.PORTABLE)
    for tv, sv, cv in zip(t, s, c):
        m = mp.mpf(float(tv))
        assert _is_nearest(mp, sv, mp.sin(m)) and _is_nearest(mp, cv, mp.cos(m)), tv
    t = rng.uniform(-np.pi, np.pi, 300_000)
    s, c = orc.sincos_v(t, orc.PORTABLE)
    sm, cm = orc.sincos_v(t, orc.LIBM)
    for k in np.flatnonzero(s != sm)[:200]:
        assert _is_nearest(mp, s[k], mp.sin(mp.mpf(float(t[k]))))
    for k in np.flatnonzero(c != cm)[:200]:
        assert _is_nearest(mp, c[k], mp.cos(mp.mpf(float(t[k]))))
