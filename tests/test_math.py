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
