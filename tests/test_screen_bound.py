"""The FP32 screen's worst-case error budget (hybrid TOED) as a checked invariant, CPU side:

* tools/screen_error_bound.py derives E_G, E_M, E_S from the tap tables and accumulation order of toed_kernels.hip; the
  constants compiled into the kernel (SCREEN_E_*) must not be smaller, and the tolerances of the relaxed test must cover a
  comparison of two screened quantities with the margins the kernel's static_asserts state;
* a float32 emulation of the screen's two passes (recentred pixels, FMA chains in the kernel's order, the DC constant added
  back) on saturating 0 / 255 inputs aligned with the tap signs stays inside the budget against exact arithmetic -- the
  algebra of the recentring (src/toed/cpu_toed.cpp:199-230 computes sum v K over the taps inside the image) and the bound at once.
"""
import importlib.util
import os
import re
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("screen_error_bound", os.path.join(ROOT, "tools", "screen_error_bound.py"))
seb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(seb)


def kernel_constants():
    text = open(seb.SRC).read()

    def num(name):
        return float(re.search(r"%s = ([0-9.eE+-]+)f?[;,]" % name, text).group(1))

    return {k: num(k) for k in ("SCREEN_TOL_M", "SCREEN_TOL_S", "SCREEN_E_G", "SCREEN_E_M", "SCREEN_E_S", "SCREEN_G_MAX")}


def test_kernel_constants_cover_the_derived_budget():
    c = kernel_constants()
    b = seb.budget()
    assert c["SCREEN_E_G"] >= b["E_G"] and c["SCREEN_E_G"] <= 1.05 * b["E_G"]
    assert c["SCREEN_E_M"] >= b["E_M"] and c["SCREEN_E_M"] <= 1.05 * b["E_M"]
    assert c["SCREEN_G_MAX"] >= b["G_MAX"]
    e_s = seb.slope_bound(b["E_G"], c["SCREEN_TOL_M"])
    assert c["SCREEN_E_S"] >= e_s and c["SCREEN_E_S"] <= 1.05 * e_s
    # the relaxed test's tolerances against what a comparison of two screened magnitudes / components / slopes needs
    assert c["SCREEN_TOL_M"] >= 1.5 * seb.compare_bound(b["E_M"], b["G_MAX"])
    assert c["SCREEN_TOL_M"] >= 3.0 * 2.0 * b["E_G"]
    assert c["SCREEN_TOL_S"] >= 2.0 * e_s


def fma32(a, b, c):
    """float32 fused multiply-add: the product of two floats is exact in double; one rounding of the double sum to float
    (a double rounding can differ from a true FMA only in a tie of the last double bit: irrelevant for a bound check)"""
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def screen_value(img, i, j, sy, sx, which, ti, th):
    """toed_screen_fused_kernel's gx (which = 0) or gy (1) at pixel (i, j) of phase (sy, sx), emulated in float32"""
    ip = sy == 0 and sx == 0
    xt, yt = (th if sx else ti), (th if sy else ti)
    pm = 8 if ip else 9
    kx = xt[1] if which == 0 else xt[0]
    ky = yt[0] if which == 0 else yt[1]
    h, w = img.shape

    def pix(ii, jj):
        v = img[ii, jj] if 0 <= ii < h and 0 <= jj < w else 0
        return np.float32(v) - np.float32(127.5)

    rows = {}
    for p in range(-pm, pm + 1):
        acc = np.float32(0)
        for q in range(-8, 9):
            acc = fma32(pix(i - p, j - q), np.float32(kx[q + 9]), acc)
        if not ip:                                   # the taps q = -9 and q = +9 last, as the kernel adds them
            acc = fma32(pix(i - p, j + 9), np.float32(kx[0]), acc)
            acc = fma32(pix(i - p, j - 9), np.float32(kx[18]), acc)
        rows[p] = acc
    g = np.float32(0)
    for p in range(-pm, pm + 1):
        g = fma32(rows[p], np.float32(ky[p + 9]), g)
    dc = np.float32(127.5 * sum(kx[q + 9] for q in range(-pm, pm + 1)) * sum(ky[p + 9] for p in range(-pm, pm + 1)))
    return np.float32(g + dc)


def exact_value(img, i, j, sy, sx, which, ti, th):
    ip = sy == 0 and sx == 0
    xt, yt = (th if sx else ti), (th if sy else ti)
    pm = 8 if ip else 9
    kx = xt[1] if which == 0 else xt[0]
    ky = yt[0] if which == 0 else yt[1]
    h, w = img.shape
    s = Fraction(0)
    for p in range(-pm, pm + 1):
        for q in range(-pm, pm + 1):
            if 0 <= i - p < h and 0 <= j - q < w:    # the reference skips taps outside the image (src/toed/cpu_toed.cpp:204)
                s += int(img[i - p, j - q]) * Fraction(kx[q + 9]) * Fraction(ky[p + 9])
    return float(s)


def test_emulated_screen_stays_inside_the_budget_on_saturating_inputs():
    ti, th = seb.tables()
    b = seb.budget()
    rng = np.random.default_rng(5)
    h, w = 40, 44
    worst = 0.0
    images = []
    for sx in (0, 1):                                 # pixels that follow the sign of the derivative taps: largest partial sums
        t = th if sx else ti
        img = np.zeros((h, w), dtype=np.uint8)
        for jj in range(w):
            q = 20 - jj
            img[:, jj] = 255 if (-9 <= q <= 9 and t[1][q + 9] > 0) else 0
        images += [img, img.T.copy()[:h, :w] if img.T.shape == (h, w) else np.ascontiguousarray(img.T)]
    images.append((rng.integers(0, 2, (h, w)) * 255).astype(np.uint8))
    images.append(np.full((h, w), 255, dtype=np.uint8))
    for img in images:
        hh, ww = img.shape
        pts = [(hh // 2, 20), (20 if hh > 20 else hh // 2, ww // 2), (3, 4), (hh - 2, ww - 3)]
        for (i, j) in pts:
            if not (0 <= i < hh and 0 <= j < ww):
                continue
            for sy in (0, 1):
                for sx in (0, 1):
                    for which in (0, 1):
                        e = abs(float(screen_value(img, i, j, sy, sx, which, ti, th)) - exact_value(img, i, j, sy, sx, which, ti, th))
                        worst = max(worst, e)
    assert worst <= b["E_G"], worst
    assert worst > 0.0                                # the emulation does round: the check is not vacuous
