#!/usr/bin/env python3
"""Worst-case error budget of the FP32 separable screen of the hybrid detector (toed_screen_fused_kernel), computed from the
kernel's own tap tables and accumulation order.  Prints the constants SCREEN_E_* of toed_kernels.hip; tests/test_screen_bound.py
re-derives them and checks that the kernel's constants are not smaller and that the tolerances cover them.

Model (u = 2^-24, round to nearest, FMA: one rounding per step):
  * pixels are recentred, x = v - 127.5 (exact in float), |x| <= X = 127.5; out-of-image pixels are v = 0, i.e. x = -127.5;
  * a chain s_k = fl(s_{k-1} + x_k c_k) with c_k = fl(K_k):  |s_n - sum x_k K_k| <= u X (sum|K| + sum_k cum_k) (1 + 20u),
    cum_k = |K_1| + ... + |K_k| in the order the kernel accumulates (each step's rounding is at most u times the magnitude
    of its result, which is at most X cum_k; the tap roundings add u X sum|K|);
  * column pass: inputs carry the row pass's error and are bounded by X sum|K_row|;
  * the DC term 127.5 * sum(K_row) * sum(K_col) is added back as a float constant: one more rounding of the result;
  * |g| = sqrt(fma(gx, gx, gy * gy)): 1-Lipschitz in (gx, gy) plus three roundings of a value <= G_MAX;
  * slope = minor / major, |major| >= |g| / sqrt(2) >= (2 - TOL_M) / sqrt(2): |d slope| <= 2 E_G / |major| + u.
"""
import math
import os
import re
import sys

U = 2.0 ** -24
X = 127.5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "edge_based_visual_odometry_amd", "csrc", "toed_kernels.hip")


def tables(src=SRC):
    text = open(src).read()

    def grab(name):
        m = re.search(r"const double %s\[4\]\[19\] = \{(.*?)\};" % name, text, re.S)
        rows = re.findall(r"\{(.*?)\}", m.group(1), re.S)
        return [[float(t) for t in r.replace("\n", " ").split(",") if t.strip()] for r in rows]

    return grab("h_TAP_INT"), grab("h_TAP_HALF")


def chain_bound(taps_in_order, xmax):
    """u * xmax * (sum|K| + sum_k cum_k) * (1 + 20 u) for an FMA chain over the taps in the given order"""
    cum, tot = 0.0, 0.0
    for k in taps_in_order:
        cum += abs(k)
        tot += cum
    return U * xmax * (cum + tot) * (1 + 20 * U)


def row_order(tab, n19):
    """toed_screen_fused_kernel's row pass: q = -8 .. 8 ascending (index q + 9), then the taps q = -9 (index 0) and q = +9
    (index 18) for the 19-tap planes"""
    o = [tab[q + 9] for q in range(-8, 9)]
    return o + ([tab[0], tab[18]] if n19 else [])


def col_order(tab, pm):
    return [tab[p + 9] for p in range(-pm, pm + 1)]


def budget():
    ti, th = tables()
    worst = dict(g=0.0, gmax=0.0)
    per_phase = {}
    for sy in (0, 1):
        for sx in (0, 1):
            ip = sy == 0 and sx == 0
            xt = th if sx else ti          # taps along x (row pass)
            yt = th if sy else ti          # taps along y (column pass)
            n19 = not ip
            pm = 8 if ip else 9
            rG, rGx = row_order(xt[0], n19), row_order(xt[1], n19)
            cG, cGx = col_order(yt[0], pm), col_order(yt[1], pm)
            eRG, eRGx = chain_bound(rG, X), chain_bound(rGx, X)
            maxRG = X * sum(abs(k) for k in rG) + eRG
            maxRGx = X * sum(abs(k) for k in rGx) + eRGx
            # gx = sum_p R_Gx[p] G[p]; gy = sum_p R_G[p] Gx[p]
            egx = sum(abs(k) for k in cG) * eRGx + chain_bound(cG, maxRGx)
            egy = sum(abs(k) for k in cGx) * eRG + chain_bound(cGx, maxRG)
            gxmax = maxRGx * sum(abs(k) for k in cG)
            gymax = maxRG * sum(abs(k) for k in cGx)
            # DC constants (added in float: one rounding of the result, plus the rounding of the constant itself)
            cx = 127.5 * sum(rGx) * sum(cG)
            cy = 127.5 * sum(rG) * sum(cGx)
            egx += U * (gxmax + abs(cx)) + U * abs(cx)
            egy += U * (gymax + abs(cy)) + U * abs(cy)
            per_phase[(sy, sx)] = dict(egx=egx, egy=egy, cx=cx, cy=cy, gxmax=gxmax + abs(cx), gymax=gymax + abs(cy))
            worst["g"] = max(worst["g"], egx, egy)
            worst["gmax"] = max(worst["gmax"], math.hypot(gxmax + abs(cx), gymax + abs(cy)))
    e_g = worst["g"]
    g_max = worst["gmax"]
    e_m = math.sqrt(2.0) * e_g + 3 * U * g_max * (1 + 1e-6)
    return dict(E_G=e_g, E_M=e_m, G_MAX=g_max, per_phase=per_phase)


def slope_bound(e_g, tol_m):
    major_min = (2.0 - tol_m - 1e-3) / math.sqrt(2.0)
    return 2.0 * e_g / major_min + U


def compare_bound(e_m, g_max):
    """what a comparison `m >= f - tol` of two screened magnitudes needs besides TOL_S |p2 - p1|: the errors of m and of the
    two interpolated neighbours' endpoints (<= E_M each side) and the float arithmetic of fp = p1 (1 - s) + p2 s, of
    tol = TOL_M + TOL_S |p2 - p1| and of f - tol (<= 6 roundings of values <= G_MAX)"""
    return 2.0 * e_m + 6 * U * g_max


if __name__ == "__main__":
    b = budget()
    tol_m = float(sys.argv[1]) if len(sys.argv) > 1 else 5e-4
    e_s = slope_bound(b["E_G"], tol_m)
    print("per phase:")
    for k, v in b["per_phase"].items():
        print("  (sy, sx) = %s: |d gx| <= %.3e  |d gy| <= %.3e  DC constants %.6e %.6e  max |gx| %.2f |gy| %.2f" %
              (k, v["egx"], v["egy"], v["cx"], v["cy"], v["gxmax"], v["gymax"]))
    print("SCREEN_E_G   = %.3e   (|d gx|, |d gy|)" % b["E_G"])
    print("SCREEN_E_M   = %.3e   (|d |g||), max |g| = %.1f" % (b["E_M"], b["G_MAX"]))
    print("SCREEN_E_S   = %.3e   (|d slope|)" % e_s)
    print("TOL_M needs >= %.3e (a comparison of two screened magnitudes); TOL_S needs >= %.3e" %
          (compare_bound(b["E_M"], b["G_MAX"]), e_s))
