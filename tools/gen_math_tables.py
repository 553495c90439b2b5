#!/usr/bin/env python3
"""Generate the double-double constant tables used by csrc/ebvo_math.h.

Prints C initialisers (hex-float literals) for:
  atan(k/16), sin(k/16), cos(k/16) as (hi, lo) pairs, pi, pi/2, and the
  three-part Cody-Waite split of pi/2.  Uses mpmath at 200 bits.
"""
import mpmath as mp

mp.mp.prec = 300


def dd(x):
    hi = float(x)
    lo = float(x - mp.mpf(hi))
    return hi, lo


def fmt(x):
    return float(x).hex()


def table(name, fn, n):
    print(f"EBVO_MATH_CONST double {name}[{n}][2] = {{")
    for k in range(n):
        hi, lo = dd(fn(mp.mpf(k) / 16))
        print(f"    {{{fmt(hi)}, {fmt(lo)}}},")
    print("};")


table("ebvo_atan_tab", mp.atan, 17)
table("ebvo_sin_tab", mp.sin, 14)
table("ebvo_cos_tab", mp.cos, 14)
for name, v in (("EBVO_PI", mp.pi), ("EBVO_PI_2", mp.pi / 2)):
    hi, lo = dd(v)
    print(f"#define {name}_HI {fmt(hi)}\n#define {name}_LO {fmt(lo)}")

# Cody-Waite: P1, P2 carry 33 significant bits each, P3 is the dd remainder.
def chop(x, bits):
    m, e = mp.frexp(x)
    return mp.ldexp(mp.floor(mp.ldexp(m, bits)), e - bits)

p = mp.pi / 2
p1 = chop(p, 33)
p2 = chop(p - p1, 33)
p3 = p - p1 - p2
p3hi, p3lo = dd(p3)
print(f"#define EBVO_PIO2_1 {fmt(p1)}\n#define EBVO_PIO2_2 {fmt(p2)}")
print(f"#define EBVO_PIO2_3_HI {fmt(p3hi)}\n#define EBVO_PIO2_3_LO {fmt(p3lo)}")
print(f"#define EBVO_2_PI {fmt(2 / mp.pi)}")

# ---- exp: exp(j/32) for j = -11 .. 11, and the Cody-Waite split of ln 2 ----
print("EBVO_MATH_CONST double ebvo_exp_tab[23][2] = {")
for j in range(-11, 12):
    hi, lo = dd(mp.exp(mp.mpf(j) / 32))
    print(f"    {{{fmt(hi)}, {fmt(lo)}}},")
print("};")
l2 = mp.log(2)
l1 = chop(l2, 32)
l2b = chop(l2 - l1, 32)
l3 = l2 - l1 - l2b
l3hi, l3lo = dd(l3)
print(f"#define EBVO_LN2_1 {fmt(l1)}\n#define EBVO_LN2_2 {fmt(l2b)}")
print(f"#define EBVO_LN2_3_HI {fmt(l3hi)}\n#define EBVO_LN2_3_LO {fmt(l3lo)}")
print(f"#define EBVO_1_LN2 {fmt(1 / l2)}")
