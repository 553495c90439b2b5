/*
 * ebvo_math.h -- portable atan2 / sincos / exp shared by the HIP kernels and the CPU oracle.
 *
 * Why this exists: the reference calls libm's std::atan2 for the third-order edge
 * orientation (src/toed/cpu_toed.cpp:229) and std::sin / std::cos for the NCC patch
 * geometry (src/utility.cpp:84-87,151).  glibc and the ROCm device libm differ by ulps,
 * which would make orientations, patch coordinates and therefore NCC scores differ in
 * their last bits between CPU and GPU.  Both sides therefore evaluate the SAME routine:
 * double-double arithmetic built from + - * / only (no FMA; every translation unit that
 * includes this header is compiled with -ffp-contract=off), rounded once at the end.
 * The result is the correctly rounded value except when the exact value lies within
 * ~2^-70 (relative) of a rounding boundary, i.e. it equals a correctly rounding libm in
 * all but ~1e-6 of calls; tests/test_math.py measures the agreement with glibc.
 *
 * Tables are produced by tools/gen_math_tables.py (mpmath, 300 bits).
 *
 * Usable from C99, C++ and HIP device code.  Under hipcc the functions are __device__
 * only (the tables live in __constant__ memory).
 */
#ifndef EBVO_MATH_H
#define EBVO_MATH_H

#if defined(__HIPCC__)
#define EBVO_MATH_FN __device__ static inline
#define EBVO_MATH_CONST __device__ __constant__ static const
#else
#define EBVO_MATH_FN static inline
#define EBVO_MATH_CONST static const
#endif

typedef struct
{
    double hi, lo;
} ebvo_dd;

EBVO_MATH_CONST double ebvo_atan_tab[17][2] = {
    {0x0.0p+0, 0x0.0p+0},
    {0x1.ff55bb72cfdeap-5, -0x1.c934d86d23f1dp-60},
    {0x1.fd5ba9aac2f6ep-4, -0x1.cd37686760c17p-59},
    {0x1.7b97b4bce5b02p-3, 0x1.347b0b4f881cap-58},
    {0x1.f5b75f92c80ddp-3, 0x1.8ab6e3cf7afbdp-57},
    {0x1.362773707ebccp-2, -0x1.963a544b672d8p-57},
    {0x1.6f61941e4def1p-2, -0x1.c63aae6f6e918p-56},
    {0x1.a64eec3cc23fdp-2, -0x1.24dec1b50b7ffp-56},
    {0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56},
    {0x1.0657e94db30d0p-1, -0x1.d5b495f6349e6p-56},
    {0x1.1e00babdefeb4p-1, -0x1.928df287a668fp-58},
    {0x1.345f01cce37bbp-1, 0x1.1021137c71102p-55},
    {0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56},
    {0x1.5d58987169b18p-1, 0x1.0028e4bc5e7cap-57},
    {0x1.700a7c5784634p-1, -0x1.8c34d25aadef6p-56},
    {0x1.819d0b7158a4dp-1, -0x1.bf76229d3b917p-56},
    {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55},
};
EBVO_MATH_CONST double ebvo_sin_tab[14][2] = {
    {0x0.0p+0, 0x0.0p+0},
    {0x1.ffaaaeeed4edbp-5, -0x1.2d16d32684b69p-59},
    {0x1.feaaeee86ee36p-4, -0x1.afcb2bcc6f03bp-59},
    {0x1.7dc102fbaf2b5p-3, 0x1.5ab50e23c97c3p-59},
    {0x1.faaeed4f31577p-3, -0x1.15d88508e32b8p-57},
    {0x1.3ad129769d3d8p-2, 0x1.03d550487839ap-63},
    {0x1.7710255764214p-2, -0x1.6ead7314bb6cep-57},
    {0x1.b1d8305321617p-2, -0x1.ae242cb99f519p-56},
    {0x1.eaee8744b05f0p-2, -0x1.789b43c9b027dp-58},
    {0x1.110d0c4b69c3bp-1, 0x1.d918998809981p-55},
    {0x1.2b91dea88421ep-1, -0x1.fa371db216ab0p-55},
    {0x1.44eb381cf386bp-1, -0x1.3ed6c1e6a5505p-55},
    {0x1.5cffc16bf8f0dp-1, 0x1.96cb370eb578ap-55},
    {0x1.73b7680dea578p-1, -0x1.2248306dc12a2p-56},
};
EBVO_MATH_CONST double ebvo_cos_tab[14][2] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.ff0015549f4d3p-1, 0x1.328387b99426fp-55},
    {0x1.fc015527d5bd3p-1, 0x1.b68f35094efb8p-55},
    {0x1.f706bdf9ece1cp-1, -0x1.698c80c36dcb4p-55},
    {0x1.f01549f7deea1p-1, 0x1.d3c1e99e5cafdp-55},
    {0x1.e733ea0193d40p-1, -0x1.6428b3546ce13p-55},
    {0x1.dc6b7eb995912p-1, 0x1.4b364776dcd35p-58},
    {0x1.cfc6cfa52ad9fp-1, 0x1.8b5b5508f2a0dp-55},
    {0x1.c1528065b7d50p-1, -0x1.892111312e828p-55},
    {0x1.b11d04162a4c6p-1, 0x1.1dd561efbc0c2p-56},
    {0x1.9f368ed912f85p-1, -0x1.1d200c5791606p-55},
    {0x1.8bb105a5dc900p-1, 0x1.863e03e9474c1p-55},
    {0x1.769fec655211fp-1, -0x1.827d5cf8c68c5p-57},
    {0x1.6018526f563dfp-1, 0x1.46ca5e0e432d0p-55},
};
#define EBVO_PI_HI 0x1.921fb54442d18p+1
#define EBVO_PI_LO 0x1.1a62633145c07p-53
#define EBVO_PI_2_HI 0x1.921fb54442d18p+0
#define EBVO_PI_2_LO 0x1.1a62633145c07p-54
#define EBVO_PIO2_1 0x1.921fb54400000p+0
#define EBVO_PIO2_2 0x1.0b4611a600000p-34
#define EBVO_PIO2_3_HI 0x1.3198a2e037073p-69
#define EBVO_PIO2_3_LO 0x1.129024e088a68p-123
#define EBVO_2_PI 0x1.45f306dc9c883p-1
EBVO_MATH_CONST double ebvo_exp_tab[23][2] = {
    {0x1.6b0ff72deb89dp-1, -0x1.dabf5975c0c02p-57},
    {0x1.769652df22f7ep-1, 0x1.3445f7544e0efp-57},
    {0x1.827a561889716p-1, -0x1.6b2eab63020c1p-57},
    {0x1.8ebef9eac820bp-1, -0x1.797d4686c5393p-57},
    {0x1.9b674f8f2f3d8p-1, -0x1.51bfdbb129094p-55},
    {0x1.a876812c0877cp-1, -0x1.fd36226fadd44p-56},
    {0x1.b5efd29f24c26p-1, 0x1.3d5fd7d70a5edp-56},
    {0x1.c3d6a24ed8222p-1, -0x1.e1e0a76cb0685p-55},
    {0x1.d22e6a0197c03p-1, -0x1.32ae7bdaf1116p-55},
    {0x1.e0fabfbc702a4p-1, -0x1.8d0e700fcfb65p-56},
    {0x1.f03f56a88b5d8p-1, -0x1.bad3fd501a227p-55},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.08205601127edp+0, -0x1.9c7d0bdf15160p-54},
    {0x1.1082b577d34edp+0, 0x1.f56c680678897p-54},
    {0x1.192937074e0cdp+0, 0x1.a24f46336ea04p-54},
    {0x1.2216045b6f5cdp+0, -0x1.8c4a5df1ec7e5p-58},
    {0x1.2b4b58b372c79p+0, 0x1.404dd9f031676p-54},
    {0x1.34cb8170b5835p+0, 0x1.6a7062465be33p-55},
    {0x1.3e98deaa11dccp+0, -0x1.5722108fefcffp-54},
    {0x1.48b5e3c3e8186p+0, 0x1.9d9ef0eda6eabp-54},
    {0x1.5325180cfacf7p+0, 0x1.b28b660a648dap-54},
    {0x1.5de9176045ff5p+0, 0x1.da89923298baap-55},
    {0x1.690492cbf9433p+0, -0x1.812833f7d6e43p-55},
};
#define EBVO_LN2_1 0x1.62e42fee00000p-1
#define EBVO_LN2_2 0x1.a39ef35600000p-33
#define EBVO_LN2_3_HI 0x1.93c7673007e5fp-65
#define EBVO_LN2_3_LO -0x1.50bf0cbcd98d6p-120
#define EBVO_1_LN2 0x1.71547652b82fep+0

/* ---- error-free transformations (Dekker / Knuth), no FMA ---- */

EBVO_MATH_FN ebvo_dd ebvo_two_sum(double a, double b)
{
    ebvo_dd r;
    double s = a + b;
    double bb = s - a;
    r.lo = (a - (s - bb)) + (b - bb);
    r.hi = s;
    return r;
}

/* requires |a| >= |b| (or a == 0) */
EBVO_MATH_FN ebvo_dd ebvo_quick_two_sum(double a, double b)
{
    ebvo_dd r;
    double s = a + b;
    r.lo = b - (s - a);
    r.hi = s;
    return r;
}

EBVO_MATH_FN ebvo_dd ebvo_two_prod(double a, double b)
{
    ebvo_dd r;
    double p = a * b;
    double ta = 134217729.0 * a;
    double ah = ta - (ta - a);
    double al = a - ah;
    double tb = 134217729.0 * b;
    double bh = tb - (tb - b);
    double bl = b - bh;
    r.lo = ((ah * bh - p) + ah * bl + al * bh) + al * bl;
    r.hi = p;
    return r;
}

EBVO_MATH_FN ebvo_dd ebvo_dd_add(ebvo_dd a, ebvo_dd b)
{
    ebvo_dd s = ebvo_two_sum(a.hi, b.hi);
    ebvo_dd t = ebvo_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = ebvo_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return ebvo_quick_two_sum(s.hi, s.lo);
}

EBVO_MATH_FN ebvo_dd ebvo_dd_neg(ebvo_dd a)
{
    ebvo_dd r;
    r.hi = -a.hi;
    r.lo = -a.lo;
    return r;
}

EBVO_MATH_FN ebvo_dd ebvo_dd_sub(ebvo_dd a, ebvo_dd b)
{
    return ebvo_dd_add(a, ebvo_dd_neg(b));
}

EBVO_MATH_FN ebvo_dd ebvo_dd_add_d(ebvo_dd a, double b)
{
    ebvo_dd s = ebvo_two_sum(a.hi, b);
    s.lo += a.lo;
    return ebvo_quick_two_sum(s.hi, s.lo);
}

EBVO_MATH_FN ebvo_dd ebvo_dd_mul(ebvo_dd a, ebvo_dd b)
{
    ebvo_dd p = ebvo_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return ebvo_quick_two_sum(p.hi, p.lo);
}

EBVO_MATH_FN ebvo_dd ebvo_dd_mul_d(ebvo_dd a, double b)
{
    ebvo_dd p = ebvo_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return ebvo_quick_two_sum(p.hi, p.lo);
}

/* a / b for plain doubles, as a double-double */
EBVO_MATH_FN ebvo_dd ebvo_d_div_d(double a, double b)
{
    double q1 = a / b;
    ebvo_dd p = ebvo_two_prod(q1, b);
    double r = (a - p.hi) - p.lo;
    double q2 = r / b;
    return ebvo_quick_two_sum(q1, q2);
}

EBVO_MATH_FN ebvo_dd ebvo_dd_div_d(ebvo_dd a, double b)
{
    double q1 = a.hi / b;
    ebvo_dd p = ebvo_two_prod(q1, b);
    double r = ((a.hi - p.hi) - p.lo) + a.lo;
    double q2 = r / b;
    return ebvo_quick_two_sum(q1, q2);
}

EBVO_MATH_FN ebvo_dd ebvo_dd_div(ebvo_dd a, ebvo_dd b)
{
    double q1 = a.hi / b.hi;
    ebvo_dd r = ebvo_dd_sub(a, ebvo_dd_mul_d(b, q1));
    double q2 = r.hi / b.hi;
    r = ebvo_dd_sub(r, ebvo_dd_mul_d(b, q2));
    double q3 = r.hi / b.hi;
    ebvo_dd q = ebvo_quick_two_sum(q1, q2);
    return ebvo_dd_add_d(q, q3);
}

/* atan of a double-double t in [0, 1] */
EBVO_MATH_FN ebvo_dd ebvo_atan_dd01(ebvo_dd t)
{
    int k = (int)(t.hi * 16.0 + 0.5);
    ebvo_dd u;
    if (k == 0)
    {
        u = t;
    }
    else
    {
        double c = (double)k * 0.0625;
        ebvo_dd num = ebvo_dd_add_d(t, -c);
        ebvo_dd den = ebvo_dd_add_d(ebvo_dd_mul_d(t, c), 1.0);
        u = ebvo_dd_div(num, den);
    }
    /* |u| <= ~1/32: atan u = u - u^3/3 + u^5/5 - ... ; first two terms in double-double */
    ebvo_dd u2 = ebvo_dd_mul(u, u);
    ebvo_dd u3 = ebvo_dd_mul(u2, u);
    ebvo_dd t3 = ebvo_dd_div_d(u3, 3.0);
    double w = u.hi;
    double z = w * w;
    double poly = 1.0 / 13.0 - z * (1.0 / 15.0);
    poly = 1.0 / 11.0 - z * poly;
    poly = 1.0 / 9.0 - z * poly;
    poly = 1.0 / 7.0 - z * poly;
    poly = 1.0 / 5.0 - z * poly;
    poly = ((z * z) * w) * poly;
    ebvo_dd s = ebvo_dd_sub(u, t3);
    s = ebvo_dd_add_d(s, poly);
    ebvo_dd a;
    a.hi = ebvo_atan_tab[k][0];
    a.lo = ebvo_atan_tab[k][1];
    return ebvo_dd_add(a, s);
}

/*
 * atan2(y, x) with the libm conventions for zeros, infinities and NaN.
 * Replaces std::atan2 at src/toed/cpu_toed.cpp:229,273,317,361.
 */
EBVO_MATH_FN double ebvo_atan2(double y, double x)
{
    if (x != x || y != y)
        return x + y;
    double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    int xneg = __builtin_signbit(x) != 0;
    double r;
    if (ay == 0.0)
    {
        r = xneg ? EBVO_PI_HI : 0.0;
        return __builtin_copysign(r, y);
    }
    if (ax == 0.0)
        return __builtin_copysign(EBVO_PI_2_HI, y);
    int xinf = ax > 1.7976931348623157e308, yinf = ay > 1.7976931348623157e308;
    if (xinf || yinf)
    {
        if (xinf && yinf)
            r = xneg ? 0x1.2d97c7f3321d2p+1 /* 3pi/4 */ : 0x1.921fb54442d18p-1 /* pi/4 */;
        else if (yinf)
            r = EBVO_PI_2_HI;
        else
            r = xneg ? EBVO_PI_HI : 0.0;
        return __builtin_copysign(r, y);
    }
    int swap = ay > ax;
    ebvo_dd t = swap ? ebvo_d_div_d(ax, ay) : ebvo_d_div_d(ay, ax);
    ebvo_dd a = ebvo_atan_dd01(t);
    ebvo_dd c;
    if (swap)
    {
        c.hi = EBVO_PI_2_HI;
        c.lo = EBVO_PI_2_LO;
        a = ebvo_dd_sub(c, a);
    }
    if (xneg)
    {
        c.hi = EBVO_PI_HI;
        c.lo = EBVO_PI_LO;
        a = ebvo_dd_sub(c, a);
    }
    r = a.hi + a.lo;
    return __builtin_copysign(r, y);
}

/*
 * sin and cos of theta.  Replaces std::sin / std::cos at src/utility.cpp:84-87,151.
 * Accurate (last-bit) for |theta| < 2^20; orientations are in (-pi, pi].
 */
EBVO_MATH_FN void ebvo_sincos(double theta, double *sn, double *cs)
{
    if (theta != theta || __builtin_fabs(theta) > 1.7976931348623157e308)
    {
        *sn = theta - theta;
        *cs = theta - theta;
        return;
    }
    double fn = theta * EBVO_2_PI;
    int n = (int)(fn + (fn >= 0.0 ? 0.5 : -0.5));
    double dn = (double)n;
    /* r = theta - n*pi/2 in double-double (Cody-Waite, n*P1 and n*P2 are exact) */
    ebvo_dd r = ebvo_two_sum(theta, -(dn * EBVO_PIO2_1));
    r = ebvo_dd_add_d(r, -(dn * EBVO_PIO2_2));
    ebvo_dd p3;
    p3.hi = EBVO_PIO2_3_HI;
    p3.lo = EBVO_PIO2_3_LO;
    r = ebvo_dd_sub(r, ebvo_dd_mul_d(p3, dn));

    /* table step: r = c + u, c = k/16, |u| <= 1/32 */
    double f16 = r.hi * 16.0;
    int k = (int)(f16 + (f16 >= 0.0 ? 0.5 : -0.5));
    ebvo_dd u = ebvo_dd_add_d(r, -((double)k * 0.0625));
    int kneg = k < 0;
    int ka = kneg ? -k : k;
    if (ka > 13)
        ka = 13; /* unreachable for finite input: |r| <= pi/4 + eps */
    ebvo_dd sk, ck;
    sk.hi = ebvo_sin_tab[ka][0];
    sk.lo = ebvo_sin_tab[ka][1];
    ck.hi = ebvo_cos_tab[ka][0];
    ck.lo = ebvo_cos_tab[ka][1];
    if (kneg)
        sk = ebvo_dd_neg(sk);

    /* sin u = u - u^3/6 + u^5/120 - ... ; cos u = 1 - u^2/2 + u^4/24 - ... */
    ebvo_dd u2 = ebvo_dd_mul(u, u);
    ebvo_dd u3 = ebvo_dd_mul(u2, u);
    double w = u.hi;
    double z = w * w;
    double ps = 1.0 / 362880.0 - z * (1.0 / 39916800.0);
    ps = 1.0 / 5040.0 - z * ps;
    ps = 1.0 / 120.0 - z * ps;
    ps = ((z * z) * w) * ps;
    ebvo_dd su = ebvo_dd_sub(u, ebvo_dd_div_d(u3, 6.0));
    su = ebvo_dd_add_d(su, ps);
    double pc = 1.0 / 3628800.0 - z * (1.0 / 479001600.0);
    pc = 1.0 / 40320.0 - z * pc;
    pc = 1.0 / 720.0 - z * pc;
    pc = 1.0 / 24.0 - z * pc;
    pc = (z * z) * pc;
    ebvo_dd cu = ebvo_dd_mul_d(u2, -0.5);
    cu = ebvo_dd_add_d(cu, pc);
    cu = ebvo_dd_add_d(cu, 1.0);

    ebvo_dd s = ebvo_dd_add(ebvo_dd_mul(sk, cu), ebvo_dd_mul(ck, su));
    ebvo_dd c = ebvo_dd_sub(ebvo_dd_mul(ck, cu), ebvo_dd_mul(sk, su));
    double sv = s.hi + s.lo, cv = c.hi + c.lo;
    switch (n & 3)
    {
    case 0:
        *sn = sv;
        *cs = cv;
        break;
    case 1:
        *sn = cv;
        *cs = -sv;
        break;
    case 2:
        *sn = -sv;
        *cs = -cv;
        break;
    default:
        *sn = -cv;
        *cs = sv;
        break;
    }
}

/*
 * exp(x).  Replaces std::exp in the Gaussian weights of EdgeClusterer::computeGaussianAverage
 * (src/EdgeClusterer.cpp:104) and in the refinement confidence exp(-rms / delta)
 * (src/Stereo_Matches.cpp:1281).  x = k ln2 + j/32 + u with |u| <= 1/64: exp(j/32) from a double-double
 * table, exp(u) by its Taylor series (u, u^2/2, u^3/6 in double-double), one rounding at the end.  Last-bit
 * accurate for results in the normal range; results below 2^-1022 are rounded twice.
 */
EBVO_MATH_FN double ebvo_exp(double x)
{
    if (x != x)
        return x + x;
    if (x > 709.782712893384)
        return 1.7976931348623157e308 * 2.0; /* +inf */
    if (x < -745.2)
        return 0.0;
    double fk = x * EBVO_1_LN2;
    int k = (int)(fk + (fk >= 0.0 ? 0.5 : -0.5));
    double dk = (double)k;
    /* r = x - k ln2 in double-double (k * LN2_1 and k * LN2_2 are exact) */
    ebvo_dd r = ebvo_two_sum(x, -(dk * EBVO_LN2_1));
    r = ebvo_dd_add_d(r, -(dk * EBVO_LN2_2));
    ebvo_dd l3;
    l3.hi = EBVO_LN2_3_HI;
    l3.lo = EBVO_LN2_3_LO;
    r = ebvo_dd_sub(r, ebvo_dd_mul_d(l3, dk));
    double f32 = r.hi * 32.0;
    int j = (int)(f32 + (f32 >= 0.0 ? 0.5 : -0.5));
    if (j > 11)
        j = 11;
    if (j < -11)
        j = -11;
    ebvo_dd u = ebvo_dd_add_d(r, -((double)j * 0.03125));
    ebvo_dd u2 = ebvo_dd_mul(u, u);
    ebvo_dd u3 = ebvo_dd_mul(u2, u);
    double w = u.hi;
    double p = 1.0 / 40320.0 + w * (1.0 / 362880.0);
    p = 1.0 / 5040.0 + w * p;
    p = 1.0 / 720.0 + w * p;
    p = 1.0 / 120.0 + w * p;
    p = 1.0 / 24.0 + w * p;
    p = ((w * w) * (w * w)) * p;
    ebvo_dd e = ebvo_dd_add(u, ebvo_dd_mul_d(u2, 0.5));
    e = ebvo_dd_add(e, ebvo_dd_div_d(u3, 6.0));
    e = ebvo_dd_add_d(e, p);
    e = ebvo_dd_add_d(e, 1.0);
    ebvo_dd t;
    t.hi = ebvo_exp_tab[j + 11][0];
    t.lo = ebvo_exp_tab[j + 11][1];
    e = ebvo_dd_mul(t, e);
    double v = e.hi + e.lo;
    return __builtin_ldexp(v, k); /* exact scaling (v_ldexp_f64 on the device, scalbn on the host) */
}

/*
 * Single-precision helpers of the fixed-scale SIFT descriptor (cv::SIFT::compute at given keypoints; OpenCV is not in the
 * reference tree, see oracle/ebvo_oracle.c: orc_sift_*).  OpenCV evaluates the Gaussian sample weights with its own
 * table-driven hal::exp32f and the gradient orientations with hal::fastAtan2, both approximations whose last bits depend
 * on the build (SIMD width, FMA); here both sides (oracle and kernels) call these two routines, float arithmetic with
 * separate multiply and add:
 *   ebvo_expf          exp(x) by Cody-Waite reduction and a degree-6 Taylor polynomial, about 1 ulp;
 *   ebvo_fast_atan2_deg  OpenCV's published fastAtan2 polynomial (degrees in [0, 360), max error ~0.3 degrees),
 *                        modules/core/src/mathfuncs_core.simd.hpp: atan_f32.
 */
EBVO_MATH_FN float ebvo_expf(float x)
{
    if (x != x)
        return x;
    if (x > 88.0f)
        return 3.4028234663852886e38f * 2.0f;
    if (x < -87.0f)
        return 0.0f;
    const float fk = x * 1.44269504088896341f;
    const int k = (int)(fk + (fk >= 0.0f ? 0.5f : -0.5f));
    const float dk = (float)k;
    const float r = (x - dk * 0.693359375f) - dk * -2.12194440e-4f;
    float p = 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    return __builtin_ldexpf(p, k);
}

EBVO_MATH_FN float ebvo_fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float a, c, c2;
    if (ax >= ay)
    {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else
    {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0)
        a = 180.f - a;
    if (y < 0)
        a = 360.f - a;
    return a;
}

#endif /* EBVO_MATH_H */
