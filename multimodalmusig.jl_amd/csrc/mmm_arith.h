/* mmm_arith.h -- the scalar functions whose BITS decide where an LD_MMA solve stops (exp, log, digamma), written once
 * and compiled on both sides: by hipcc into the gfx950 kernels (csrc/ctm.hip) and by gcc into the order-matched CPU
 * restatement the parity tests check those kernels against.  Every operation is an IEEE-754 basic operation or an fma with a fixed
 * association, both compilers run with -ffp-contract=off, so the two sides produce identical bits for identical inputs.
 *
 * Why: NLopt's LD_MMA (call sites MMCTM.jl:127-170) stops on discontinuous tests (gval >= fcur, fcur < fbest, the
 * x-tolerance).  A 1-ulp difference in exp() between two libm's moves a stopping decision now and then, that document's
 * lambda by < xtol = 1e-4, and through the M-step every document of the next pass.  With one restatement of exp/log/
 * digamma on both sides the device and the CPU checker take the same decisions.
 *
 * exp: the algorithm of fdlibm's e_exp.c (argument reduction by ln2 in two pieces, degree-5 minimax in r^2, one
 * division); log: fdlibm's e_log.c kernel; both < 1 ulp.  digamma (x > 0): psi(x) = psi(x+7) - Q'(x)/Q(x), Q = prod_{v<7}
 * (x+v), then the 8-term asymptotic series of SpecialFunctions.jl (call sites MMCTM.jl:218; IMMCTM.jl:192-193).
 *
 * Division and square root: `/` and sqrt() on the host; dev_div / dev_sqrt on the device (rcp/rsq seed + Newton + residual
 * correction = the compiler's own IEEE sequences without their range handling, correctly rounded for normal-range operands;
 * checked bit for bit against the host by tests/test_arith_gpu.py through mmm_debug_math).
 */
#ifndef MMM_ARITH_H
#define MMM_ARITH_H

#if defined(__HIPCC__)
#define AR_FN __device__ __forceinline__
#define AR_DIV(a, b) dev_div((a), (b))
#define AR_BITS(x) ((unsigned long long)__double_as_longlong(x))
#define AR_FROM_BITS(u) __longlong_as_double((long long)(u))
#define AR_FMA(a, b, c) fma((a), (b), (c))
#define AR_RINT(x) rint(x)
#define AR_LDEXP(x, n) ldexp((x), (n))                 /* v_ldexp_f64: one rounding, subnormal results included -- as the host's ldexp */
#define AR_MAXNUM(a, b) dev_max_raw((a), (b))          /* IEEE maxNum / minNum (a NaN operand loses), without a canonicalising copy */
#define AR_MINNUM(a, b) dev_min_raw((a), (b))
#else
#include <math.h>
#include <stdint.h>
#include <string.h>
#define AR_FN static inline
#define AR_DIV(a, b) ((a) / (b))
static inline unsigned long long ar_bits_(double x) { unsigned long long u; memcpy(&u, &x, 8); return u; }
static inline double ar_from_bits_(unsigned long long u) { double x; memcpy(&x, &u, 8); return x; }
#define AR_BITS(x) ar_bits_(x)
#define AR_FROM_BITS(u) ar_from_bits_(u)
#define AR_FMA(a, b, c) fma((a), (b), (c))
#define AR_RINT(x) rint(x)
#define AR_LDEXP(x, n) ldexp((x), (n))
#define AR_MAXNUM(a, b) fmax((a), (b))
#define AR_MINNUM(a, b) fmin((a), (b))
#endif

/* exp(x), any x; NaN -> NaN.  Branch-free: the argument is clamped into the range where k = rint(x / ln 2) fits the two-step
 * scaling, the out-of-range results are selected at the end (kernels evaluate it for several coordinates per lane in straight-line
 * code; early returns would turn into divergent branches). */
AR_FN double ar_exp(double x)
{
    const double xc = x > 7.1e+02 ? 7.1e+02 : (x < -7.5e+02 ? -7.5e+02 : x);           /* NaN stays NaN (both comparisons false) */
    const double kd = AR_RINT(xc * 1.44269504088896338700e+00);
    const double hi = AR_FMA(-kd, 6.93147180369123816490e-01, xc);     /* exact: ln2HI has 21 trailing zero bits */
    const double lo = kd * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    double p = 4.13813679705723846039e-08;
    p = AR_FMA(p, t, -1.65339022054652515390e-06);
    p = AR_FMA(p, t, 6.61375632143793436117e-05);
    p = AR_FMA(p, t, -2.77777777770155933842e-03);
    p = AR_FMA(p, t, 1.66666666666666019037e-01);
    const double c = AR_FMA(-t, p, r);                                  /* r - t*P(t) */
    const double y = 1.0 - ((lo - AR_DIV(r * c, 2.0 - c)) - hi);
    /* y * 2^k in two exact-or-once-rounded steps (k in [-1083, 1025]) */
    const int k = (x != x) ? 0 : (int)kd;
    const int k1 = k / 2, k2 = k - k1;
    const double s1 = AR_FROM_BITS((unsigned long long)(1023 + k1) << 52);
    const double s2 = AR_FROM_BITS((unsigned long long)(1023 + k2) << 52);
    const double res = (y * s1) * s2;
    return x > 7.09782712893383973096e+02 ? AR_FROM_BITS(0x7ff0000000000000ull)       /* overflow */
         : (x < -7.45133219101941108420e+02 ? 0.0 : res);                            /* below the smallest subnormal; NaN falls through as NaN */
}

/* log(x), finite x > 0 (subnormals are scaled up first) */
AR_FN double ar_log(double x)
{
    const int sub = x < 2.2250738585072014e-308;                                         /* selects, not a branch (see ar_exp) */
    x = sub ? x * 18014398509481984.0 : x;                                               /* 2^54 */
    const unsigned long long u = AR_BITS(x);
    int e = (int)((u >> 52) & 0x7ff) - 1022 + (sub ? -54 : 0);                           /* x = m 2^e, m in [0.5, 1) */
    double m = AR_FROM_BITS((u & 0x000fffffffffffffull) | 0x3fe0000000000000ull);
    const int lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                                                  /* m in [sqrt(1/2), sqrt(2)) */
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = AR_DIV(f, 2.0 + f);
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = AR_FMA(R, z, 1.531383769920937332e-01);
    R = AR_FMA(R, z, 1.818357216161805012e-01);
    R = AR_FMA(R, z, 2.222219843214978396e-01);
    R = AR_FMA(R, z, 2.857142874366239149e-01);
    R = AR_FMA(R, z, 3.999999999940941908e-01);
    R = AR_FMA(R, z, 6.666666666666735130e-01);
    R *= z;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return AR_FMA(dk, 6.93147180369123816490e-01, f - (hfsq - AR_FMA(s, hfsq + R, dk * 1.90821492927058770002e-10)));
}

/* ---- table-driven exp and log for the LD_MMA objectives (the solve phase is bound by vector-f64 issue, and these two are evaluated once
 * per coordinate and objective evaluation): no division, a third fewer instructions, one 16-byte table read each.  `tab` points at the
 * values of csrc/mmm_exptab.h / mmm_logtab.h -- in LDS on the device, a static array in the CPU restatement: same numbers, same bits.
 *
 * exp(x) = 2^m (hi_j + (lo_j + hi_j p(r))),  x = (128 m + j) ln2/128 + r, |r| <= ln2/256, hi_j + lo_j = 2^(j/128),
 * p(r) = r + r^2/2 + r^3/6 + r^4/24 + r^5/120 (truncation < 5e-19 relative).  The reduction is exact in its first step (ln2/128 split like
 * fdlibm's ln2: the high part has 21 trailing zero bits, |k| < 2^18).  < 0.51 ulp.  Any x; NaN -> NaN. */
AR_FN double ar_exp_tab(double x, const double* tab)
{
    double kd = AR_RINT(x * 1.84664965233787316142e+02);                      /* 128 / ln 2 */
    kd = AR_MINNUM(AR_MAXNUM(kd, -140000.0), 140000.0);                        /* keeps (int)kd defined; beyond +-758 the result is inf / 0 anyway */
    const double r = AR_FMA(-kd, 1.4907929134926466e-12, AR_FMA(-kd, 5.41521234663378e-03, x));     /* ln2LO/128, ln2HI/128 = 0x1.62e42feep-8 */
    const int k = (int)kd;
    const int j = k & 127, m = k >> 7;                                         /* arithmetic shift: k = 128 m + j, 0 <= j < 128 */
    const double hi = tab[2 * j], lo = tab[2 * j + 1];
    double q = AR_FMA(r, 8.33333333333333322e-03, 4.16666666666666644e-02);
    q = AR_FMA(q, r, 1.66666666666666657e-01);
    q = AR_FMA(q, r, 0.5);
    const double p = AR_FMA(r * r, q, r);
    const double y = hi + AR_FMA(hi, p, lo);
    const double res = AR_LDEXP(y, m);
    return x < -7.45133219101941108420e+02 ? 0.0 : res;                        /* (far below: r, p are -inf and y is not a number worth scaling) */
}

/* log(x) for NORMAL x > 0 (the nu of an LD_MMA solve: >= its lower bound 1e-7): x = 2^e m, m in [1, 2), interval j = top 7 mantissa
 * bits with midpoint c_j, r = m / c_j - 1 (|r| < 2^-8; the table holds 1 / c_j and log c_j), log x = e ln 2 + log c_j + log1p(r),
 * log1p by its series to r^6.  Absolute error < 2.5e-15 for x <= 30 (a few ulp of |log x| away from 1; the objective it enters is
 * O(10..1e4), i.e. it stays below that sum's own rounding); no division. */
AR_FN double ar_log_tab(double x, const double* tab)
{
    const unsigned long long u = AR_BITS(x);
    const int hw = (int)(u >> 32);
    const int e = (hw >> 20) - 1023, j = (hw >> 13) & 127;
    const double m = AR_FROM_BITS((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double ic = tab[2 * j], lc = tab[2 * j + 1];
    const double r = AR_FMA(m, ic, -1.0);
    double p = AR_FMA(r, -1.0 / 6.0, 0.2);
    p = AR_FMA(p, r, -0.25);
    p = AR_FMA(p, r, 1.0 / 3.0);
    p = AR_FMA(p, r, -0.5);
    p = AR_FMA(p, r, 1.0);
    return AR_FMA((double)e, 0.6931471805599453, lc + p * r);
}

/* psi(x) for x >= 7: log x - 1/(2x) - sum_k B_2k / (2k x^2k), 8 terms (the series of SpecialFunctions.jl) */
AR_FN double ar_digamma_series(double x)
{
    double t = AR_DIV(1.0, x);
    const double psi = ar_log(x) - 0.5 * t;
    t *= t;
    double p = -0.4432598039215686;
    p = AR_FMA(p, t, 0.08333333333333333);
    p = AR_FMA(p, t, -0.021092796092796094);
    p = AR_FMA(p, t, 0.007575757575757576);
    p = AR_FMA(p, t, -0.004166666666666667);
    p = AR_FMA(p, t, 0.003968253968253968);
    p = AR_FMA(p, t, -0.008333333333333333);
    p = AR_FMA(p, t, 0.08333333333333333);
    return psi - t * p;
}

/* psi(x), 0 < x < 1e40 (every argument on the CTM path is a Dirichlet parameter or a sum of them) */
AR_FN double ar_digamma_pos(double x)
{
    double q = x, dq = 1.0;
    for (int v = 1; v < 7; ++v) {
        const double f = x + (double)v;
        dq = AR_FMA(dq, f, q);
        q *= f;
    }
    return ar_digamma_series(x + 7.0) - AR_DIV(dq, q);
}

/* psi(x), 0 < x < 1e40, with the table-driven log (ar_log_tab): the same recurrence and series, 20 instructions fewer.  For the LDA E-step
 * kernels of large corpora, whose prologue (K + 1 digammas and K exps per document) is a third of their vector work; absolute error of
 * the log (< 2.5e-15) carries over -- LDA results are compared at 1e-9, no bit-identity rests on this function. */
AR_FN double ar_digamma_pos_tab(double x, const double* logtab)
{
    double q = x, dq = 1.0;
    for (int v = 1; v < 7; ++v) {
        const double f = x + (double)v;
        dq = AR_FMA(dq, f, q);
        q *= f;
    }
    const double y = x + 7.0;
    double t = AR_DIV(1.0, y);
    const double psi = ar_log_tab(y, logtab) - 0.5 * t;
    t *= t;
    double p = -0.4432598039215686;
    p = AR_FMA(p, t, 0.08333333333333333);
    p = AR_FMA(p, t, -0.021092796092796094);
    p = AR_FMA(p, t, 0.007575757575757576);
    p = AR_FMA(p, t, -0.004166666666666667);
    p = AR_FMA(p, t, 0.003968253968253968);
    p = AR_FMA(p, t, -0.008333333333333333);
    p = AR_FMA(p, t, 0.08333333333333333);
    return (psi - t * p) - AR_DIV(dq, q);
}

#endif /* MMM_ARITH_H */
