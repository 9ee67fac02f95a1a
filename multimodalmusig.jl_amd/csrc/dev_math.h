// dev_math.h -- device-side scalar math and wavefront (64-lane) collectives for gfx950
#pragma once
#include <hip/hip_runtime.h>

#define MMM_WAVE 64

// digamma for any real x, same algorithm as SpecialFunctions.jl (reflection for x <= 0, recurrence to x >= 7,
// 8-term asymptotic series); call sites it replaces: LDA.jl:79,97; MMCTM.jl:218; IMMCTM.jl:192-193.
__device__ __forceinline__ double dev_digamma_series(double x)   // x >= 7
{
    double t = 1.0 / x;
    double psi = log(x) - 0.5 * t;
    t *= t;
    double p = -0.4432598039215686;
    p = fma(p, t, 0.08333333333333333);
    p = fma(p, t, -0.021092796092796094);
    p = fma(p, t, 0.007575757575757576);
    p = fma(p, t, -0.004166666666666667);
    p = fma(p, t, 0.003968253968253968);
    p = fma(p, t, -0.008333333333333333);
    p = fma(p, t, 0.08333333333333333);
    return psi - t * p;
}

// x > 0 (every argument on the hot path is a Dirichlet parameter): csrc/mmm_arith.h's ar_digamma_pos -- psi(x) = psi(x+7) - Q'(x)/Q(x),
// Q = prod_{v<7} (x+v) (one division instead of up to seven, no data-dependent trip count), with the 8-instruction division and the
// fdlibm-style log of that header: ~110 instructions where the ocml log and two IEEE division sequences took ~220.  Defined below.
__device__ __forceinline__ double dev_digamma_pos(double x);

__device__ __forceinline__ double dev_digamma(double x)
{
    if (x > 0.0 && x < 1e40) return dev_digamma_pos(x);
    double psi = 0.0;
    if (x <= 0.0) { psi -= M_PI / tan(M_PI * x); x = 1.0 - x; }
    if (x < 7.0) {
        int n = 7 - (int)floor(x);
        for (int v = 1; v < n; ++v) psi -= 1.0 / (x + (double)v);
        psi -= 1.0 / x;
        x += (double)n;
    }
    return psi + dev_digamma_series(x);
}

// 1/x to <= 1 ulp: hardware reciprocal seed (v_rcp_f64) + two Newton steps.  ~6 instructions instead of the ~15 of
// the IEEE division sequence; used where a quotient feeds a sum that is compared at >= 1e-11 relative.
__device__ __forceinline__ double dev_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// v_max_f64 without the canonicalising v_max(x, x) the compiler puts before fmax()
__device__ __forceinline__ double dev_max_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double dev_min_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// a / b, bit-identical to the compiler's IEEE division whenever no operand or intermediate leaves the normal range: the
// same rcp + 2 Newton + residual-correction sequence, without the v_div_scale / v_div_fmas / v_div_fixup range handling
// (8 instead of ~25 instructions).  For the MMA step algebra: all operands there are O(1e-7 .. 1e7).
__device__ __forceinline__ double dev_div(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    y = fma(fma(-b, y, 1.0), y, y);
    y = fma(fma(-b, y, 1.0), y, y);
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}

// sqrt(x) for x = 0 or x in the normal range, the compiler's rsq-based sequence without its subnormal scaling
__device__ __forceinline__ double dev_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = 0.5 * y;
    const double r0 = fma(-h0, g0, 0.5);
    const double g1 = fma(g0, r0, g0), h1 = fma(h0, r0, h0);
    const double g2 = fma(fma(-g1, g1, x), h1, g1);
    const double g3 = fma(fma(-g2, g2, x), h1, g2);
    return x == 0.0 ? 0.0 : g3;
}

// the same for a NORMAL x > 0 (no zero to special-case: three instructions fewer).  For the root of the LD_MMA step, rho (|g| sigma + rho / 4) >= rho^2 / 4 > 0.
__device__ __forceinline__ double dev_sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = 0.5 * y;
    const double r0 = fma(-h0, g0, 0.5);
    const double g1 = fma(g0, r0, g0), h1 = fma(h0, r0, h0);
    const double g2 = fma(fma(-g1, g1, x), h1, g1);
    return fma(fma(-g2, g2, x), h1, g2);
}

// exp / log / digamma whose bits the parity tests' order-matched CPU restatement reproduces (one source for both sides)
#include "mmm_arith.h"
__device__ __forceinline__ double dev_digamma_pos(double x) { return ar_digamma_pos(x); }
__device__ __forceinline__ double dev_digamma_ar(double x) { return (x > 0.0 && x < 1e40) ? ar_digamma_pos(x) : dev_digamma(x); }

// natural log of a NORMAL x > 0 from a 128-interval table in LDS (csrc/mmm_logtab.h, staged by the caller: tab[2 j] = 1 / c_j,
// tab[2 j + 1] = log c_j, c_j the midpoint of the mantissa interval): x = 2^e m, r = m / c_j - 1 (|r| < 2^-8), log x = e ln 2 + log c_j
// + log1p(r) with log1p by its series to r^6.  15 instructions and one 16-byte LDS read instead of ~35; absolute error < 5e-16
// (2.5e-15 relative where |log x| > 0.05) -- for the log-likelihood sweeps, whose sums are compared at 1e-9.
__device__ __forceinline__ double dev_log_pos(double x);
__device__ __forceinline__ double dev_log_tab(double x, const double* __restrict__ tab)
{
    const int hi = __double2hiint(x), lo = __double2loint(x);
    if (__builtin_expect(hi < 0x00100000, 0)) return x > 0.0 ? dev_log_pos(x) : (x == 0.0 ? -__builtin_inf() : __builtin_nan(""));      // zero, subnormal, negative: as log()
    const int e = (hi >> 20) - 1023, j = (hi >> 13) & 127;
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
    const double2 t = *reinterpret_cast<const double2*>(tab + 2 * j);
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, -1.0 / 6.0, 0.2);
    p = fma(p, r, -0.25);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    return fma((double)e, 0.6931471805599453, t.y + p * r);
}

// natural log for finite x > 0 (normal or subnormal-free inputs: probabilities and Dirichlet parameters), fdlibm-style:
// x = 2^e m, m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f/(2+f), log(1+f) = f - (f^2/2 - s (f^2/2 + R(s^2))).
// ~35 instructions (ocml's log is ~90); error < 2 ulp.
__device__ __forceinline__ double dev_log_pos(double x)
{
    int e = __builtin_amdgcn_frexp_exp(x);              // x = m * 2^e, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                 // m in [sqrt(1/2), sqrt(2))
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * dev_rcp(2.0 + f);
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = fma(R, z, 1.531383769920937332e-01);
    R = fma(R, z, 1.818357216161805012e-01);
    R = fma(R, z, 2.222219843214978396e-01);
    R = fma(R, z, 2.857142874366239149e-01);
    R = fma(R, z, 3.999999999940941908e-01);
    R = fma(R, z, 6.666666666666735130e-01);
    R *= z;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return fma(dk, 6.93147180369123816490e-01, f - (hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)));
}

// x*log(x) with the reference's 0^0 = 1 convention of log(x^x) (LDA.jl:157; MMCTM.jl:365)
__device__ __forceinline__ double dev_xlogx(double x) { return x > 0.0 ? x * log(x) : 0.0; }

// ---- cross-lane ---------------------------------------------------------------------------------------
// full-wave (64 lanes) butterfly sum: every lane ends with the total
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, MMM_WAVE);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, MMM_WAVE));
    return v;
}
__device__ __forceinline__ double wave_bcast(double v, int lane) { return __shfl(v, lane, MMM_WAVE); }
// the same for a lane known at compile time, through v_readlane: the result lives in scalar registers
__device__ __forceinline__ double wave_readlane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// DPP move of a double (two 32-bit halves); CTRL is a DPP control word (quad_perm 0x00-0xFF, row_shr 0x110+n,
// row_mirror 0x140, row_half_mirror 0x141)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    // bound_ctrl = 1 and no `old` operand: every control used here (quad_perm, row_half_mirror, row_mirror) has a source lane for
    // every lane, and without an `old` value the compiler emits the bare v_mov_b32_dpp pair (with old = src it copies the
    // register first: two extra VALU instructions per stage, 8 per 16-lane sum -- measured in the f64-VALU-bound solve kernel)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Largest value of the wave in every lane, without LDS: four DPP stages inside the 16-lane rows, then the rows through the row swaps
// of rows_sum4 below.
__device__ __forceinline__ double wave_max_dpp(double v);
// Sum over the four 16-lane rows of a wave, per row lane, valid in every lane: (row 0 + row 2) + (row 1 + row 3).  gfx950's row swaps
// (v_permlane32_swap: the upper 32 lanes of one register against the lower 32 of another; v_permlane16_swap: odd against even rows)
// applied to two copies of the value put lane i + 32 (then i + 16) beside lane i without a trip through LDS: 2 moves, 2 swaps and an
// add per stage.
typedef unsigned mmm_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double rows_sum4(double x)
{
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    mmm_u2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    mmm_u2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double s = __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
    lo = (unsigned)__double2loint(s); hi = (unsigned)__double2hiint(s);
    a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}

__device__ __forceinline__ double wave_max_dpp(double v)
{
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    mmm_u2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    mmm_u2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = fmax(__hiloint2double((int)b.x, (int)a.x), __hiloint2double((int)b.y, (int)a.y));
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
    a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return fmax(__hiloint2double((int)b.x, (int)a.x), __hiloint2double((int)b.y, (int)a.y));
}

// sum over each aligned group of L lanes (L = 16, 32 or 64); every lane of the group ends with the group total.
// The 16-lane part stays inside a DPP row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror.
template <int L>
__device__ __forceinline__ double group_sum(double v)
{
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    if (L >= 32) v += __shfl_xor(v, 16, MMM_WAVE);
    if (L >= 64) v += __shfl_xor(v, 32, MMM_WAVE);
    return v;
}
