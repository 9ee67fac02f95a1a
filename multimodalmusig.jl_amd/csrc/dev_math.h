// dev_math.h -- device-side scalar math and wavefront (64-lane) collectives for gfx950
#pragma once
#include <hip/hip_runtime.h>

#define MMM_WAVE 64

// digamma, same algorithm as SpecialFunctions.jl (reflection for x <= 0, recurrence to x >= 7, 8-term
// asymptotic series); call sites it replaces: LDA.jl:79,97; MMCTM.jl:218; IMMCTM.jl:192-193.
__device__ __forceinline__ double dev_digamma(double x)
{
    double psi = 0.0;
    if (x <= 0.0) {
        psi -= M_PI / tan(M_PI * x);
        x = 1.0 - x;
    }
    if (x < 7.0) {
        // psi(x) = psi(x+n) - sum_{v=0}^{n-1} 1/(x+v); pair the terms to halve the divisions
        int n = 7 - (int)floor(x);
        for (int v = 1; v < n; ++v) psi -= 1.0 / (x + (double)v);
        psi -= 1.0 / x;
        x += (double)n;
    }
    double t = 1.0 / x;
    psi += log(x) - 0.5 * t;
    t *= t;
    double p = -0.4432598039215686;
    p = fma(p, t, 0.08333333333333333);
    p = fma(p, t, -0.021092796092796094);
    p = fma(p, t, 0.007575757575757576);
    p = fma(p, t, -0.004166666666666667);
    p = fma(p, t, 0.003968253968253968);
    p = fma(p, t, -0.008333333333333333);
    p = fma(p, t, 0.08333333333333333);
    psi -= t * p;
    return psi;
}

// x*log(x) with the reference's 0^0 = 1 convention of log(x^x) (LDA.jl:157; MMCTM.jl:365)
__device__ __forceinline__ double dev_xlogx(double x) { return x > 0.0 ? x * log(x) : 0.0; }

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
