// ctm_big.cuh -- MMCTM / IMMCTM with MORE THAN 64 COORDINATES (64 < sum K <= 256, every K[m] <= 64), included by ctm.hip.
//
// The tuned kernels of ctm.hip give every coordinate of lambda / nu its own lane (16 / 32 / 64 lanes per document) and keep Sigma^-1
// in LDS.  The reference has no such limit (MMCTM.jl:29-91).  This file is the generic path behind them, written for coverage, not
// for speed: one WAVE per document, and
//   * the per-modality pieces (zeta, the softmax of props, exp(lambda - max), sum theta) map lane k to topic k of ONE modality at a
//     time (K[m] <= 64), so the number of modalities x topics does not matter;
//   * the pieces that couple all coordinates (the two LD_MMA solves, the Gaussian terms of the ELBO) give lane l the coordinates
//     l, l + 64, l + 128, l + 192 (CPLB = 4 slots, masked beyond sum K); Sigma^-1 is read through L2 (256 x 256 doubles do not fit LDS),
//     coalesced: lane l reads column entries l + 64 q of a row;
//   * theta sweeps keep their topic loops rolled, with a_k in LDS and one LDS column of partial sums per lane and topic (the scheme
//     of k_lda_estep_big); gamma statistics come from the posting sweep of the wide-table path (k_ctm_stats_terms<64>);
//   * the Gaussian M-step inverts Sigma in device memory (block_inverse_big).
// Same formulas, same operation order per coordinate as ctm.hip (MMCTM.jl:110-250, common.jl:11-36; LD_MMA as in mma_group); sums over a
// document are wave butterflies, so results agree with the tuned path / the CPU restatements to rounding, not bit for bit.  The kernels honour the
// flags, the replica index (blockIdx.y) and the activity flags of CtmEArgs, so fit, the stage API, inference and restart batches all work.
#pragma once

constexpr int kBigSlots = 4;          // coordinates per lane in the coupled pieces: sum K <= 256

// ---- theta phase (PH = 0 of k_ctm_estep, wide-table flavour): zeta, theta, sum theta, a_k rows for the posting sweep ---------------
// grid (blocks, replicas), 256 threads; dynamic LDS per wave: [64] a_k of the current modality | [64][64] column sums
__global__ __launch_bounds__(256) void k_ctm_theta_big(CtmEArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const CtmDims& dm = a.c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D, GT = dm.GT;
    const size_t rep = blockIdx.y;
    if (a.active && !a.active[rep]) return;
    const double* p_lam_in = a.lam_in + rep * D * MK;
    const double* p_nu = a.nu + rep * D * MK;
    double* p_zeta = a.zeta + rep * D * M;
    double* p_sumth = a.sumth + rep * D * MK;
    const double* __restrict__ p_expE = a.expE ? a.expE + rep * GT : nullptr;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, NW = blockDim.x >> 6;
    double* wav = smem + (size_t)wid * (64 + 64 * 64);
    double* wacc = wav + 64;
    const int flags = a.flags;
    for (int d = blockIdx.x * NW + wid; d < D; d += gridDim.x * NW) {
        for (int m = 0; m < M; ++m) {
            const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m];
            const bool act = lane < Km;
            const double lam = act ? p_lam_in[(size_t)d * MK + off + lane] : 0.0;
            if (a.lam_keep && act) a.lam_keep[(rep * D + d) * MK + off + lane] = lam;
            if (flags & F_ZETA) {       // update_ζ! (MMCTM.jl:172-181)
                const double nu = act ? p_nu[(size_t)d * MK + off + lane] : 1.0;
                const double zm = wave_sum(act ? ar_exp(lam + 0.5 * nu) : 0.0);
                if (lane == 0) p_zeta[(size_t)d * M + m] = zm;
            }
            if (!(flags & (F_THETA_COMPUTE | F_THETA_STORED))) continue;
            // update_θ! (MMCTM.jl:183-198): theta_kw = a_k e_kv / sum_k a_k e_kv with a = exp(lambda - max)
            const double mx = wave_max(act ? lam : -1e300);
            const double ak = act ? ar_exp(lam - mx) : 0.0;
            lds_wave_sync();
            wav[lane] = ak;
            if (a.aexp && act && (flags & F_SLAB)) a.aexp[(rep * D + d) * MK + off + lane] = ak;
            for (int k = 0; k < Km; ++k) wacc[(size_t)k * 64 + lane] = 0.0;
            lds_wave_sync();
            const double* __restrict__ tb = p_expE ? p_expE + dm.goff[m] : nullptr;
            const int64_t* dp = a.c.doc_ptr + (size_t)m * (D + 1);
            const int64_t start = dp[d];
            const int W = (int)(dp[d + 1] - start);
            for (int w = lane; w < W; w += 64) {
                const int2 tcv = a.c.tc[start + w];
                const double n = (double)tcv.y;
                double* th = a.theta ? a.theta + dm.toff[m] + (size_t)(start + w - dm.estart[m]) * Km : nullptr;
                if (flags & F_THETA_COMPUTE) {
                    double s = 0.0;
                    for (int k = 0; k < Km; ++k) s += wav[k] * tb[(size_t)k * Vm + tcv.x];
                    const double inv = dev_div(1.0, s);
                    const double r = n * inv;
                    for (int k = 0; k < Km; ++k) {
                        const double e = wav[k] * tb[(size_t)k * Vm + tcv.x];
                        wacc[(size_t)k * 64 + lane] += e * r;
                        if (flags & F_THETA_STORE) th[k] = e * inv;
                    }
                } else {
                    for (int k = 0; k < Km; ++k) wacc[(size_t)k * 64 + lane] += th[k] * n;
                }
            }
            lds_wave_sync();
            if (act) {       // sumθ_k (MMCTM.jl:110-117): lane k adds its topic's 64 column sums, starting at column k (bank rotation)
                const double* row = wacc + (size_t)lane * 64;
                double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
                for (int j = 0; j < 64; j += 4) {
                    r0 += row[(j + lane) & 63]; r1 += row[(j + 1 + lane) & 63]; r2 += row[(j + 2 + lane) & 63]; r3 += row[(j + 3 + lane) & 63];
                }
                p_sumth[(size_t)d * MK + off + lane] = (r0 + r1) + (r2 + r3);
            }
            lds_wave_sync();
        }
    }
    if (a.expE_keep && blockIdx.x == 0 && p_expE) for (int i = threadIdx.x; i < GT; i += blockDim.x) a.expE_keep[rep * GT + i] = p_expE[i];
}

// ---- the two LD_MMA solves (PH = 1), one wave per document, kBigSlots coordinates per lane --------------------------------------
struct BigDoc {          // what a lane holds of its document: coordinates i_q = lane + 64 q
    bool on[kBigSlots];
    double c[kBigSlots];             // Ndivζ (MMCTM.jl:119-125)
};

struct NuObjBig {
    const BigDoc* dc; double lam[kBigSlots], Sll[kBigSlots];
    const double* tabs;   // LDS: [exp table | log table] (stage_solve_tabs)
    __device__ __forceinline__ double eval(const double (&x)[kBigSlots], double (&g)[kBigSlots]) const
    {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) {
            g[q] = 0.0;
            if (dc->on[q]) {
                const double E = ar_exp_tab(lam[q] + 0.5 * x[q], tabs);
                g[q] = 0.5 * Sll[q] + 0.5 * dc->c[q] * E - dev_div(1.0, 2.0 * x[q]);
                t += 0.5 * x[q] * Sll[q] + dc->c[q] * E - 0.5 * ar_log_tab(x[q], tabs + MMM_EXPTAB_N);
            }
        }
        return wave_sum(t);
    }
};

struct LamObjBig {
    const BigDoc* dc; double nu[kBigSlots], sumth[kBigSlots], mu[kBigSlots];
    const double* S;      // Sigma^-1 [j * MK + i], symmetric, device memory (L2-resident)
    double* scr;          // the wave's LDS row, MK doubles: the differences x - mu of the whole document
    const double* tabs;   // LDS: [exp table | log table]
    int MK, lane;
    __device__ __forceinline__ double eval(const double (&x)[kBigSlots], double (&g)[kBigSlots]) const
    {
        double diff[kBigSlots], Sd[kBigSlots];
        lds_wave_sync();
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) { diff[q] = dc->on[q] ? x[q] - mu[q] : 0.0; if (dc->on[q]) scr[lane + 64 * q] = diff[q]; Sd[q] = 0.0; }
        lds_wave_sync();
        for (int j = 0; j < MK; ++j) {
            const double dj = scr[j];
            const double* row = S + (size_t)j * MK + lane;
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) if (dc->on[q]) Sd[q] = fma(row[64 * q], dj, Sd[q]);
        }
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) {
            g[q] = 0.0;
            if (dc->on[q]) {
                const double E = ar_exp_tab(x[q] + 0.5 * nu[q], tabs);
                g[q] = Sd[q] - sumth[q] + dc->c[q] * E;
                t += 0.5 * diff[q] * Sd[q] - x[q] * sumth[q] + dc->c[q] * E;
            }
        }
        return wave_sum(t);
    }
};

// NLopt LD_MMA, zero constraints, for the document of the calling wave (the algorithm and the formulas of mma_group in ctm.hip; one
// document per wave, so the control flow is uniform and plain branches do).  Returns the number of objective evaluations (negative: cap hit).
template <class Obj>
__device__ int mma_big(const Obj& obj, const BigDoc& dc, double (&x)[kBigSlots], bool has_lb, double lb, const SolveOpts& o)
{
    double sigma[kBigSlots], grad[kBigSlots], gcur[kBigSlots], xc[kBigSlots], xprev[kBigSlots], xprevprev[kBigSlots];
    double rho = 1.0;
    double fbest = obj.eval(x, grad);
#pragma unroll
    for (int q = 0; q < kBigSlots; ++q) { sigma[q] = 1.0; xc[q] = x[q]; xprev[q] = x[q]; xprevprev[q] = x[q]; }
    int k = 1, nev = 1;
    bool nonfin = !isfinite(fbest);
    const int cap = o.max_eval > 0 ? o.max_eval : 2000;
    for (;;) {
        double gl = 0.0, wl = 0.0;
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) {
            xc[q] = x[q];
            if (dc.on[q]) {
                const double sigma2 = sigma[q] * sigma[q];
                const double ags = fabs(grad[q]) * sigma[q];
                const double v = ags + 0.5 * rho;
                const double gs2 = grad[q] * sigma2;
                double dx = dev_div(-gs2, v + dev_sqrt_pos(rho * (ags + 0.25 * rho)));      // see mma_group (ctm_estep.cuh): NLopt's step with one quotient
                double c = x[q] + dx;
                if (has_lb) c = dev_max_raw(c, lb);                  // (the clamps by v_max / v_min: see mma_group)
                c = dev_min_raw(dev_max_raw(c, x[q] - 0.9 * sigma[q]), x[q] + 0.9 * sigma[q]);
                xc[q] = c;
                dx = c - x[q];
                const double dx2 = dx * dx;
                const double denominv = dev_div(1.0, sigma2 - dx2);
                gl += (fma(v, dx, gs2) * dx) * denominv;
                wl += dx2 * denominv;
            }
        }
        const double gval = fbest + wave_sum(gl);
        const double wval = 0.5 * wave_sum(wl);
        const double fcur = obj.eval(xc, gcur);
        ++nev;
        nonfin = nonfin || !isfinite(fcur);
        const bool inner_done = gval >= fcur;
        if (fcur < fbest) {
            fbest = fcur;
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) { x[q] = xc[q]; grad[q] = gcur[q]; }
        }
        if (nev >= cap) return nev_code(nev, true, nonfin);
        if (!inner_done) {
            if (fcur > gval) rho = fmin(10.0 * rho, 1.1 * (rho + dev_div(fcur - gval, wval)));
            continue;
        }
        // an outer iteration is complete: NLopt's x-tolerance test on (xcur, xprev)
        bool stop;
        if (o.xtol_rule == 0) {
            double dn = 0.0, xn = 0.0; bool big = false;
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) if (dc.on[q]) { const double ad = fabs(xc[q] - xprev[q]); dn += ad; xn += fabs(xc[q]); big = big || !(ad < o.xtol_abs); }
            dn = wave_sum(dn); xn = wave_sum(xn);
            stop = (dn < o.xtol_rel * xn) || __ballot(big) == 0ull;
        } else {
            bool bad = false;
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) if (dc.on[q]) {
                const double ad = fabs(xc[q] - xprev[q]);
                const bool ok = isinf(xprev[q]) ? false : (ad < o.xtol_abs || ad < o.xtol_rel * (fabs(xc[q]) + fabs(xprev[q])) * 0.5 || (o.xtol_rel > 0 && xc[q] == xprev[q]));
                bad = bad || !ok;
            }
            stop = __ballot(bad) == 0ull;
        }
        if (stop) return nev_code(nev, false, nonfin);
        rho = fmax(0.1 * rho, 1e-5);
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) {
            if (k > 1) {
                const double sgn = (xc[q] - xprev[q]) * (xprev[q] - xprevprev[q]);
                sigma[q] *= (sgn < 0 ? 0.7 : (sgn > 0 ? 1.2 : 1.0));
            }
            xprevprev[q] = xprev[q];
            xprev[q] = xc[q];
        }
        ++k;
    }
}

// grid (blocks, replicas), 256 threads; dynamic LDS: [waves][MK] difference vectors
__global__ __launch_bounds__(256) void k_ctm_solve_big(CtmEArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const CtmDims& dm = a.c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D;
    const size_t rep = blockIdx.y;
    if (a.active && !a.active[rep]) return;
    const double* __restrict__ S = a.invSigma + rep * MK * MK;
    const double* __restrict__ p_mu = a.mu + rep * MK;
    const double* p_lam_in = a.lam_in + rep * D * MK;
    double* p_lam_out = a.lam_out ? a.lam_out + rep * D * MK : nullptr;
    double* p_nu = a.nu + rep * D * MK;
    const double* p_zeta = a.zeta + rep * D * M;
    const double* p_sumth = a.sumth + rep * D * MK;
    int* p_nev_nu = a.nev_nu ? a.nev_nu + rep * D : nullptr;
    int* p_nev_lam = a.nev_lam ? a.nev_lam + rep * D : nullptr;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, NW = blockDim.x >> 6;
    __shared__ __attribute__((aligned(16))) double sTabs[MMM_EXPTAB_N + MMM_LOGTAB_N];      // exp | log tables of the objectives
    stage_solve_tabs(sTabs);
    __syncthreads();
    int mod[kBigSlots];
    double muq[kBigSlots], Sll[kBigSlots];
#pragma unroll
    for (int q = 0; q < kBigSlots; ++q) {
        const int i = lane + 64 * q;
        int mm = 0;
        for (int m = 0; m < M; ++m) if (i >= dm.koff[m] && i < dm.koff[m + 1]) mm = m;
        mod[q] = mm;
        muq[q] = i < MK ? p_mu[i] : 0.0;
        Sll[q] = i < MK ? S[(size_t)i * MK + i] : 0.0;
    }
    const SolveOpts o = a.opt;
    for (int d = blockIdx.x * NW + wid; d < D; d += gridDim.x * NW) {
        BigDoc dc;
        double lam[kBigSlots], nu[kBigSlots], sumth[kBigSlots];
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) {
            const int i = lane + 64 * q;
            dc.on[q] = i < MK;
            lam[q] = dc.on[q] ? p_lam_in[(size_t)d * MK + i] : 0.0;
            nu[q] = dc.on[q] ? p_nu[(size_t)d * MK + i] : 1.0;
            sumth[q] = dc.on[q] ? p_sumth[(size_t)d * MK + i] : 0.0;
            dc.c[q] = dc.on[q] ? a.c.Ndm[(size_t)d * M + mod[q]] / p_zeta[(size_t)d * M + mod[q]] : 0.0;
        }
        // update_ν! (MMCTM.jl:156-170): LD_MMA, lower bound 1e-7, from the current ν, with the old λ
        if (a.flags & F_NU) {
            NuObjBig obj; obj.dc = &dc; obj.tabs = sTabs;
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) { obj.lam[q] = lam[q]; obj.Sll[q] = Sll[q]; }
            const int nev = mma_big(obj, dc, nu, true, o.nu_lower, o);
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) if (dc.on[q]) p_nu[(size_t)d * MK + lane + 64 * q] = nu[q];
            if (p_nev_nu && lane == 0) p_nev_nu[d] = nev;
        }
        // update_λ! (MMCTM.jl:127-143): LD_MMA, unbounded, with the new ν
        if (a.flags & F_LAMBDA) {
            LamObjBig obj; obj.dc = &dc; obj.S = S; obj.scr = smem + (size_t)wid * MK; obj.MK = MK; obj.lane = lane; obj.tabs = sTabs;
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) { obj.nu[q] = nu[q]; obj.sumth[q] = sumth[q]; obj.mu[q] = muq[q]; }
            const int nev = mma_big(obj, dc, lam, false, 0.0, o);
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) if (dc.on[q]) p_lam_out[(size_t)d * MK + lane + 64 * q] = lam[q];
            if (p_nev_lam && lane == 0) p_nev_lam[d] = nev;
        }
    }
}

// ---- props = softmax(lambda block) (MMCTM.jl:145-154) and the per-modality ll numerators (MMCTM.jl:384-418) -------------------------
// grid (blocks, replicas), kBlockS threads; llpart[block][M]; dynamic LDS: [waves][64] props of the current modality
__global__ __launch_bounds__(kBlockS) void k_ctm_loglik_big(CtmDev c, const double* lam, const double* phieff, double* props, double* llpart,
                                                            int compute_ll, const int* active)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesS][kMaxM];
    const CtmDims& dm = c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D, GT = dm.GT;
    if (active && !active[blockIdx.y]) return;
    lam += (size_t)blockIdx.y * D * MK; phieff += (size_t)blockIdx.y * GT; llpart += (size_t)blockIdx.y * gridDim.x * M;
    if (props) props += (size_t)blockIdx.y * D * MK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double* sPr = smem + (size_t)wid * 64;
    double acc[kMaxM];
    for (int m = 0; m < kMaxM; ++m) acc[m] = 0.0;
    for (int d = blockIdx.x * kWavesS + wid; d < D; d += gridDim.x * kWavesS) {
        for (int m = 0; m < M; ++m) {
            const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m];
            const bool in = lane < Km;
            const double x = in ? lam[(size_t)d * MK + off + lane] : 0.0;
            const double mx = wave_max(in ? x : -1e300);
            const double e = in ? exp(x - mx) : 0.0;
            const double s = wave_sum(e);
            const double pr = e / s;
            if (in && props) props[(size_t)d * MK + off + lane] = pr;
            if (!compute_ll) continue;
            lds_wave_sync();
            sPr[lane] = pr;
            lds_wave_sync();
            const double* __restrict__ tb = phieff + dm.goff[m];
            const int64_t* dp = c.doc_ptr + (size_t)m * (D + 1);
            const int64_t start = dp[d];
            const int W = (int)(dp[d + 1] - start);
            double a = 0.0;
            for (int w = lane; w < W; w += 64) {
                const int2 t = c.tc[start + w];
                double p = 0.0;
                for (int k = 0; k < Km; ++k) p = fma(sPr[k], tb[(size_t)k * Vm + t.x], p);
                a += (double)t.y * log(p);
            }
            acc[m] += a;
        }
    }
    if (compute_ll) {
        for (int m = 0; m < M; ++m) { const double tot = wave_sum(acc[m]); if (lane == 0) shw[wid][m] = tot; }
        __syncthreads();
        if (tid < M) { double s = 0.0; for (int w = 0; w < kWavesS; ++w) s += shw[w][tid]; llpart[(size_t)blockIdx.x * M + tid] = s; }
    }
}

// ---- per-document ELBO pieces (MMCTM.jl:286-370), out[block][5] as k_ctm_elbo_docs; theta must be resident --------------------------
// dynamic LDS: [waves][MK] difference vectors
__global__ __launch_bounds__(kBlockS) void k_ctm_elbo_docs_big(CtmDev c, const double* __restrict__ invSigma, const double* mu, const double* lam, const double* nu,
                                                               const double* zeta, const double* theta, const double* Eeff, double* out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesS][5];
    const CtmDims& dm = c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double* scr = smem + (size_t)wid * MK;
    int mod[kBigSlots];
#pragma unroll
    for (int q = 0; q < kBigSlots; ++q) {
        const int i = lane + 64 * q;
        int mm = 0;
        for (int m = 0; m < M; ++m) if (i >= dm.koff[m] && i < dm.koff[m + 1]) mm = m;
        mod[q] = mm;
    }
    double t[5] = {0, 0, 0, 0, 0};
    for (int d = blockIdx.x * kWavesS + wid; d < D; d += gridDim.x * kWavesS) {
        double x[kBigSlots], v[kBigSlots], diff[kBigSlots], Sd[kBigSlots];
        lds_wave_sync();
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) {
            const int i = lane + 64 * q;
            const bool on = i < MK;
            x[q] = on ? lam[(size_t)d * MK + i] : 0.0; v[q] = on ? nu[(size_t)d * MK + i] : 1.0;
            diff[q] = on ? x[q] - mu[i] : 0.0; Sd[q] = 0.0;
            if (on) scr[i] = diff[q];
        }
        lds_wave_sync();
        for (int j = 0; j < MK; ++j) {
            const double dj = scr[j];
            const double* row = invSigma + (size_t)j * MK + lane;
#pragma unroll
            for (int q = 0; q < kBigSlots; ++q) if (lane + 64 * q < MK) Sd[q] = fma(row[64 * q], dj, Sd[q]);
        }
        double e0 = 0.0, e3 = 0.0, pz = 0.0;
#pragma unroll
        for (int q = 0; q < kBigSlots; ++q) {
            const int i = lane + 64 * q;
            if (i < MK) {
                e0 += -0.5 * (v[q] * invSigma[(size_t)i * MK + i] + diff[q] * Sd[q]);      // ElnPη without the constants (MMCTM.jl:286-300)
                e3 += -0.5 * log(v[q]);                                                     // ElnQη without the constant (MMCTM.jl:352-358)
                const double Nl = c.Ndm[(size_t)d * M + mod[q]], zl = zeta[(size_t)d * M + mod[q]];
                pz += -(Nl / zl) * exp(x[q] + 0.5 * v[q]);
            }
        }
        t[0] += wave_sum(e0); t[3] += wave_sum(e3);
        if (lane < M) { const double Nm = c.Ndm[(size_t)d * M + lane]; pz += Nm - Nm * log(zeta[(size_t)d * M + lane]); }
        double px = 0.0, qz = 0.0, lin = 0.0;
        for (int m = 0; m < M; ++m) {
            const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m];
            const int64_t* dp = c.doc_ptr + (size_t)m * (D + 1);
            const int64_t start = dp[d];
            const int W = (int)(dp[d + 1] - start);
            for (int w = lane; w < W; w += MMM_WAVE) {
                const int2 tc = c.tc[start + w];
                const double n = (double)tc.y;
                const double* th = theta + dm.toff[m] + (size_t)(start + w - dm.estart[m]) * Km;
                for (int k = 0; k < Km; ++k) {
                    const double p = th[k];
                    lin += n * p * lam[(size_t)d * MK + off + k];
                    px += n * p * Eeff[dm.goff[m] + (size_t)k * Vm + tc.x];
                    qz += n * dev_xlogx(p);
                }
            }
        }
        t[1] += wave_sum(pz + lin); t[2] += wave_sum(px); t[4] += wave_sum(qz);
    }
    if (lane == 0) for (int j = 0; j < 5; ++j) shw[wid][j] = t[j];
    __syncthreads();
    if (tid < 5) { double s = 0.0; for (int w = 0; w < kWavesS; ++w) s += shw[w][tid]; out[(size_t)blockIdx.x * 5 + tid] = s; }
}

// ---- Gauss-Jordan inverse with partial pivoting of an n x n matrix in DEVICE memory (row stride n), n <= 256, one block of 256 threads --------
// A is destroyed, Ainv receives the inverse, *logdet = log |det A|.  The algorithm of block_inverse_wide (pivot = largest |entry| of the
// column at or below the diagonal, ties to the smaller row), rows beyond one wave and matrices beyond LDS.
__device__ void block_inverse_big(int n, double* A, double* Ainv, double* logdet, int* singular)
{
    __shared__ double s_best[256];
    __shared__ int s_row[256];
    __shared__ double s_col[256];
    __shared__ int s_p;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n * n; i += nt) Ainv[i] = (i / n == i % n) ? 1.0 : 0.0;
    if (tid == 0) { *logdet = 0.0; *singular = 0; }
    __syncthreads();
    for (int c = 0; c < n; ++c) {
        double best = -1.0; int p = n;
        for (int r = c + tid; r < n; r += nt) { const double v = fabs(A[(size_t)r * n + c]); if (v > best) { best = v; p = r; } }
        s_best[tid] = best; s_row[tid] = p;
        __syncthreads();
        if (tid == 0) {
            double b = -1.0; int pr = n;
            for (int i = 0; i < nt && i < n - c; ++i) if (s_best[i] > b || (s_best[i] == b && s_row[i] < pr)) { b = s_best[i]; pr = s_row[i]; }
            s_p = pr < n ? pr : c;      // (no row compares greater only if the column holds NaNs: keep the row, as the LDS version does)
            if (b == 0.0) *singular = 1;      // (a NaN column is not an error upstream: see block_inverse_wide)
            *logdet += log(b);
        }
        __syncthreads();
        const int pr = s_p;
        const double piv = A[(size_t)pr * n + c];
        __syncthreads();
        for (int e = tid; e < 2 * n; e += nt) {
            double* Mx = e < n ? A : Ainv;
            const int j = e < n ? e : e - n;
            const double top = Mx[(size_t)c * n + j], low = Mx[(size_t)pr * n + j];
            Mx[(size_t)c * n + j] = low / piv;
            if (pr != c) Mx[(size_t)pr * n + j] = top;
        }
        __syncthreads();
        for (int r = tid; r < n; r += nt) s_col[r] = A[(size_t)r * n + c];
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {
            const int r = e / n, j = e % n;
            if (r == c) continue;
            const double f = s_col[r];
            A[(size_t)r * n + j] = (j == c) ? 0.0 : A[(size_t)r * n + j] - f * A[(size_t)c * n + j];
            Ainv[(size_t)r * n + j] -= f * Ainv[(size_t)c * n + j];
        }
        __syncthreads();
    }
}
