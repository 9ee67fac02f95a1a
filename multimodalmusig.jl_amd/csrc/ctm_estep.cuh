// ctm_estep.cuh -- device code of the E-step of ctm.hip (included there, inside its anonymous namespace): lane-group helpers, the objective
// functors, NLopt LD_MMA as a device function, the theta-phase kernels (k_ctm_estep<.., 0>, k_ctm_theta_dense) and the solve-phase kernels
// (k_ctm_estep<.., 1>, k_ctm_solve_cpl).  MMCTM.jl:110-198, common.jl:11-36.
__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int L>
__device__ __forceinline__ double group_max(double v)
{
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    if (L >= 32) v = fmax(v, __shfl_xor(v, 16, MMM_WAVE));
    if (L >= 64) v = fmax(v, __shfl_xor(v, 32, MMM_WAVE));
    return v;
}

// true when `pred` is false on every lane of the caller's L-lane group
template <int L>
__device__ __forceinline__ bool group_none(bool pred, int g)
{
    const unsigned long long b = __ballot(pred);
    if (L == 64) return b == 0ull;
    const unsigned long long mask = ((1ull << (L & 63)) - 1ull) << (g * L);
    return (b & mask) == 0ull;
}

// ---- packed document groups (solve phase, sum K not a divisor of 64): LP = sum K lanes per document, floor(64 / LP) documents per
// wave instead of 64 / 16 -- sum K = 10 (BASELINE config 5): 6 documents per wave instead of 4.  Groups straddle the 16-lane DPP
// rows, so the group sum goes through the LDS crossbar (ds_bpermute, no VALU slot -- the solve phase is f64-VALU bound): a tree
// that folds lane l+off onto lane l for off = 8, 4, 2, 1 and broadcasts lane 0's total.  Lanes without a partner read a spare lane
// of the wave (64 % LP of them exist) whose value is 0 at every stage.
struct PackCtx { int a[5]; };       // byte addresses (lane * 4) of the partner per stage [8, 4, 2, 1] and of the group's lane 0

template <int LP>
__device__ __forceinline__ PackCtx pack_ctx(int lane)
{
    constexpr int G = MMM_WAVE / LP;
    static_assert(G * LP < MMM_WAVE, "packed groups need a spare lane");
    const int g = lane / LP, l = lane % LP;
    const bool in = g < G;
    PackCtx c;
    const int offs[4] = {8, 4, 2, 1};
#pragma unroll
    for (int q = 0; q < 4; ++q) c.a[q] = 4 * ((in && l < offs[q] && l + offs[q] < LP) ? lane + offs[q] : MMM_WAVE - 1);
    c.a[4] = 4 * (in ? g * LP : MMM_WAVE - 1);
    return c;
}

__device__ __forceinline__ double bperm_f64(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// v must be 0 in the spare lanes
template <int LP>
__device__ __forceinline__ double packed_sum(const PackCtx& c, double v)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) if ((8 >> q) < LP) v += bperm_f64(c.a[q], v);
    return bperm_f64(c.a[4], v);
}

// two independent sums through the same stages: their LDS round trips overlap (the packed path is bound by that latency)
template <int LP>
__device__ __forceinline__ void packed_sum2(const PackCtx& c, double& v, double& w)
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if ((8 >> q) < LP) { const double pv = bperm_f64(c.a[q], v), pw = bperm_f64(c.a[q], w); v += pv; w += pw; }
    const double tv = bperm_f64(c.a[4], v), tw = bperm_f64(c.a[4], w);
    v = tv; w = tw;
}

template <int L, int LP>
__device__ __forceinline__ void gsum2(const PackCtx& c, double& v, double& w)
{
    if constexpr (LP > 0) packed_sum2<LP>(c, v, w);
    else { v = group_sum<L>(v); w = group_sum<L>(w); }
}

// sum over the caller's document group: L-lane DPP rows (LP = 0) or packed LP-lane groups
template <int L, int LP>
__device__ __forceinline__ double gsum(const PackCtx& c, double v)
{
    if constexpr (LP > 0) return packed_sum<LP>(c, v);
    else return group_sum<L>(v);
}

template <int L, int LP>
__device__ __forceinline__ bool gnone(bool pred, int g)
{
    if constexpr (LP > 0) {
        const unsigned long long b = __ballot(pred);
        return (b & (((1ull << LP) - 1ull) << (g * LP))) == 0ull;
    } else return group_none<L>(pred, g);
}

// ---- objectives in NLopt's minimisation form (common.jl:11-36 negated) ----------------------------------------------
// nu: f = 1/2 sum nu_i S_ii + sum c_i exp(lambda_i + nu_i/2) - 1/2 sum log nu_i
// exp and log of the objectives come from the two tables the solve kernels stage into LDS (SolveTabs below; mmm_arith.h: no division, a
// third fewer instructions -- the phase is bound by vector-f64 issue)
struct NuObj {
    double lam, c, Sll; bool act;
    const double* tabs;       // LDS: [exp table | log table]
    template <int L, int LP = 0>
    __device__ __forceinline__ double eval(double x, double& g, const PackCtx& pc = PackCtx{}) const
    {
        const double E = ar_exp_tab(lam + 0.5 * x, tabs);
        g = act ? 0.5 * Sll + 0.5 * c * E - dev_div(1.0, 2.0 * x) : 0.0;
        const double t = act ? 0.5 * x * Sll + c * E - 0.5 * ar_log_tab(x, tabs + MMM_EXPTAB_N) : 0.0;
        return gsum<L, LP>(pc, t);
    }
};

// lambda: f = 1/2 (x-mu)' S (x-mu) - x . sumtheta + sum c_i exp(x_i + nu_i/2)
template <int MKT>      // MKT = sum K when known at compile time (the matrix-vector product unrolls fully), 0 = runtime
struct LamObj {
    double nu, c, sumth, mu; bool act; int l, MK;
    const double* sS;     // [j*MK + i], symmetric
    double* scr;          // group-private LDS, >= MK doubles
    const double* tabs;   // LDS: [exp table | log table]
    template <int L, int LP = 0>
    __device__ __forceinline__ double eval(double x, double& g, const PackCtx& pc = PackCtx{}) const
    {
        const double diff = act ? x - mu : 0.0;
        lds_wave_sync();
        scr[l] = diff;
        lds_wave_sync();
        double Sd = 0.0;
        if (act) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // four independent chains, combined pairwise
            if (MKT) {
                const double* col = sS + l;                     // immediate LDS offsets: j * MKT * 8 bytes
#pragma unroll
                for (int j = 0; j + 3 < MKT; j += 4) {
                    s0 = fma(col[j * MKT], scr[j], s0); s1 = fma(col[(j + 1) * MKT], scr[j + 1], s1);
                    s2 = fma(col[(j + 2) * MKT], scr[j + 2], s2); s3 = fma(col[(j + 3) * MKT], scr[j + 3], s3);
                }
#pragma unroll
                for (int j = MKT & ~3; j < MKT; ++j) s0 = fma(col[j * MKT], scr[j], s0);
            } else {
                int j = 0;
                for (; j + 3 < MK; j += 4) {
                    s0 = fma(sS[j * MK + l], scr[j], s0); s1 = fma(sS[(j + 1) * MK + l], scr[j + 1], s1);
                    s2 = fma(sS[(j + 2) * MK + l], scr[j + 2], s2); s3 = fma(sS[(j + 3) * MK + l], scr[j + 3], s3);
                }
                for (; j < MK; ++j) s0 = fma(sS[j * MK + l], scr[j], s0);
            }
            Sd = (s0 + s1) + (s2 + s3);
        }
        const double E = ar_exp_tab(x + 0.5 * nu, tabs);
        g = act ? Sd - sumth + c * E : 0.0;
        const double t = act ? 0.5 * diff * Sd - x * sumth + c * E : 0.0;
        return gsum<L, LP>(pc, t);
    }
};

// the function tables of the objectives, staged once per block (call from every thread of the block, before a __syncthreads())
__device__ __forceinline__ void stage_solve_tabs(double* tabs)
{
    for (int i = threadIdx.x; i < MMM_EXPTAB_N + MMM_LOGTAB_N; i += blockDim.x) tabs[i] = i < MMM_EXPTAB_N ? g_mmm_exptab[i] : g_mmm_logtab[i - MMM_EXPTAB_N];
}

// What a solve leaves in nev_nu / nev_lam: its number of objective evaluations, bit 30 set when an objective value was not finite in any of
// them (the reference ignores NLopt's return code and never looks for NaN -- MMCTM.jl:141,168 --: such events are COUNTED, mmm_ctm_events),
// the whole negated when the evaluation cap was hit.
#define MMM_NEV_NONFINITE (1 << 30)
__device__ __forceinline__ int nev_code(int nev, bool capped, bool nonfinite)
{
    const int v = nev | (nonfinite ? MMM_NEV_NONFINITE : 0);
    return capped ? -v : v;
}

// NLopt LD_MMA, zero constraints, for the L-lane group of the calling lane (lane l holds coordinate l).  All lanes of the
// wave execute every trip; a finished group keeps its state through selects.  Returns nev_code(evaluations, cap hit, non-finite objective).
template <int L, int LP, class Obj>
__device__ int mma_group(const Obj& obj, bool act, int g, double& x, bool has_lb, double lb, const SolveOpts& o, const PackCtx& pc)
{
    double sigma = 1.0, rho = 1.0;
    double gcur, grad;
    double fbest = obj.template eval<L, LP>(x, grad, pc);
    double xcur = x, xprev = x, xprevprev = x;
    int k = 1, nev = 1;
    bool done = false, capped = false;
    bool nonfin = !isfinite(fbest);
    const int cap = o.max_eval > 0 ? o.max_eval : 2000;
    while (!__all(done)) {
        // closed-form minimiser of the separable approximation (dual problem is trivial for m = 0)
        // NLopt: u = g sigma^2, v = |g| sigma + rho/2, dx = (u/v) / (-1 - sqrt|1 - (u/(v sigma))^2|).  Multiplied through by v:
        // dx = -u / (v + sqrt(v^2 - g^2 sigma^2)), and v^2 - g^2 sigma^2 = rho (|g| sigma + rho/4) exactly -- positive, no cancellation (NLopt's
        // 1 - q^2 loses its digits next to a bound, where |q| -> 1): ONE correctly rounded quotient and one root instead of two quotients and a
        // root (round 4 had three -> two).  The forms agree to rounding; the order-matched CPU checker of the parity tests follows this one,
        // the index-order one keeps NLopt's (MMA_STEP in mmm_twin.c / orc_mma_minimize in mmm_oracle.c)
        const double sigma2 = sigma * sigma;
        const double ags = fabs(grad) * sigma;
        const double v = ags + 0.5 * rho;
        const double gs2 = grad * sigma2;
        double dx = dev_div(-gs2, v + dev_sqrt_pos(rho * (ags + 0.25 * rho)));
        double xc = x + dx;
        // the three clamps by v_max / v_min (one instruction each instead of a compare and two selects): the same value as NLopt's
        // `if (xc < lb) xc = lb; ...` for every finite xc (lo <= hi; a NaN candidate, which only a non-finite objective produces, would be
        // replaced by the bound instead of kept)
        if (has_lb) xc = dev_max_raw(xc, lb);
        xc = dev_min_raw(dev_max_raw(xc, x - 0.9 * sigma), x + 0.9 * sigma);
        if (!act) xc = x;
        dx = xc - x;
        const double dx2 = dx * dx;
        const double denominv = dev_div(1.0, sigma2 - dx2);
        // NLopt: gval += (g sigma^2 dx + v dx^2) / (sigma^2 - dx^2), wval += (dx^2 / 2) / (sigma^2 - dx^2) -- as dx (g sigma^2 + v dx) / (...)
        // with the product g sigma^2 the step has formed already, and the factor 1/2 applied to the SUM (a power of two: the same bits)
        const double gl = act ? (fma(v, dx, gs2) * dx) * denominv : 0.0;
        const double wl = act ? dx2 * denominv : 0.0;
        double gsm = gl, wval = wl;
        gsum2<L, LP>(pc, gsm, wval);
        wval *= 0.5;
        const double gval = fbest + gsm;
        const double fcur = obj.template eval<L, LP>(xc, gcur, pc);
        bool inner_done = false;
        if (!done) {
            ++nev;
            xcur = xc;
            nonfin = nonfin || !isfinite(fcur);
            inner_done = gval >= fcur;
            if (fcur < fbest) { fbest = fcur; x = xc; grad = gcur; }
            if (nev >= cap) { done = true; capped = true; inner_done = false; }
        }
        // rho grows only in a group whose approximation was not conservative; the division is skipped while no group of the wave needs it
        const bool grow = !done && !inner_done && fcur > gval;
        if (__any(grow)) { const double rn = fmin(10.0 * rho, 1.1 * (rho + dev_div(fcur - gval, wval))); rho = grow ? rn : rho; }
        // outer iteration finished in at least one group of this wave: NLopt's x-tolerance test on (xcur, xprev)
        if (__any(inner_done)) {
            const double ad = fabs(xcur - xprev);
            bool stop;
            if (o.xtol_rule == 0) {
                double dn = act ? ad : 0.0, xn = act ? fabs(xcur) : 0.0;
                gsum2<L, LP>(pc, dn, xn);
                stop = (dn < o.xtol_rel * xn) || gnone<L, LP>(act && !(ad < o.xtol_abs), g);
            } else {
                const bool ok = isinf(xprev) ? false
                                              : (ad < o.xtol_abs || ad < o.xtol_rel * (fabs(xcur) + fabs(xprev)) * 0.5 ||
                                                 (o.xtol_rel > 0 && xcur == xprev));
                stop = gnone<L, LP>(act && !ok, g);
            }
            if (inner_done) {
                if (stop) done = true;
                else {
                    rho = fmax(0.1 * rho, 1e-5);
                    if (k > 1) {
                        const double sgn = (xcur - xprev) * (xprev - xprevprev);
                        sigma *= (sgn < 0 ? 0.7 : (sgn > 0 ? 1.2 : 1.0));
                    }
                    ++k;
                    xprevprev = xprev;
                    xprev = xcur;
                }
            }
        }
    }
    return nev_code(nev, capped, nonfin);
}

// ---------------------------------------------------------------------------------------------------------------------
struct CtmEArgs {
    CtmDev c;
    const double* invSigma; const double* mu; const double* expE;     // topic tables used by theta: exp(Eeff)
    const double* lam_in; double* lam_out; double* nu; double* zeta; double* theta;
    double* sumth;          // [D][MK]: written by the theta phase, read by the solve phase
    double* partial;        // [gridDim][GT] (F_SLAB)
    double* aexp;           // wide tables: [D][MK] exp(lambda - max) of the theta phase, for k_ctm_stats_terms
    int* nev_nu; int* nev_lam;   // per document (may be NULL)
    SolveOpts opt;
    int flags;
    const int* active;      // batched launches (grid.y = replicas): per-replica activity flags, may be NULL
    // fused pass (F_SLAB): the theta phase also keeps lambda_{t-1} and the exp table of this pass (theta_t is rebuilt from them on
    // demand) -- it reads both anyway, which saves the copy launch.  Base of replica 0, may be NULL.
    double* lam_keep; double* expE_keep;
};

// PH = 0: zeta / theta / sumtheta / gamma slabs (register-heavy, table- and slab-staged);
// PH = 1: the two LD_MMA solves (few registers, high occupancy: the solves are latency-bound dependent chains)
// OCC (solve phase): 4 waves per SIMD -- 128 VGPRs, with a few spilled values for MK = 10 / 14 -- when the launch has the waves to
// fill them; 3 -- no scratch at all, and no scratch set-up between dispatches -- for small launches (a 560-document fit: +7 %)
// WIDE (theta phase): topic tables too large for LDS (a 1536-term modality, ...): the table is read through L2, no slabs --
// the gamma statistics come from k_ctm_stats_terms, a term-major sweep over posting lists that evaluates theta_kw again from
// the exp(lambda - max) rows this phase leaves in `aexp` (the scheme of the LDA wide path, lda.hip)
// PACK (solve phase, MKT = sum K with 64 % MKT != 0): MKT lanes per document instead of L (packed_sum above)
template <int L, int PH, int MKT = 0, int KMX = 16, int OCC = 4, bool WIDE = false, bool PACK = false>
__global__ __launch_bounds__(PH ? 256 : 512, PH ? OCC : 1) void k_ctm_estep(CtmEArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    static_assert(!PACK || (PH == 1 && MKT > 0 && MMM_WAVE % MKT != 0), "packed groups: solve phase with compile-time sum K");
    constexpr int LG = PACK ? MKT : L;          // lanes per document group
    constexpr int LP = PACK ? MKT : 0;
    constexpr int G = MMM_WAVE / LG;
    const CtmDims& dm = a.c.dm;
    const int MK = MKT ? MKT : dm.MK, M = dm.M, D = dm.D, GT = dm.GT;
    // replica r = blockIdx.y of a batched launch works on the r-th copy of every per-model array.  The kernel arguments are
    // NOT modified in place: that would force the whole struct into scratch and turn its scalar loads into private-memory loads.
    const size_t rep = blockIdx.y;
    if (a.active && !a.active[rep]) return;
    const double* __restrict__ p_invSigma = a.invSigma + rep * MK * MK;
    const double* __restrict__ p_mu = a.mu + rep * MK;
    const double* p_lam_in = a.lam_in + rep * D * MK;
    double* p_lam_out = a.lam_out ? a.lam_out + rep * D * MK : nullptr;
    double* p_nu = a.nu + rep * D * MK;
    double* p_zeta = a.zeta + rep * D * M;
    double* p_sumth = a.sumth + rep * D * MK;
    const double* __restrict__ p_expE = a.expE ? a.expE + rep * GT : nullptr;
    double* p_partial = a.partial ? a.partial + rep * gridDim.x * GT : nullptr;
    int* p_nev_nu = a.nev_nu ? a.nev_nu + rep * D : nullptr;
    int* p_nev_lam = a.nev_lam ? a.nev_lam + rep * D : nullptr;
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / LG, l = lane % LG;
    const bool ingrp = g < G;                          // packed groups leave 64 % MKT spare lanes
    PackCtx pc{};
    if constexpr (PACK) pc = pack_ctx<MKT>(lane);
    const int flags = a.flags;
    // PH 1: [MK*MK invSigma | MK mu | scratch];  PH 0: [scratch | GT table | NW*GT slabs]
    constexpr int SCRW = PACK ? (G + 1) * 2 * LG : 2 * MMM_WAVE;      // scratch doubles per wave: [G][2 LG] (+ one dummy group for the spare lanes)
    double* sScr = smem;                               // [NW][SCRW]
    double* sS = sScr + (size_t)NW * SCRW;             // [MK*MK]   (PH 1)
    double* sMu = sS + MK * MK;                        // [MK]      (PH 1)
    // the modality-major sweep of the fused pass needs one a_k row per group: half the scratch (BASELINE config 4: 82,400 -> 78,304 bytes
    // per block, which is what lets two blocks share a CU's 160 KB)
    const bool slabpass = PH == 0 && !WIDE && (flags & F_SLAB);
    double* sB = sScr + (size_t)NW * (slabpass ? MMM_WAVE : SCRW);             // [GT]      (PH 0)
    double* sSlab = sB + GT;                           // [NW][GT]  (PH 0, F_SLAB)
    const double* sTabs = nullptr;                     // exp | log tables of the objectives (PH 1)
    if constexpr (PH == 1) {
        __shared__ __attribute__((aligned(16))) double s_tabs[MMM_EXPTAB_N + MMM_LOGTAB_N];
        stage_solve_tabs(s_tabs);
        sTabs = s_tabs;
    }
    if (PH == 1) {
        for (int i = tid; i < MK * MK; i += blockDim.x) sS[i] = p_invSigma[i];
        for (int i = tid; i < MK; i += blockDim.x) sMu[i] = p_mu[i];
    } else {
        if (!WIDE && (flags & F_THETA_COMPUTE)) for (int i = tid; i < GT; i += blockDim.x) sB[i] = p_expE[i];
    }
    __syncthreads();
    // ---- fused pass, theta phase with gamma statistics (F_SLAB): MODALITY-MAJOR.  A wave's slab holds one modality at a time
    // (max_m K_m V_m doubles instead of sum_m K_m V_m: config 4 115 -> 76 KB of LDS per block, two blocks per CU instead of one); the
    // block sweeps its documents once per modality and flushes the slabs in between.  Every statistic receives its addends in the
    // same order as in a document-major sweep (documents in step order, lanes ascending), so the sums keep their bits.
    if constexpr (PH == 0 && !WIDE) {
        if (flags & F_SLAB) {
            int slabn = 0;
            for (int m = 0; m < M; ++m) slabn = max(slabn, dm.K[m] * dm.V[m]);
            double* myslab = sSlab + (size_t)wid * slabn;
            double* scr = sScr + ((size_t)wid * G + g) * L;          // a_k of the group's document
            int ml = 0;
            for (int m = 0; m < M; ++m) if (l >= dm.koff[m] && l < dm.koff[m + 1]) ml = m;
            if (a.expE_keep && blockIdx.x == 0) for (int i = tid; i < GT; i += blockDim.x) a.expE_keep[rep * GT + i] = sB[i];
            for (int m = 0; m < M; ++m) {
                const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m];
                for (int i = lane; i < Km * Vm; i += MMM_WAVE) myslab[i] = 0.0;
                lds_wave_sync();
                const double* tb = sB + dm.goff[m];
                const int64_t* dp = a.c.doc_ptr + (size_t)m * (D + 1);
                for (int base = (blockIdx.x * NW + wid) * G; base < D; base += gridDim.x * NW * G) {
                    const int d = base + g;
                    const bool valid = d < D;
                    const bool act = valid && l < MK, mine = act && ml == m;
                    const double lam = act ? p_lam_in[(size_t)d * MK + l] : 0.0;
                    if (m == 0 && a.lam_keep && act) a.lam_keep[(rep * D + d) * MK + l] = lam;
                    if (m == 0 && (flags & F_ZETA)) {       // update_ζ! (MMCTM.jl:172-181), once per document
                        const double nu = act ? p_nu[(size_t)d * MK + l] : 1.0;
                        const double E = act ? ar_exp(lam + 0.5 * nu) : 0.0;
                        for (int q = 0; q < M; ++q) {
                            const double zm = group_sum<L>((act && ml == q) ? E : 0.0);
                            if (valid && l == q) p_zeta[(size_t)d * M + q] = zm;
                        }
                    }
                    const double mx = group_max<L>(mine ? lam : -1e300);
                    lds_wave_sync();
                    scr[l] = mine ? ar_exp(lam - mx) : 0.0;
                    lds_wave_sync();
                    const int64_t start = valid ? dp[d] : 0;
                    const int W = valid ? (int)(dp[d + 1] - start) : 0;
                    double av[KMX], acc[KMX];
#pragma unroll
                    for (int k = 0; k < KMX; ++k) { av[k] = (k < Km) ? scr[off + k] : 0.0; acc[k] = 0.0; }
                    // the document's (term,count) pairs of this modality: the first PRE chunks are requested together, before the
                    // first chunk computes (one memory latency per document and modality instead of one per chunk)
                    // (32-lane groups: three chunks cover a 96-term document -- cfg 4 theta phase 138 -> 131 us; with 16-lane groups the six
                    // chunk registers cost the fourth wave per SIMD and the phase got slower, 64 -> 83 us at cfg 5: one chunk there)
                    constexpr int PRE = L >= 32 ? (96 / L > 0 ? 96 / L : 1) : 1;
                    int2 tcp[PRE];
#pragma unroll
                    for (int j = 0; j < PRE; ++j) { const int w = j * L + l; tcp[j] = (w < W) ? a.c.tc[start + w] : make_int2(-1, 0); }
                    int j = 0;
                    for (int w0 = 0; __any(w0 < W); w0 += L, ++j) {
                        const int w = w0 + l;
                        int2 tcv = tcp[0];
#pragma unroll
                        for (int q = 1; q < PRE; ++q) tcv = (j == q) ? tcp[q] : tcv;
                        if (j >= PRE) tcv = (w < W) ? a.c.tc[start + w] : make_int2(-1, 0);
                        const bool aw = tcv.x >= 0;
                        tcv.x = aw ? tcv.x : 0;
                        const double n = (double)tcv.y;
                        double e[KMX], s = 0.0;
#pragma unroll
                        for (int k = 0; k < KMX; ++k) { e[k] = (k < Km) ? av[k] * tb[k * Vm + tcv.x] : 0.0; s += e[k]; }
                        // s = sum_k a_k exp(Elnphi_kv): one a_k is 1 and Elnphi >= psi(alpha) - psi(sum gamma), so s is far inside the normal
                        // range, where dev_div is the correctly rounded quotient (8 instructions instead of the ~25 of the general sequence)
                        const double inv = aw ? dev_div(1.0, s) : 0.0;
                        const double r = n * inv;
                        double pn[KMX];
#pragma unroll
                        for (int k = 0; k < KMX; ++k) { pn[k] = e[k] * r; acc[k] += pn[k]; }
                        if (aw) {
#pragma unroll
                            for (int k = 0; k < KMX; ++k) if (k < Km) unsafeAtomicAdd(&myslab[k * Vm + tcv.x], pn[k]);
                        }
                    }
                    double st = 0.0;
#pragma unroll
                    for (int k = 0; k < KMX; ++k) {
                        if (k < Km) { const double tot = group_sum<L>(acc[k]); if (l == off + k) st = tot; }
                    }
                    if (mine) p_sumth[(size_t)d * MK + l] = st;
                }
                __syncthreads();
                double* out = p_partial + (size_t)blockIdx.x * GT + dm.goff[m];
                for (int i = tid; i < Km * Vm; i += blockDim.x) {
                    double s = 0.0;
                    for (int w = 0; w < NW; ++w) s += sSlab[(size_t)w * slabn + i];
                    out[i] = s;
                }
                __syncthreads();
            }
            return;
        }
    }
    double* slab = sSlab + (size_t)wid * GT;
    (void)slab;
    double* scrA = sScr + (size_t)wid * SCRW + (size_t)g * 2 * LG;   // a_k values (spare lanes of a packed wave: the dummy group g = G)
    double* scrD = scrA + LG;                              // lambda-objective differences
    int mod_l = 0;
    for (int m = 0; m < M; ++m) if (l >= dm.koff[m] && l < dm.koff[m + 1]) mod_l = m;

    for (int base = (blockIdx.x * NW + wid) * G; base < D; base += gridDim.x * NW * G) {
        const int d = base + g;
        const bool valid = ingrp && d < D;
        const bool act = valid && l < MK;
        double lam = act ? p_lam_in[(size_t)d * MK + l] : 0.0;
        double nu = act ? p_nu[(size_t)d * MK + l] : 1.0;
        const double Nl = act ? a.c.Ndm[(size_t)d * M + mod_l] : 0.0;
        // ---- update_ζ! (MMCTM.jl:172-181) -------------------------------------------------------------------------
        double zl = 1.0;
        if (PH == 0 && (flags & F_ZETA)) {
            const double E = act ? ar_exp(lam + 0.5 * nu) : 0.0;
            for (int m = 0; m < M; ++m) {
                const double zm = group_sum<L>((act && mod_l == m) ? E : 0.0);
                if (mod_l == m) zl = zm;
                if (valid && l == m) p_zeta[(size_t)d * M + m] = zm;
            }
        } else if (act) zl = p_zeta[(size_t)d * M + mod_l];
        const double cl = Nl / zl;                                   // Ndivζ (MMCTM.jl:119-125)
        // ---- update_θ! (MMCTM.jl:183-198) and sumθ (MMCTM.jl:110-117) ------------------------------------------------
        double sumth = 0.0;
        if (PH == 1) sumth = act ? p_sumth[(size_t)d * MK + l] : 0.0;
        if (PH == 0 && (flags & (F_THETA_COMPUTE | F_THETA_STORED))) {
            double mx = 0.0;
            for (int m = 0; m < M; ++m) {
                const double mm = group_max<L>((act && mod_l == m) ? lam : -1e300);
                if (mod_l == m) mx = mm;
            }
            lds_wave_sync();
            scrA[l] = act ? ar_exp(lam - mx) : 0.0;
            if (WIDE && act && a.aexp && (flags & F_SLAB)) a.aexp[(rep * D + d) * MK + l] = scrA[l];
            lds_wave_sync();
            for (int m = 0; m < M; ++m) {
                const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m];
                const double* tb = WIDE ? p_expE + dm.goff[m] : sB + dm.goff[m];
                const int64_t* dp = a.c.doc_ptr + (size_t)m * (D + 1);
                const int64_t start = valid ? dp[d] : 0;
                const int W = valid ? (int)(dp[d + 1] - start) : 0;
                double av[KMX], acc[KMX];
#pragma unroll
                for (int k = 0; k < KMX; ++k) { av[k] = (k < Km) ? scrA[off + k] : 0.0; acc[k] = 0.0; }
                for (int w0 = 0; __any(w0 < W); w0 += L) {
                    const int w = w0 + l;
                    const bool aw = w < W;
                    const int2 tcv = aw ? a.c.tc[start + w] : make_int2(0, 0);
                    const double n = (double)tcv.y;
                    double* th = a.theta ? a.theta + dm.toff[m] + (size_t)(start + w - dm.estart[m]) * Km : nullptr;
                    double e[KMX], r, inv = 0.0;
                    if (flags & F_THETA_COMPUTE) {
                        double s = 0.0;
#pragma unroll
                        for (int k = 0; k < KMX; ++k) { e[k] = (k < Km) ? av[k] * tb[k * Vm + tcv.x] : 0.0; s += e[k]; }
                        inv = aw ? 1.0 / s : 0.0;
                        r = n * inv;
                    } else {
#pragma unroll
                        for (int k = 0; k < KMX; ++k) e[k] = (aw && k < Km) ? th[k] : 0.0;
                        r = n;
                    }
#pragma unroll
                    for (int k = 0; k < KMX; ++k) {
                        const double pn = e[k] * r;
                        acc[k] += pn;
                        if (aw && k < Km) {
                            if (flags & F_THETA_STORE) th[k] = e[k] * inv;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < KMX; ++k) {
                    if (k < Km) { const double tot = group_sum<L>(acc[k]); if (l == off + k) sumth = tot; }
                }
            }
        }
        if (PH == 0) { if (act) p_sumth[(size_t)d * MK + l] = sumth; continue; }
        SolveOpts o = a.opt;
        // ---- update_ν! (MMCTM.jl:156-170): LD_MMA, lower bound 1e-7, from the current ν, with the old λ ----------------
        if (flags & F_NU) {
            NuObj obj{lam, cl, act ? sS[l * MK + l] : 1.0, act, sTabs};
            const int nev = mma_group<L, LP>(obj, act, g, nu, true, o.nu_lower, o, pc);
            if (act) p_nu[(size_t)d * MK + l] = nu;
            if (p_nev_nu && valid && l == 0) p_nev_nu[d] = nev;
        }
        // ---- update_λ! (MMCTM.jl:127-143): LD_MMA, unbounded, with the new ν ---------------------------------------------
        if (flags & F_LAMBDA) {
            LamObj<MKT> obj{nu, cl, sumth, act ? sMu[l] : 0.0, act, l, MK, sS, scrD, sTabs};
            const int nev = mma_group<L, LP>(obj, act, g, lam, false, 0.0, o, pc);
            if (act) p_lam_out[(size_t)d * MK + l] = lam;
            if (p_nev_lam && valid && l == 0) p_nev_lam[d] = nev;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Theta phase of the fused pass over ROWS OF COUNTS (round 3; dense corpora -- the shipped BRCA tables and the synthetic configurations
// are 85-100 % dense): one launch per modality; a document takes 16 lanes whatever sum K is (a modality has at most 16 topics here),
// four documents per wave step; lane l owns the terms l, 16 + l, ... of EVERY document it meets, so the gamma statistics sum_d theta_kv n_dv
// of its terms stay in registers for the whole launch (SL KMX doubles) and reach the wave's slab once, at the end -- no LDS atomics, no
// (term, count) loads: 2 bytes per term slot.  The scheme of k_lda_estep_dense (lda.hip) with the CTM's prologue (zeta, exp(lambda -
// max)); same formulas and the same per-element operations as the theta phase of k_ctm_estep (MMCTM.jl:172-198, 110-117), other
// association of the sums (the order-matched CPU restatement of the parity tests mirrors it: tw_theta_dense).
// LDS: [16 SL][KMX] table, term-major | [NW][16 SL][KMX] slabs, term-major | [NW][4][KMX] a_k | [NW][64][KMX] sum-theta scratch
template <class T> __device__ __forceinline__ T* at_byte(T* base, unsigned off) { return (T*)((char*)base + off); }   // uniform base + 32-bit lane offset: one VGPR per address

template <int KMX, int SL>
__global__ __launch_bounds__(512, 2) void k_ctm_theta_dense(CtmEArgs a, int m, const unsigned short* __restrict__ rows)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int L = 16, G = MMM_WAVE / L, Vp = L * SL;
    const CtmDims& dm = a.c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D, GT = dm.GT;
    const size_t rep = blockIdx.y;
    if (a.active && !a.active[rep]) return;
    const double* p_lam_in = a.lam_in + rep * D * MK;
    const double* p_nu = a.nu + rep * D * MK;
    double* p_zeta = a.zeta + rep * D * M;
    double* p_sumth = a.sumth + rep * D * MK;
    const double* __restrict__ p_expE = a.expE + rep * GT;
    double* p_partial = a.partial + rep * gridDim.x * GT;
    const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m];
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / L, l = lane % L;
    const int KV = Km * Vm;
    double* sT = smem;                                   // [Vp][KMX]; rows v >= Vm hold 1 (their counts are 0), topics k >= Km hold 0
    double* sSlab = sT + (size_t)Vp * KMX;               // [NW][Vp][KMX], term-major like the table (written once, in the epilogue)
    double* sA = sSlab + (size_t)NW * Vp * KMX;          // [NW][G][KMX]
    double* sR = sA + (size_t)NW * G * KMX;              // [NW][64][KMX]
    double* slab = sSlab + (size_t)wid * Vp * KMX;
    double* myA = sA + ((size_t)wid * G + g) * KMX;
    double* myR = sR + (size_t)wid * MMM_WAVE * KMX;
    const double* __restrict__ tbg = p_expE + dm.goff[m];
    for (int i = tid; i < Vp * KMX; i += blockDim.x) {
        const int v = i / KMX, k = i % KMX;
        sT[i] = (k < Km) ? (v < Vm ? tbg[(size_t)k * Vm + v] : 1.0) : 0.0;
    }
    if (a.expE_keep && blockIdx.x == 0 && m == 0) for (int i = tid; i < GT; i += blockDim.x) a.expE_keep[rep * GT + i] = p_expE[i];
    __syncthreads();
    double st[SL][KMX];
#pragma unroll
    for (int q = 0; q < SL; ++q)
#pragma unroll
        for (int k = 0; k < KMX; ++k) st[q][k] = 0.0;
    const int flags = a.flags;
    const int stride = gridDim.x * NW * G;
    int base = (blockIdx.x * NW + wid) * G;
    // The next step's lambda (nu) row and counts are requested a step ahead and must STAY in flight across the term phase: vmcnt counts in
    // order, so nothing between a request and its use may wait for memory (k_lda_estep_dense, lda.hip, has the measurements).  Hence: loads
    // are unconditional (clamped document and topic index; the masks are applied when the values are taken over), a lane's part of a row
    // of counts is ONE load of NC 32-bit words (lane-major rows), addresses are a uniform base + a 32-bit offset (D sum K 8 bytes < 4 GB,
    // checked at create), the request follows the prologue, and the values are taken over before the step's last (lane-conditional) store.
    // (PIN: where the registers allow it the first use of the loaded values is pinned behind the term phase; the 60-statistic builds
    // would spill the statistics for it -- there the compiler takes the values over early and the block's other wave covers the wait)
    constexpr int NC = (SL + 1) / 2, SLs = 2 * NC;
    constexpr bool PIN = KMX * SL <= 32;
    const int lk = l < Km ? l : Km - 1;
    const bool zeta = (flags & F_ZETA) != 0;
    int d = base + g;
    bool valid = d < D;
    unsigned dl = valid ? (unsigned)d : 0u;
    unsigned c[NC], cn[NC];
    auto request = [&](unsigned dd, double& lam_o, double& nu_o, unsigned* o) {
        const unsigned ob = (dd * (unsigned)MK + (unsigned)(off + lk)) * 8u;
        lam_o = *at_byte(p_lam_in, ob);
        if (zeta) nu_o = *at_byte(p_nu, ob);
        const unsigned* row = at_byte((const unsigned*)rows, (dd * (unsigned)(16 * SLs) + l * SLs) * 2u);
#pragma unroll
        for (int j = 0; j < NC; ++j) o[j] = row[j];
    };
    double lam, nu = 1.0;
    request(dl, lam, nu, cn);
    lam = (valid && l < Km) ? lam : 0.0;
    nu = (zeta && valid && l < Km) ? nu : 1.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) c[j] = valid ? cn[j] : 0u;
    for (; base < D; base += stride) {
        const int dn = d + stride;
        const bool validn = base + stride < D && dn < D;
        const unsigned dnl = validn ? (unsigned)dn : dl;
        const bool act = valid && l < Km;
        if (a.lam_keep && act) a.lam_keep[(rep * D + d) * MK + off + l] = lam;
        if (flags & F_ZETA) {       // update_ζ! (MMCTM.jl:172-181)
            const double zm = group_sum<L>(act ? ar_exp(lam + 0.5 * nu) : 0.0);
            if (valid && l == 0) *at_byte(p_zeta, (dl * (unsigned)M + (unsigned)m) * 8u) = zm;
        }
        const double mx = group_max<L>(act ? lam : -1e300);
        lds_wave_sync();
        if (l < KMX) myA[l] = act ? ar_exp(lam - mx) : 0.0;
        lds_wave_sync();
        double lamn, nun = 1.0;
        request(dnl, lamn, nun, cn);
        double av[KMX], acc[KMX];
#pragma unroll
        for (int k = 0; k < KMX; ++k) { av[k] = myA[k]; acc[k] = 0.0; }
        // theta_kv n_v (MMCTM.jl:183-198) for the lane's SL terms: e_k = a_k B_kv, s = sum_k e_k, 1/s correctly rounded, the sums by fma
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            const double* tb = sT + (size_t)(q * L + l) * KMX;
            // (s in topic order, as the slab kernel and the on-demand rebuild of theta form it: theta_kv = e_k / s must be the same bits
            // wherever it is evaluated; two interleaved chains were 3 us faster at cfg 5 and broke exactly that)
            double e[KMX], s = 0.0;
#pragma unroll
            for (int k = 0; k < KMX; ++k) { e[k] = av[k] * tb[k]; s += e[k]; }
            // (a document group beyond the corpus has a = 0, hence s = 0: 0 x (1 / 0) must not reach the statistics -- v_max with the
            // smallest normal leaves every real s as it is)
            const unsigned cq = (q & 1) ? c[q / 2] >> 16 : c[q / 2] & 0xffffu;
            const double r = (double)cq * dev_div(1.0, dev_max_raw(s, 2.2250738585072014e-308));
#pragma unroll
            for (int k = 0; k < KMX; ++k) { acc[k] = fma(e[k], r, acc[k]); st[q][k] = fma(e[k], r, st[q][k]); }
#pragma unroll
            for (int k = 0; k < KMX; ++k) asm volatile("" : "+v"(st[q][k]));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // the requested values are taken over here (the empty statements are the first use of the loaded registers and stay behind the term
        // phase's own)
        if (PIN) {
            asm volatile("" : "+v"(lamn) :: "memory");
            if (zeta) asm volatile("" : "+v"(nun) :: "memory");
#pragma unroll
            for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(cn[j]) :: "memory");
        }
        const bool valid_now = valid;
        const unsigned dl_now = dl;
        lam = (validn && l < Km) ? lamn : 0.0;
        nu = (zeta && validn && l < Km) ? nun : 1.0;
#pragma unroll
        for (int j = 0; j < NC; ++j) c[j] = validn ? cn[j] : 0u;
        if (PIN) {
#pragma unroll
            for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(c[j]));
            __builtin_amdgcn_sched_barrier(0);
        }
        // sumθ_k (MMCTM.jl:110-117): the lanes' sums meet in LDS, lane k of the group adds its column of the 16 lanes' values
#pragma unroll
        for (int k = 0; k < KMX; ++k) myR[(size_t)lane * KMX + k] = acc[k];
        lds_wave_sync();
        if (l < Km) {
            const double* col = myR + (size_t)(g * L) * KMX + l;
            double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
#pragma unroll
            for (int j = 0; j < L; j += 4) { r0 += col[j * KMX]; r1 += col[(j + 1) * KMX]; r2 += col[(j + 2) * KMX]; r3 += col[(j + 3) * KMX]; }
            if (valid_now) *at_byte(p_sumth, (dl_now * (unsigned)MK + (unsigned)(off + l)) * 8u) = (r0 + r1) + (r2 + r3);
        }
        d = dn; valid = validn; dl = dnl;
        lds_wave_sync();
    }
    // the wave's statistics: the four document groups' registers are added across the rows of the wave (rows_sum4: (g0 + g2) + (g1 + g3), no
    // LDS) and the first group's lanes store them -- the slab is term-major with padded bounds like the table, written once (no zero
    // fill, no read-modify-write; 16-byte pairs at compile-time offsets; k_lda_estep_dense has the measurements); then the block's waves
    // in order
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        double* sl = slab + (size_t)(q * L + l) * KMX;
        double t[KMX];
#pragma unroll
        for (int k = 0; k < KMX; ++k) t[k] = rows_sum4(st[q][k]);
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < KMX; ++k) sl[k] = t[k];
        }
    }
    __syncthreads();
    double* out = p_partial + (size_t)blockIdx.x * GT + dm.goff[m];
    for (int i = tid; i < KV; i += blockDim.x) {
        const int kk = i / Vm, v = i - kk * Vm;
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += sSlab[((size_t)w * Vp + v) * KMX + kk];
        out[i] = s;
    }
}

// =====================================================================================================================
// Solve phase, several coordinates per lane.  The solve phase is f64-VALU bound (PMC: the vector pipes are ~100 % busy), and with
// one coordinate per lane most of a trip's instructions are not arithmetic on coordinates: five lane-butterfly sums, the scalars of
// the LD_MMA state machine replicated in every lane, the padding lanes.  Here a document takes LPD lanes (2 or 4) with CPL = sum K /
// LPD coordinates each: 16 or 32 documents per wave instead of 2-4, the per-document scalars are paid once per LPD lanes, a sum over
// the document is CPL-1 local additions and log2(LPD) quad-permute stages.  Same algorithm, same formulas as mma_group; the sums
// are associated as: each lane adds its coordinates in index order (from 0), the lanes' partial sums go through the quad butterfly.
// Coordinate i of a document sits in lane i / CPL, slot i % CPL.
template <int LPD>
__device__ __forceinline__ double qsum(double v)
{
    v += dpp_mov_f64<0xB1>(v);                      // quad_perm [1,0,3,2]
    if (LPD >= 4) v += dpp_mov_f64<0x4E>(v);        // quad_perm [2,3,0,1]
    if (LPD >= 8) v += dpp_mov_f64<0x141>(v);       // row_half_mirror: the other quad of the 8-lane half row
    if (LPD >= 16) v += dpp_mov_f64<0x140>(v);      // row_mirror
    if (LPD >= 32) v += __shfl_xor(v, 16, MMM_WAVE);
    return v;
}

template <int LPD>
__device__ __forceinline__ bool qnone(bool pred, int lane)
{
    const unsigned long long b = __ballot(pred);
    return ((b >> (lane & ~(LPD - 1))) & ((LPD >= 64 ? 0ull : (1ull << (LPD & 63))) - 1ull)) == 0ull;
}

// invSigma in LDS for this layout: row j (the factor's index), the document's coordinates padded per lane to CPLP = CPL rounded up
// to even, so that a lane reads its CPL entries of a row as 16-byte pairs: sS[j * (LPD * CPLP) + l * CPLP + c]
// LPD lanes per document of which ACT = sum K / CPL hold coordinates (sum K = 28: 8 lanes, 7 of them with 4 coordinates each)
template <int MKT, int LPD>
struct CplGeom {
    static constexpr int CPL = (MKT + LPD - 1) / LPD, ACT = MKT / CPL, CPLP = CPL == 1 ? 1 : (CPL + 1) & ~1, ROW = LPD * CPLP, G = MMM_WAVE / LPD;
    static_assert(MKT % CPL == 0 && ACT <= LPD, "sum K must be a whole number of lanes of CPL coordinates");
};

// per-document inputs of a solve (pointers of the launch's replica)
struct CplDocs {
    const double* lam_in; double* lam_out; double* nu; const double* zeta; const double* sumth; const double* Ndm;
    int M;
};

template <int MKT, int LPD, bool SB>
struct NuObjC {
    using Gm = CplGeom<MKT, LPD>;
    double lam[Gm::CPL], c[Gm::CPL], Sll[Gm::CPL];
    int modpack, l;       // the modality of coordinate q of this lane in bits [4q, 4q + 4)
    int uni;              // wave-uniform: bit q = every lane's coordinates q and q - 1 share their modality
    bool lane_on, on;      // lane_on: the lane holds coordinates (l < ACT); on: ... of a document
    const double* tabs;    // LDS: [exp table | log table]
    // start point and constants of document d (d < 0: an empty slot).  Lanes that are not `on` keep x = 0 and contribute exact zeros.
    __device__ __forceinline__ void load(const CplDocs& dc, int d, double (&x)[Gm::CPL])
    {
        if (!lane_on) d = -1;
        on = d >= 0;
        const size_t row = (size_t)(d < 0 ? 0 : d) * MKT + l * Gm::CPL;
#pragma unroll
        for (int q = 0; q < Gm::CPL; ++q) {
            x[q] = d < 0 ? 0.0 : dc.nu[row + q];
            lam[q] = d < 0 ? 0.0 : dc.lam_in[row + q];
            // Ndivζ (MMCTM.jl:119-125): one quotient per modality -- where every lane's coordinate q lies in the modality of its coordinate
            // q - 1 (`uni`, wave-uniform: always, for one modality) the quotient is taken over; dev_div = the IEEE quotient for these operands
            if (q > 0 && ((uni >> q) & 1)) { c[q] = c[q - 1]; continue; }
            const double Nl = d < 0 ? 0.0 : dc.Ndm[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)], zl = d < 0 ? 1.0 : dc.zeta[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)];
            c[q] = dev_div(Nl, zl);
        }
    }
    __device__ __forceinline__ void store(const CplDocs& dc, int d, const double (&x)[Gm::CPL]) const
    {
        const size_t row = (size_t)d * MKT + l * Gm::CPL;
        if (lane_on) {
#pragma unroll
            for (int q = 0; q < Gm::CPL; ++q) dc.nu[row + q] = x[q];
        }
    }
    __device__ __forceinline__ double eval(const double (&x)[Gm::CPL], double (&g)[Gm::CPL]) const
    {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < Gm::CPL; ++q) {
            const double E = ar_exp_tab(lam[q] + 0.5 * x[q], tabs);
            const bool msk = (Gm::ACT < LPD) ? on : true;        // only layouts with idle lanes need the mask
            const double gq = 0.5 * Sll[q] + 0.5 * c[q] * E - dev_div(1.0, 2.0 * x[q]);
            const double tq = 0.5 * x[q] * Sll[q] + c[q] * E - 0.5 * ar_log_tab(x[q], tabs + MMM_EXPTAB_N);
            g[q] = msk ? gq : 0.0;
            s += msk ? tq : 0.0;
            if (SB) __builtin_amdgcn_sched_barrier(0);
        }
        return qsum<LPD>(s);
    }
};

template <int MKT, int LPD, bool SB>
struct LamObjC {
    using Gm = CplGeom<MKT, LPD>;
    double nu[Gm::CPL], c[Gm::CPL], sumth[Gm::CPL];
    const double* smu;    // mu in LDS, [LPD * CPL] in the lane layout (0 for lanes without coordinates): CPL registers fewer than a copy per lane
    int modpack, l;       // the modality of coordinate q of this lane in bits [4q, 4q + 4)
    int uni;              // (see NuObjC)
    bool lane_on, on;
    const double* sS;     // padded layout above
    double* scr;          // group-private LDS, MKT doubles (+ pad): the differences x - mu of the whole document
    const double* tabs;   // LDS: [exp table | log table]
    __device__ __forceinline__ void load(const CplDocs& dc, int d, double (&x)[Gm::CPL])
    {
        if (!lane_on) d = -1;
        on = d >= 0;
        const size_t row = (size_t)(d < 0 ? 0 : d) * MKT + l * Gm::CPL;
#pragma unroll
        for (int q = 0; q < Gm::CPL; ++q) {
            x[q] = d < 0 ? 0.0 : dc.lam_in[row + q];
            nu[q] = d < 0 ? 1.0 : dc.nu[row + q];
            sumth[q] = d < 0 ? 0.0 : dc.sumth[row + q];
            if (q > 0 && ((uni >> q) & 1)) { c[q] = c[q - 1]; continue; }      // (see NuObjC::load)
            const double Nl = d < 0 ? 0.0 : dc.Ndm[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)], zl = d < 0 ? 1.0 : dc.zeta[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)];
            c[q] = dev_div(Nl, zl);
        }
    }
    __device__ __forceinline__ void store(const CplDocs& dc, int d, const double (&x)[Gm::CPL]) const
    {
        const size_t row = (size_t)d * MKT + l * Gm::CPL;
        if (lane_on) {
#pragma unroll
            for (int q = 0; q < Gm::CPL; ++q) dc.lam_out[row + q] = x[q];
        }
    }
    __device__ __forceinline__ double eval(const double (&x)[Gm::CPL], double (&g)[Gm::CPL]) const
    {
        double diff[Gm::CPL];
        lds_wave_sync();
#pragma unroll
        for (int q = 0; q < Gm::CPL; ++q) { diff[q] = x[q] - smu[l * Gm::CPL + q]; if (lane_on) scr[l * Gm::CPL + q] = diff[q]; }
        lds_wave_sync();
        // Sd_i = sum_j S_ij diff_j with four chains over j, combined pairwise (the association of LamObj::eval)
        double s0[Gm::CPL], s1[Gm::CPL], s2[Gm::CPL], s3[Gm::CPL];
#pragma unroll
        for (int q = 0; q < Gm::CPL; ++q) { s0[q] = 0.0; s1[q] = 0.0; s2[q] = 0.0; s3[q] = 0.0; }
        const double* row = sS + l * Gm::CPLP;
#pragma unroll
        for (int j = 0; j + 3 < MKT; j += 4) {
            const double d0 = scr[j], d1 = scr[j + 1], d2 = scr[j + 2], d3 = scr[j + 3];
#pragma unroll
            for (int q = 0; q < Gm::CPL; ++q) {
                s0[q] = fma(row[j * Gm::ROW + q], d0, s0[q]); s1[q] = fma(row[(j + 1) * Gm::ROW + q], d1, s1[q]);
                s2[q] = fma(row[(j + 2) * Gm::ROW + q], d2, s2[q]); s3[q] = fma(row[(j + 3) * Gm::ROW + q], d3, s3[q]);
            }
        }
#pragma unroll
        for (int j = MKT & ~3; j < MKT; ++j) {
            const double dj = scr[j];
#pragma unroll
            for (int q = 0; q < Gm::CPL; ++q) s0[q] = fma(row[j * Gm::ROW + q], dj, s0[q]);
        }
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < Gm::CPL; ++q) {
            const double Sd = (s0[q] + s1[q]) + (s2[q] + s3[q]);
            const double E = ar_exp_tab(x[q] + 0.5 * nu[q], tabs);
            const bool msk = (Gm::ACT < LPD) ? on : true;
            const double gq = Sd - sumth[q] + c[q] * E;
            const double tq = 0.5 * diff[q] * Sd - x[q] * sumth[q] + c[q] * E;
            g[q] = msk ? gq : 0.0;
            s += msk ? tq : 0.0;
            if (SB) __builtin_amdgcn_sched_barrier(0);
        }
        return qsum<LPD>(s);
    }
};

// NLopt LD_MMA, zero constraints (the algorithm of mma_group; statement: DESIGN.md "MMA" and SURVEY.md section 7) for the documents
// [r0, r1) of the calling wave, LPD lanes per document, 64 / LPD document SLOTS.  A slot whose solve stops takes the next
// document of the range at once (documents finish after very different numbers of evaluations: in lock step a wave would run
// to its slowest document with the other slots idle -- 1.3-2x the mean at 32 slots); a new document's first evaluation f(x0)
// rides in the common trip with the candidate point = x0.  Every document goes through exactly the operations of mma_group,
// whatever its slot and its neighbours.
// The order in which a wave's slots take the documents of its range: by the evaluation counts of the PREVIOUS pass' solve (nev_out still
// holds them when the phase starts), longest first, ties in index order -- s_perm[p] = offset in the range of the p-th document taken.
// In lock step a wave runs to its last solve with the other slots idle; the long solves started first leave the short ones for the end
// (the counts of consecutive passes correlate at 0.5-0.7 for the lambda solves of config 5, not at all for the nu solves:
// tools/sim_solve_schedule.py).  A document's solve does not depend on its slot or its neighbours: not a bit changes.  n <= 64.
__device__ __forceinline__ void order_range(const int* prev, int r0, int n, int lane, int* s_perm)
{
    const int pv = lane < n ? prev[r0 + lane] : 0;
    const int key = lane < n ? ((pv < 0 ? -pv : pv) & (MMM_NEV_NONFINITE - 1)) : -1;
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const int kj = __builtin_amdgcn_readlane(key, j);
        rank += (kj > key || (kj == key && j < lane)) ? 1 : 0;
    }
    lds_wave_sync();
    if (lane < n) s_perm[rank] = lane;
    lds_wave_sync();
}

template <int MKT, int LPD, bool SB, class Obj>
__device__ __forceinline__ void solve_range(Obj& obj, const CplDocs& dc, int r0, int r1, int lane, bool has_lb, double lb, const SolveOpts& o, int* nev_out,
                                            bool by_count, int* s_perm)
{
    constexpr int CPL = CplGeom<MKT, LPD>::CPL, G = MMM_WAVE / LPD;
    const int g = lane / LPD, l = lane % LPD;
    const int n = r1 - r0;
    const bool ord = by_count && nev_out != nullptr && n > G && n <= MMM_WAVE;      // (n <= G: every document has a slot at once)
    if (ord) order_range(nev_out, r0, n, lane, s_perm);
    auto doc_at = [&](int p) { return r0 + (ord ? s_perm[p < n ? p : 0] : p); };
    int next = G;
    bool have = g < n, fresh = true;
    int d = doc_at(g);
    double x[CPL], sigma[CPL], grad[CPL], gcur[CPL], xcur[CPL], xprev[CPL];
    // NLopt's sigma update looks at the SIGN of (xcur - xprev) (xprev - xprevprev).  Only the sign of the older step is kept (+1 / 0 / -1 as a
    // float): the product of two nonzero steps can neither underflow (steps are >= 1e-23 in magnitude here) nor overflow, so
    // sign(a b) = sign(a) sign(b) exactly -- same decisions, CPL registers fewer than carrying xprevprev
    float sprev[CPL];
    double rho = 1.0, fbest = 0.0;
    int k = 1, nev = 0;
    bool nonfin = false;      // an objective value of the slot's document was not finite
    obj.load(dc, have ? d : -1, x);
#pragma unroll
    for (int q = 0; q < CPL; ++q) { sigma[q] = 1.0; grad[q] = 0.0; xcur[q] = x[q]; xprev[q] = x[q]; sprev[q] = 0.f; }
    const int cap = o.max_eval > 0 ? o.max_eval : 2000;
    while (__any(have)) {
        double gls = 0.0, wls = 0.0;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const double sigma2 = sigma[q] * sigma[q];
            const double ags = fabs(grad[q]) * sigma[q];
            const double v = ags + 0.5 * rho;
            const double gs2 = grad[q] * sigma2;
            double dx = dev_div(-gs2, v + dev_sqrt_pos(rho * (ags + 0.25 * rho)));      // NLopt's step with one quotient (see mma_group)
            double c = x[q] + dx;
            if (has_lb) c = dev_max_raw(c, lb);                  // (the clamps by v_max / v_min: see mma_group)
            const double hi = x[q] + 0.9 * sigma[q], lo = x[q] - 0.9 * sigma[q];
            c = dev_min_raw(dev_max_raw(c, lo), hi);
            c = (fresh || ((CplGeom<MKT, LPD>::ACT < LPD) && !obj.on)) ? x[q] : c;      // a new document: evaluate its start point; a lane without coordinates stays at 0
            xcur[q] = c;
            dx = c - x[q];
            const double dx2 = dx * dx;
            const double denominv = dev_div(1.0, sigma2 - dx2);
            // (q = 0 assigns: 0 + t = t in every bit but the sign of a zero, which no comparison below can see)
            const double gt = (fma(v, dx, gs2) * dx) * denominv, wt = dx2 * denominv;      // (the 1/2 of wval: applied to the sum, see mma_group)
            gls = q == 0 ? gt : gls + gt;
            wls = q == 0 ? wt : wls + wt;
            if (SB) __builtin_amdgcn_sched_barrier(0);       // one coordinate at a time (the interleaved chains of all coordinates need more registers)
        }
        const double gval = fbest + qsum<LPD>(gls);
        const double wval = 0.5 * qsum<LPD>(wls);
        const double fcur = obj.eval(xcur, gcur);
        nonfin = (fresh ? false : nonfin) || (have && !isfinite(fcur));
        const bool live = have && !fresh;
        bool inner_done = live && gval >= fcur;
        const bool take = fresh || (live && fcur < fbest);        // accepted before the cap is looked at, as in mma_group
        fbest = take ? fcur : fbest;
#pragma unroll
        for (int q = 0; q < CPL; ++q) { x[q] = take ? xcur[q] : x[q]; grad[q] = take ? gcur[q] : grad[q]; }
        nev = fresh ? 1 : nev + (live ? 1 : 0);
        const bool capped = live && nev >= cap;
        inner_done = inner_done && !capped;
        fresh = false;
        const bool grow = live && !capped && !inner_done && fcur > gval;
        if (__any(grow)) { const double rn = fmin(10.0 * rho, 1.1 * (rho + dev_div(fcur - gval, wval))); rho = grow ? rn : rho; }
        bool stopped = false;
        // outer iteration finished in at least one document of this wave: NLopt's x-tolerance test on (xcur, xprev)
        if (__any(inner_done)) {
            bool stop;
            if (o.xtol_rule == 0) {
                double dn = 0.0, xn = 0.0;
                bool big = false;
#pragma unroll
                for (int q = 0; q < CPL; ++q) { const double ad = fabs(xcur[q] - xprev[q]); dn += ad; xn += fabs(xcur[q]); big = big || !(ad < o.xtol_abs); }
                dn = qsum<LPD>(dn); xn = qsum<LPD>(xn);
                stop = (dn < o.xtol_rel * xn) || qnone<LPD>(big, lane);
            } else {
                bool bad = false;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const double ad = fabs(xcur[q] - xprev[q]);
                    const bool ok = isinf(xprev[q]) ? false
                                                    : (ad < o.xtol_abs || ad < o.xtol_rel * (fabs(xcur[q]) + fabs(xprev[q])) * 0.5 ||
                                                       (o.xtol_rel > 0 && xcur[q] == xprev[q]));
                    bad = bad || !ok;
                }
                stop = qnone<LPD>(bad, lane);
            }
            stopped = inner_done && stop;
            const bool nxt = inner_done && !stop;           // this document starts another outer iteration
            rho = nxt ? fmax(0.1 * rho, 1e-5) : rho;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const double dcur = xcur[q] - xprev[q];
                const float scur = dcur > 0.0 ? 1.f : (dcur < 0.0 ? -1.f : 0.f);
                const float sgn = scur * sprev[q];
                const double fac = (nxt && k > 1) ? (sgn < 0.f ? 0.7 : (sgn > 0.f ? 1.2 : 1.0)) : 1.0;
                sigma[q] *= fac;
                sprev[q] = nxt ? scur : sprev[q];
                xprev[q] = nxt ? xcur[q] : xprev[q];
            }
            k += nxt ? 1 : 0;
        }
        const bool finished = have && (stopped || capped);
        if (__any(finished)) {
            if (finished) {
                obj.store(dc, d, x);
                if (nev_out && l == 0) nev_out[d] = nev_code(nev, capped, nonfin);
            }
            // the finished slots take the next documents of the range, in slot order
            const unsigned long long fm = __ballot(finished && l == 0);
            const int nd = next + __popcll(fm & ((1ull << (g * LPD)) - 1ull));
            next += __popcll(fm);
            if (finished) {
                d = doc_at(nd); have = nd < n;
                obj.load(dc, have ? d : -1, x);
                rho = 1.0; k = 1; nev = 0; fresh = true;
#pragma unroll
                for (int q = 0; q < CPL; ++q) { sigma[q] = 1.0; grad[q] = 0.0; xcur[q] = x[q]; xprev[q] = x[q]; sprev[q] = 0.f; }
            }
        }
    }
}

template <int MKT, int LPD, int OCC, bool SB>
__global__ __launch_bounds__(256, OCC) void k_ctm_solve_cpl(CtmEArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using Gm = CplGeom<MKT, LPD>;
    constexpr int CPL = Gm::CPL, G = Gm::G, MK = MKT;
    const CtmDims& dm = a.c.dm;
    const int M = dm.M, D = dm.D;
    const size_t rep = blockIdx.y;
    if (a.active && !a.active[rep]) return;
    const double* __restrict__ p_invSigma = a.invSigma + rep * MK * MK;
    const double* __restrict__ p_mu = a.mu + rep * MK;
    const CplDocs dc{a.lam_in + rep * D * MK, a.lam_out + rep * D * MK, a.nu + rep * D * MK, a.zeta + rep * D * M, a.sumth + rep * D * MK, a.c.Ndm, M};
    int* p_nev_nu = a.nev_nu ? a.nev_nu + rep * D : nullptr;
    int* p_nev_lam = a.nev_lam ? a.nev_lam + rep * D : nullptr;
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / LPD, l = lane % LPD;
    const bool lane_on = (Gm::ACT == LPD) ? true : l < Gm::ACT;
    // LDS: [MK rows][ROW] invSigma (padded) | [NW][G][MK + 2] difference vectors
    double* sS = smem;
    double* sScr = sS + MK * Gm::ROW;
    double* sMu = sScr + (size_t)NW * G * (MK + 2);      // [LPD * CPL]
    __shared__ __attribute__((aligned(16))) double sTabs[MMM_EXPTAB_N + MMM_LOGTAB_N];      // exp | log tables of the objectives
    __shared__ int sPerm[16][MMM_WAVE];      // per wave: the order of its documents (order_range)
    stage_solve_tabs(sTabs);
    {
        for (int e = tid; e < LPD * CPL; e += blockDim.x) sMu[e] = e < MK ? p_mu[e] : 0.0;
        for (int e = tid; e < MK * Gm::ROW; e += blockDim.x) {
            const int j = e / Gm::ROW, r = e % Gm::ROW, ll = r / Gm::CPLP, q = r % Gm::CPLP;
            sS[e] = (q < CPL && ll < Gm::ACT) ? p_invSigma[(size_t)j * MK + ll * CPL + q] : 0.0;       // sS[j][i] = invSigma(i, j), column-major source
        }
        __syncthreads();
    }
    // the wave's documents: a contiguous range
    // (balanced: the first D % nwaves waves take one document more -- with ceil(D / nwaves) per wave a launch of one wave per SIMD over a small
    // shard left its last waves without documents)
    const int nwaves = gridDim.x * NW, w = blockIdx.x * NW + wid;
    const int base = D / nwaves, rem = D % nwaves;
    const int r0 = w * base + min(w, rem), r1 = r0 + base + (w < rem ? 1 : 0);
    static_assert(CPL <= 7, "modality indices are packed 4 bits each into one int");
    int modpack = 0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int i = l * CPL + q;
        int mm = 0;
        for (int m = 0; m < M; ++m) if (i >= dm.koff[m] && i < dm.koff[m + 1]) mm = m;
        modpack |= mm << (4 * q);
    }
    int uni = 0;
#pragma unroll
    for (int q = 1; q < CPL; ++q) uni |= __all(((modpack >> (4 * q)) & 15) == ((modpack >> (4 * (q - 1))) & 15)) ? 1 << q : 0;
    uni = __builtin_amdgcn_readfirstlane(uni);
    const SolveOpts o = a.opt;
    // ---- update_ν! (MMCTM.jl:156-170): LD_MMA, lower bound 1e-7, from the current ν, with the old λ -- for every document of the range
    if (a.flags & F_NU) {
        NuObjC<MKT, LPD, SB> obj;
#pragma unroll
        for (int q = 0; q < CPL; ++q) obj.Sll[q] = lane_on ? p_invSigma[(size_t)(l * CPL + q) * MK + l * CPL + q] : 0.0;
        obj.modpack = modpack; obj.uni = uni;
        obj.l = l; obj.lane_on = lane_on; obj.tabs = sTabs;
        solve_range<MKT, LPD, SB>(obj, dc, r0, r1, lane, true, o.nu_lower, o, p_nev_nu, false, sPerm[wid]);      // (the nu solves' counts of consecutive passes do not correlate)
    }
    // the λ solves read the ν this wave has just stored (any slot may have solved a given document's ν)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- update_λ! (MMCTM.jl:127-143): LD_MMA, unbounded, with the new ν
    if (a.flags & F_LAMBDA) {
        LamObjC<MKT, LPD, SB> obj;
        obj.modpack = modpack; obj.uni = uni;
        obj.tabs = sTabs;
        obj.smu = sMu;
        obj.l = l; obj.lane_on = lane_on; obj.sS = sS; obj.scr = sScr + ((size_t)wid * G + g) * (MK + 2);
        solve_range<MKT, LPD, SB>(obj, dc, r0, r1, lane, false, 0.0, o, p_nev_lam, (a.flags & F_ORDER_LAM) != 0, sPerm[wid]);
    }
}
