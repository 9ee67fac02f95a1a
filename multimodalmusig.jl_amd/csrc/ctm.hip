// ctm.hip -- MMCTM / IMMCTM variational EM on gfx950 (replaces the hot path of src/MMCTM.jl, src/IMMCTM.jl, common.jl)
//
// Data layout in HBM (per model handle)
//   corpus     doc_ptr int64[M*(D+1)] absolute offsets; tc int2[nnz] = (term0, count); Ndm double[D*M]
//   documents  lambda ring[2] [d][MK]; nu [d][MK]; zeta [d][M]; props [d][MK]; theta flat (on demand)
//   topics     gamma, Elnphi in the model layout (MMCTM [m][k][v]; IMMCTM [m][k][i][j]) and three "effective"
//              [m][k][v] tables: Eeff = Elnphi (IMMCTM: sum over features), exp(Eeff) (ring of 2), phieff
//   Gaussian   mu[MK], Sigma, invSigma [MK x MK]
//
// One outer iteration (MMCTM.jl:462-479) on one GPU:
//   k_ctm_estep<L>   fitdoc! for every document: a wave handles 64/L documents, L >= MK lanes per document.  Lanes are
//                    the MK coordinates for zeta and the two MMA solves (nu, lambda), and the document's nonzero terms
//                    for theta.  theta_kw = a_k B_kv / sum_k a_k B_kv with a = exp(lambda - max) and B = exp(Eeff) in LDS
//                    (no exp per (term, topic)); theta n is scattered into the wave's LDS slab (-> gamma) and summed
//                    over terms (-> sumtheta); theta itself is not written (rebuilt on demand from lambda_{t-1}).
//   k_reduce_partials  per-block slabs -> gamma statistics, fixed order
//   k_ctm_moments    sum lambda, sum nu, sum lambda lambda^T per block -> fixed-order reduce
//   (RCCL all-reduce of the packed statistics when documents are sharded over ranks)
//   k_ctm_mstep      mu, Sigma = (diag sum nu + sum ll^T)/D - mu mu^T, invSigma (Gauss-Jordan in LDS), gamma, Elnphi, tables
//   k_ctm_loglik     props = softmax(lambda block) and the per-modality log-likelihood sums (MMCTM.jl:384-448)
//
// The MMA solves restate NLopt's LD_MMA for zero constraints (algorithm statement: DESIGN.md "MMA" and
// SURVEY.md §7): one objective evaluation per trip of a single wave-uniform loop, per-group state, select-based commits.
#include <memory>
#include "dev_math.h"
#include "mmm_logtab.h"
#include "mmm_exptab.h"
#include "mmm_internal.h"

namespace {

constexpr int kMaxM = 8;
constexpr int kKmax = 64;          // topics per modality (the theta loop is unrolled to 8 / 10 / 16 / 32; 33..64: the 64-topic build of the wide-table path;
                                   // sum K > 64: ctm_big.cuh)
constexpr int kWavesS = 4;         // stage / auxiliary kernels
constexpr int kBlockS = kWavesS * MMM_WAVE;

enum { F_ZETA = 1, F_THETA_COMPUTE = 2, F_THETA_STORED = 4, F_THETA_STORE = 8, F_NU = 16, F_LAMBDA = 32, F_SLAB = 64 };

struct CtmDims {
    int D, M, MK, GT;                    // GT = sum_m K_m V_m
    int K[kMaxM], V[kMaxM], koff[kMaxM + 1], goff[kMaxM + 1];
    long long estart[kMaxM], toff[kMaxM];
};

struct CtmDev {
    CtmDims dm;
    const int64_t* doc_ptr;
    const int2* tc;
    const double* Ndm;
};

struct SolveOpts { double xtol_rel, xtol_abs, nu_lower; int xtol_rule, max_eval; };

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

// NLopt LD_MMA, zero constraints, for the L-lane group of the calling lane (lane l holds coordinate l).  All lanes of the
// wave execute every trip; a finished group keeps its state through selects.  Returns the number of objective
// evaluations (negative: the evaluation cap was hit).
template <int L, int LP, class Obj>
__device__ int mma_group(const Obj& obj, bool act, int g, double& x, bool has_lb, double lb, const SolveOpts& o, const PackCtx& pc)
{
    double sigma = 1.0, rho = 1.0;
    double gcur, grad;
    double fbest = obj.template eval<L, LP>(x, grad, pc);
    double xcur = x, xprev = x, xprevprev = x;
    int k = 1, nev = 1;
    bool done = false, capped = false;
    const int cap = o.max_eval > 0 ? o.max_eval : 2000;
    while (!__all(done)) {
        // closed-form minimiser of the separable approximation (dual problem is trivial for m = 0)
        // NLopt: u = g sigma^2, v = |g| sigma + rho/2, dx = (u/v) / (-1 - sqrt|1 - (u/(v sigma))^2|).  u/(v sigma) = (g sigma)/v =: q and u/v = q sigma:
        // ONE correctly rounded quotient instead of three (the two forms differ by an ulp or two; the order-matched CPU checker of the
        // parity tests follows this one, the index-order one keeps NLopt's)
        const double sigma2 = sigma * sigma;
        const double v = fabs(grad) * sigma + 0.5 * rho;
        const double q = dev_div(grad * sigma, v);
        double dx = dev_div(q * sigma, -1.0 - dev_sqrt(fabs(1.0 - q * q)));
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
        const double gl = act ? (grad * (sigma2 * dx) + (fabs(grad) * sigma + 0.5 * rho) * dx2) * denominv : 0.0;
        const double wl = act ? 0.5 * dx2 * denominv : 0.0;
        double gsm = gl, wval = wl;
        gsum2<L, LP>(pc, gsm, wval);
        const double gval = fbest + gsm;
        const double fcur = obj.template eval<L, LP>(xc, gcur, pc);
        bool inner_done = false;
        if (!done) {
            ++nev;
            xcur = xc;
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
    return capped ? -nev : nev;
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
            const double Nl = d < 0 ? 0.0 : dc.Ndm[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)], zl = d < 0 ? 1.0 : dc.zeta[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)];
            c[q] = Nl / zl;                                     // Ndivζ (MMCTM.jl:119-125)
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
            const double Nl = d < 0 ? 0.0 : dc.Ndm[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)], zl = d < 0 ? 1.0 : dc.zeta[(size_t)d * dc.M + ((modpack >> (4 * q)) & 15)];
            c[q] = Nl / zl;
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
template <int MKT, int LPD, bool SB, class Obj>
__device__ __forceinline__ void solve_range(Obj& obj, const CplDocs& dc, int r0, int r1, int lane, bool has_lb, double lb, const SolveOpts& o, int* nev_out)
{
    constexpr int CPL = CplGeom<MKT, LPD>::CPL, G = MMM_WAVE / LPD;
    const int g = lane / LPD, l = lane % LPD;
    int d = r0 + g, next = r0 + G;
    bool have = d < r1, fresh = true;
    double x[CPL], sigma[CPL], grad[CPL], gcur[CPL], xcur[CPL], xprev[CPL];
    // NLopt's sigma update looks at the SIGN of (xcur - xprev) (xprev - xprevprev).  Only the sign of the older step is kept (+1 / 0 / -1 as a
    // float): the product of two nonzero steps can neither underflow (steps are >= 1e-23 in magnitude here) nor overflow, so
    // sign(a b) = sign(a) sign(b) exactly -- same decisions, CPL registers fewer than carrying xprevprev
    float sprev[CPL];
    double rho = 1.0, fbest = 0.0;
    int k = 1, nev = 0;
    obj.load(dc, have ? d : -1, x);
#pragma unroll
    for (int q = 0; q < CPL; ++q) { sigma[q] = 1.0; grad[q] = 0.0; xcur[q] = x[q]; xprev[q] = x[q]; sprev[q] = 0.f; }
    const int cap = o.max_eval > 0 ? o.max_eval : 2000;
    while (__any(have)) {
        double gls = 0.0, wls = 0.0;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const double sigma2 = sigma[q] * sigma[q];
            const double v = fabs(grad[q]) * sigma[q] + 0.5 * rho;
            const double qq = dev_div(grad[q] * sigma[q], v);                 // = u / (v sigma) of NLopt's formula; u / v = qq sigma (see mma_group)
            double dx = dev_div(qq * sigma[q], -1.0 - dev_sqrt(fabs(1.0 - qq * qq)));
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
            const double gt = (grad[q] * (sigma2 * dx) + (fabs(grad[q]) * sigma[q] + 0.5 * rho) * dx2) * denominv, wt = 0.5 * dx2 * denominv;
            gls = q == 0 ? gt : gls + gt;
            wls = q == 0 ? wt : wls + wt;
            if (SB) __builtin_amdgcn_sched_barrier(0);       // one coordinate at a time (the interleaved chains of all coordinates need more registers)
        }
        const double gval = fbest + qsum<LPD>(gls);
        const double wval = qsum<LPD>(wls);
        const double fcur = obj.eval(xcur, gcur);
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
                if (nev_out && l == 0) nev_out[d] = capped ? -nev : nev;
            }
            // the finished slots take the next documents of the range, in slot order
            const unsigned long long fm = __ballot(finished && l == 0);
            const int nd = next + __popcll(fm & ((1ull << (g * LPD)) - 1ull));
            next += __popcll(fm);
            if (finished) {
                d = nd; have = nd < r1;
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
    const int nwaves = gridDim.x * NW, w = blockIdx.x * NW + wid;
    const int per = (D + nwaves - 1) / nwaves;
    const int r0 = min(D, w * per), r1 = min(D, r0 + per);
    static_assert(CPL <= 7, "modality indices are packed 4 bits each into one int");
    int modpack = 0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int i = l * CPL + q;
        int mm = 0;
        for (int m = 0; m < M; ++m) if (i >= dm.koff[m] && i < dm.koff[m + 1]) mm = m;
        modpack |= mm << (4 * q);
    }
    const SolveOpts o = a.opt;
    // ---- update_ν! (MMCTM.jl:156-170): LD_MMA, lower bound 1e-7, from the current ν, with the old λ -- for every document of the range
    if (a.flags & F_NU) {
        NuObjC<MKT, LPD, SB> obj;
#pragma unroll
        for (int q = 0; q < CPL; ++q) obj.Sll[q] = lane_on ? p_invSigma[(size_t)(l * CPL + q) * MK + l * CPL + q] : 0.0;
        obj.modpack = modpack;
        obj.l = l; obj.lane_on = lane_on; obj.tabs = sTabs;
        solve_range<MKT, LPD, SB>(obj, dc, r0, r1, lane, true, o.nu_lower, o, p_nev_nu);
    }
    // the λ solves read the ν this wave has just stored (any slot may have solved a given document's ν)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- update_λ! (MMCTM.jl:127-143): LD_MMA, unbounded, with the new ν
    if (a.flags & F_LAMBDA) {
        LamObjC<MKT, LPD, SB> obj;
        obj.modpack = modpack;
        obj.tabs = sTabs;
        obj.smu = sMu;
        obj.l = l; obj.lane_on = lane_on; obj.sS = sS; obj.scr = sScr + ((size_t)wid * G + g) * (MK + 2);
        solve_range<MKT, LPD, SB>(obj, dc, r0, r1, lane, false, 0.0, o, p_nev_lam);
    }
}

#include "ctm_big.cuh"

// objective values / gradients of one document at its stored (lambda, nu), in the reference's MAXIMISATION form
// (common.jl:11-36), evaluated by the same device functors the solvers use.  One wave.
__global__ __launch_bounds__(64) void k_ctm_objectives(CtmDev c, int d, const double* invSigma, const double* mu, const double* lam,
                                                       const double* nu, const double* zeta, const double* theta, double* out)
{
    __shared__ double sS[64 * 64];
    __shared__ double scr[64];
    __shared__ __attribute__((aligned(16))) double sTabs[MMM_EXPTAB_N + MMM_LOGTAB_N];
    const CtmDims& dm = c.dm;
    const int MK = dm.MK, M = dm.M, l = threadIdx.x;
    for (int i = l; i < MK * MK; i += 64) sS[i] = invSigma[i];
    stage_solve_tabs(sTabs);
    __syncthreads();
    const bool act = l < MK;
    int mod_l = 0;
    for (int m = 0; m < M; ++m) if (l >= dm.koff[m] && l < dm.koff[m + 1]) mod_l = m;
    const double x = act ? lam[(size_t)d * MK + l] : 0.0, v = act ? nu[(size_t)d * MK + l] : 1.0;
    const double cl = act ? c.Ndm[(size_t)d * M + mod_l] / zeta[(size_t)d * M + mod_l] : 0.0;
    double sumth = 0.0;
    if (act) {
        const int m = mod_l, Km = dm.K[m], k = l - dm.koff[m];
        const int64_t* dp = c.doc_ptr + (size_t)m * (dm.D + 1);
        for (int64_t e = dp[d]; e < dp[d + 1]; ++e) sumth += theta[dm.toff[m] + (size_t)(e - dm.estart[m]) * Km + k] * (double)c.tc[e].y;
    }
    double g1, g2;
    LamObj<0> lo{v, cl, sumth, act ? mu[l] : 0.0, act, l, MK, sS, scr, sTabs};
    const double f1 = lo.eval<64>(x, g1);
    NuObj no{x, cl, act ? sS[l * MK + l] : 1.0, act, sTabs};
    const double f2 = no.eval<64>(v, g2);
    if (l == 0) { out[0] = -f1; out[1] = -f2; }
    if (act) { out[2 + l] = -g1; out[2 + MK + l] = -g2; }
}

// calculate_sumθ(model, d) / calculate_Ndivζ(model, d) (MMCTM.jl:110-125) from the stored θ and ζ: out[i] = Σ_w θ[k, w] n_w (w ascending),
// out[MK + i] = N_dm / ζ_dm for coordinate i = off_m + k.  One block, one thread per coordinate.
__global__ __launch_bounds__(256) void k_ctm_doc_sums(CtmDev c, int d, const double* zeta, const double* theta, double* out)
{
    const CtmDims& dm = c.dm;
    const int MK = dm.MK, M = dm.M, l = threadIdx.x;
    if (l >= MK) return;
    int m = 0;
    for (int q = 0; q < M; ++q) if (l >= dm.koff[q] && l < dm.koff[q + 1]) m = q;
    const int Km = dm.K[m], k = l - dm.koff[m];
    const int64_t* dp = c.doc_ptr + (size_t)m * (dm.D + 1);
    double s = 0.0;
    for (int64_t e = dp[d]; e < dp[d + 1]; ++e) s += theta[dm.toff[m] + (size_t)(e - dm.estart[m]) * Km + k] * (double)c.tc[e].y;
    out[l] = s;
    out[MK + l] = c.Ndm[(size_t)d * M + m] / zeta[(size_t)d * M + m];
}

// dst's theta columns of document d <- src's (per-document stage calls: only document d keeps the stage's result)
__global__ __launch_bounds__(256) void k_ctm_copy_doc_theta(CtmDev c, int d, const double* src, double* dst)
{
    const CtmDims& dm = c.dm;
    for (int m = 0; m < dm.M; ++m) {
        const int64_t* dp = c.doc_ptr + (size_t)m * (dm.D + 1);
        const size_t b = dm.toff[m] + (size_t)(dp[d] - dm.estart[m]) * dm.K[m], n = (size_t)(dp[d + 1] - dp[d]) * dm.K[m];
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[b + i] = src[b + i];
    }
}

// wide tables: gamma statistics of one (modality, term) per block -- sums[goff[m] + k V_m + v] = sum over the term's postings of
// n theta_kw (MMCTM.jl:230-240), theta_kw = a_dk e_kv / sum_k' a_dk' e_k'v from the theta phase's a_d rows and the term's table
// column (scalar registers).  Postings (doc, count) in document order, split over the block's waves in contiguous segments,
// segment sums added in segment order: a fixed summation order, no atomics.
template <int KMX>
__global__ __launch_bounds__(512) void k_ctm_stats_terms(CtmDims dm, const int64_t* __restrict__ term_ptr, const int2* __restrict__ tpost,
                                                         const double* __restrict__ aexp, const double* __restrict__ expE, double* __restrict__ out,
                                                         size_t out_stride, const int* active)
{
    __shared__ double sh[8][KMX];
    if (active && !active[blockIdx.y]) return;
    const size_t rep = blockIdx.y;
    int m = 0, v = blockIdx.x;
    while (m + 1 < dm.M && v >= dm.V[m]) { v -= dm.V[m]; ++m; }
    const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m], MK = dm.MK;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double* __restrict__ col = expE + rep * dm.GT + dm.goff[m] + v;
    aexp += rep * (size_t)dm.D * MK + off;
    double eb[KMX], acc[KMX];
#pragma unroll
    for (int k = 0; k < KMX; ++k) { eb[k] = (k < Km) ? col[(size_t)k * Vm] : 0.0; acc[k] = 0.0; }
    const int64_t p0 = term_ptr[blockIdx.x], p1 = term_ptr[blockIdx.x + 1];
    const int64_t seg = (p1 - p0 + nw - 1) / nw;
    const int64_t q0 = p0 + wid * seg, q1 = (q0 + seg < p1) ? q0 + seg : p1;
    for (int64_t j = q0 + lane; j < q1; j += MMM_WAVE) {
        const int2 dn = tpost[j];
        const double* __restrict__ ad = aexp + (size_t)dn.x * MK;
        double e[KMX], s = 0.0;
#pragma unroll
        for (int k = 0; k < KMX; ++k) { e[k] = (k < Km) ? ad[k] * eb[k] : 0.0; s += e[k]; }
        const double rn = (double)dn.y / s;
#pragma unroll
        for (int k = 0; k < KMX; ++k) acc[k] = fma(e[k], rn, acc[k]);
    }
#pragma unroll
    for (int k = 0; k < KMX; ++k) { const double tot = wave_sum(acc[k]); if (lane == 0) sh[wid][k] = tot; }
    __syncthreads();
    if ((int)threadIdx.x < Km) {
        double tot = 0.0;
        for (int w = 0; w < nw; ++w) tot += sh[w][threadIdx.x];
        out[rep * out_stride + dm.goff[m] + (size_t)threadIdx.x * Vm + v] = tot;
    }
}

// partial[nslab][n] -> out[n], fixed summation order; grid = ceil(n/16) blocks of (16, 64)
// an optional second job (part2 ... out2) rides in the same launch: blocks [nb1, gridDim.x)
__global__ __launch_bounds__(1024) void k_reduce_partials(const double* __restrict__ part, int nslab, int n, double* __restrict__ out,
                                                          size_t out_stride, const int* active, int nb1 = 0x7fffffff,
                                                          const double* __restrict__ part2 = nullptr, int nslab2 = 0, int n2 = 0,
                                                          double* __restrict__ out2 = nullptr)
{
    __shared__ double sm[64][17];
    if (active && !active[blockIdx.y]) return;
    int bx = blockIdx.x;
    if (bx >= nb1) { bx -= nb1; part = part2; nslab = nslab2; n = n2; out = out2; }
    part += (size_t)blockIdx.y * nslab * n; out += (size_t)blockIdx.y * out_stride;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int e = bx * 16 + tx;
    double acc = 0.0;
    if (e < n) for (int sl = ty; sl < nslab; sl += 64) acc += part[(size_t)sl * n + e];
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty < 8) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[ty * 8 + j][tx];
        sm[ty * 8][tx] = v;
    }
    __syncthreads();
    if (ty == 0 && e < n) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[j * 8][tx];
        out[e] = v;
    }
}

// per-block partial sums of lambda (MK), nu (MK), lambda lambda^T (MK*MK): part[block][2MK + MK*MK].  A block walks its
// contiguous document range in tiles of 32 documents staged in LDS (coalesced loads); thread e owns output entry e.
__global__ __launch_bounds__(256) void k_ctm_moments(int D, int MK, const double* __restrict__ lam, const double* __restrict__ nu, double* part,
                                                     const int* active)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];      // [32*MK] lambda tile, [32*MK] nu tile
    constexpr int T = 32;
    if (active && !active[blockIdx.y]) return;
    lam += (size_t)blockIdx.y * D * MK; nu += (size_t)blockIdx.y * D * MK; part += (size_t)blockIdx.y * gridDim.x * (2 * MK + MK * MK);
    double* sL = smem; double* sN = smem + T * MK;
    const int n = 2 * MK + MK * MK;
    const int per = (D + gridDim.x - 1) / gridDim.x;
    const int d0 = blockIdx.x * per, d1 = min(D, d0 + per);
    // up to 4 output entries per thread (n <= 2*64 + 64*64 needs more: loop)
    for (int e0 = 0; e0 < n; e0 += 4 * blockDim.x) {
        double acc[4] = {0, 0, 0, 0};
        // what each of this thread's entries reads: two LDS columns (a, b) as OFFSETS into smem (pointers picked from sL / sN at run time
        // lose their address space: the loop's reads became flat loads through the vector-memory path, 473 per wave); kind 0: sum a, 1: sum a*b
        int oa[4], ob[4], kind[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * blockDim.x + threadIdx.x;
            kind[q] = -1; oa[q] = 0; ob[q] = 0;
            if (e < MK) { kind[q] = 0; oa[q] = e; }
            else if (e < 2 * MK) { kind[q] = 0; oa[q] = T * MK + (e - MK); }
            else if (e < n) { kind[q] = 1; oa[q] = (e - 2 * MK) % MK; ob[q] = (e - 2 * MK) / MK; }
        }
        for (int t0 = d0; t0 < d1; t0 += T) {
            const int nt = min(T, d1 - t0);
            __syncthreads();
            // a short last tile is zero-padded, so the sums below always run over T documents (compile-time trip count)
            for (int i = threadIdx.x; i < T * MK; i += blockDim.x) {
                const bool in = i < nt * MK;
                sL[i] = in ? lam[(size_t)t0 * MK + i] : 0.0; sN[i] = in ? nu[(size_t)t0 * MK + i] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (kind[q] < 0) continue;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // independent chains: the LDS reads pipeline
                const int A = oa[q], B = ob[q];
                if (kind[q] == 0) {
#pragma unroll 2
                    for (int d = 0; d < T; d += 4) { s0 += smem[A + d * MK]; s1 += smem[A + (d + 1) * MK]; s2 += smem[A + (d + 2) * MK]; s3 += smem[A + (d + 3) * MK]; }
                } else {
#pragma unroll 2
                    for (int d = 0; d < T; d += 4) {
                        s0 = fma(smem[A + d * MK], smem[B + d * MK], s0); s1 = fma(smem[A + (d + 1) * MK], smem[B + (d + 1) * MK], s1);
                        s2 = fma(smem[A + (d + 2) * MK], smem[B + (d + 2) * MK], s2); s3 = fma(smem[A + (d + 3) * MK], smem[B + (d + 3) * MK], s3);
                    }
                }
                acc[q] += (s0 + s1) + (s2 + s3);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * blockDim.x + threadIdx.x;
            if (e < n) part[(size_t)blockIdx.x * n + e] = acc[q];
        }
    }
}

// ---- M-step (one block) -------------------------------------------------------------------------------------------
struct CtmTopics {            // model-layout topic parameters
    int immctm;
    int nfeat[kMaxM], aoff[kMaxM + 1];      // IMMCTM: features per modality, offset into J/alpha
    int J[4 * kMaxM];                        // concatenated J[m][i]
    int SJ[kMaxM];                           // sum_i J[m][i]
    int mgoff[kMaxM + 1];                    // model-layout offsets of gamma/Elnphi per modality
    long long foff[kMaxM];                   // offsets into features
    const int* features;                     // [foff[m] + i*V + v]
    const double* alpha;                     // MMCTM: [M]; IMMCTM: [aoff]
};

struct MstepArgs {
    CtmDims dm; CtmTopics tp;
    const double* stats;    // [MK | MK | MK*MK | GT]
    double Dglobal;
    double* mu; double* Sigma; double* invSigma;
    double* gamma; double* Elnphi; double* phi;      // model layout (phi: MMCTM only, may be NULL)
    double* Eeff; double* expEeff; double* phieff;   // [GT]
    int* status;            // 0 ok, 1 singular Sigma
    int do_mu, do_sigma, do_gamma, gamma_from_stats;
    size_t stats_stride; int GM; const int* active;      // batched launches
    int nalpha;
    double* big_scratch;    // sum K > 64: [R][2 MK^2] doubles in device memory for the inversion (block_inverse_big); else NULL
};

// the per-replica pointers of a batched M-step launch.  They are formed in locals (registers); the argument struct itself
// stays untouched in the kernarg segment, so that its dimension arrays keep being read with scalar loads.
struct MstepPtrs {
    const double* stats; double* mu; double* Sigma; double* invSigma; double* gamma; double* Elnphi; double* phi;
    double* Eeff; double* expEeff; double* phieff; int* status; const double* alpha;
};

__device__ __forceinline__ bool mstep_replica(const MstepArgs& a, MstepPtrs& q)
{
    const size_t r = blockIdx.y;
    if (a.active && !a.active[r]) return false;
    const size_t MK = a.dm.MK, GT = a.dm.GT, GM = a.GM;
    q.stats = a.stats + r * a.stats_stride; q.mu = a.mu + r * MK; q.Sigma = a.Sigma + r * MK * MK; q.invSigma = a.invSigma + r * MK * MK;
    q.gamma = a.gamma + r * GM; q.Elnphi = a.Elnphi + r * GM; q.phi = a.phi ? a.phi + r * GM : nullptr;
    q.Eeff = a.Eeff + r * GT; q.expEeff = a.expEeff + r * GT; q.phieff = a.phieff + r * GT; q.status = a.status + r;
    q.alpha = a.tp.alpha + r * a.nalpha;
    return true;
}

// in-place Gauss-Jordan inverse with partial pivoting of the n x n matrix A (LDS, row stride n) into Ainv; log|det A| in
// *logdet.  One block of >= 2n threads, n <= 64.  (A single-wave variant -- lanes own columns, multipliers by readlane, no
// block barriers -- was measured slower: 113 vs 59 us for the launch at n = 28; its row updates are LDS-latency bound.)
__device__ void block_inverse_wide(int n, double* A, double* Ainv, double* logdet, int* singular, int* s_piv);

__device__ void block_inverse(int n, double* A, double* Ainv, double* logdet, int* singular, int* s_piv)
{
    block_inverse_wide(n, A, Ainv, logdet, singular, s_piv);
}

// Per column, three block barriers: (1) wave 0 finds the pivot -- lane = row, the largest magnitude of the wave by DPP and row swaps
// (wave_max_dpp), its lowest row by ballot (the tie rule of a sequential search) -- and leaves the pivot, the entry A[c][c] it is swapped
// with and the magnitude in LDS cells of their own, so that nobody has to read them from rows that the next sweep rewrites; (2) one
// sweep swaps + scales the pivot row and collects the column's multipliers; (3) one sweep eliminates; threads keep a fixed (row-phase,
// column) assignment.  log|det| is summed after the loop, in column order (the logs in parallel).  Per element the operations are
// those of the five-barrier version of rounds 1-2 (pivot search by shuffles, the log inside the loop: 1.9 us per column, 54 us at n = 28).
__device__ void block_inverse_wide(int n, double* A, double* Ainv, double* logdet, int* singular, int* s_piv)
{
    __shared__ double s_col[64], s_best[64], s_pv[2];
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n * n; i += nt) Ainv[i] = 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += nt) Ainv[i * n + i] = 1.0;
    if (tid == 0) *singular = 0;
    const int j0 = tid % n, r0 = tid / n, rstep = nt / n;      // thread -> column j0, rows r0, r0 + rstep, ...
    __syncthreads();
    for (int c = 0; c < n; ++c) {
        if (tid < 64) {
            const double a = tid < n ? A[tid * n + c] : 0.0;
            const double mag = (tid >= c && tid < n) ? fabs(a) : -1.0;
            const double best = wave_max_dpp(mag);
            const unsigned long long eq = __ballot(mag == best);
            const int p = eq ? (int)__builtin_ctzll(eq) : c;        // (no lane compares equal only if the column holds NaNs)
            if (tid == p) { *s_piv = p; s_pv[0] = a; }
            if (tid == c) s_pv[1] = a;
            if (tid == 0) { s_best[c] = best; if (!(best > 0.0)) *singular = 1; }
        }
        __syncthreads();
        const int p = *s_piv;
        const double piv = s_pv[0];
        if (tid < 2 * n) {
            double* Mx = tid < n ? A : Ainv;
            const int j = tid < n ? tid : tid - n;
            const double top = Mx[c * n + j], low = Mx[p * n + j];
            // the column's multipliers as the elimination will find them after the swap: row p holds the old A[c][c] (row c is skipped)
            if (tid < n) s_col[tid] = (tid == p) ? s_pv[1] : A[tid * n + c];
            Mx[c * n + j] = low / piv;
            if (p != c) Mx[p * n + j] = top;
        }
        __syncthreads();
        if (r0 < rstep) {
            const double ac = A[c * n + j0], ic = Ainv[c * n + j0];
            for (int r = r0; r < n; r += rstep) {
                if (r == c) continue;
                const double f = s_col[r];
                A[r * n + j0] = (j0 == c) ? 0.0 : A[r * n + j0] - f * ac;
                Ainv[r * n + j0] -= f * ic;
            }
        }
        __syncthreads();
    }
    if (tid < n) s_col[tid] = log(s_best[tid]);
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int c = 0; c < n; ++c) s += s_col[c]; *logdet = s; }
}

// update_μ! / update_Σ! of one replica by the calling block (>= 128 threads); smem: 2 MK^2 doubles.  BIG (sum K > 64): the matrices live
// in device memory -- a compile-time switch, so that the LDS build keeps LDS addressing (a run-time choice of the base pointer turned every
// access of the inversion into a flat one: 54 -> 84 us for the Gaussian block at sum K = 28)
template <bool BIG>
__device__ void ctm_gauss_mstep(const MstepArgs& a, const MstepPtrs& q, double* smem)
{
    __shared__ double s_logdet; __shared__ int s_sing, s_piv;
    const CtmDims& dm = a.dm;
    const int MK = dm.MK, tid = threadIdx.x, nt = blockDim.x;
    const double* sLam = q.stats; const double* sNu = q.stats + MK; const double* sLL = q.stats + 2 * MK;
    // update_μ! (MMCTM.jl:200-202)
    if (a.do_mu) { for (int i = tid; i < MK; i += nt) q.mu[i] = sLam[i] / a.Dglobal; }
    __syncthreads();
    // update_Σ! (MMCTM.jl:204-212) from raw moments: (diag Σν + Σ (λ-μ)(λ-μ)') / D with the NEW μ
    if (a.do_sigma) {
        double* A = BIG ? a.big_scratch + (size_t)blockIdx.y * 2 * MK * MK : smem;
        double* Ai = A + MK * MK;
        for (int e = tid; e < MK * MK; e += nt) {
            const int i = e % MK, j = e / MK;
            // Σ_d (λ_i-μ_i)(λ_j-μ_j) = Σλλ' - μ_i Σλ_j - μ_j Σλ_i + D μ_i μ_j
            const double mi = q.mu[i], mj = q.mu[j];
            double v = sLL[e] - mi * sLam[j] - mj * sLam[i] + a.Dglobal * mi * mj;
            if (i == j) v += sNu[i];
            v /= a.Dglobal;
            q.Sigma[e] = v; A[i * MK + j] = v;
        }
        __syncthreads();
        if constexpr (BIG) block_inverse_big(MK, A, Ai, &s_logdet, &s_sing);
        else block_inverse(MK, A, Ai, &s_logdet, &s_sing, &s_piv);
        __syncthreads();
        for (int e = tid; e < MK * MK; e += nt) { const int i = e % MK, j = e / MK; q.invSigma[e] = Ai[i * MK + j]; }
        if (tid == 0 && s_sing) *q.status = 1;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_ctm_mstep(MstepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    MstepPtrs q;
    if (!mstep_replica(a, q)) return;
    if (a.big_scratch) ctm_gauss_mstep<true>(a, q, smem);
    else ctm_gauss_mstep<false>(a, q, smem);
}

// update_γ! / update_Elnϕ! / update_ϕ! (MMCTM.jl:214-250; IMMCTM.jl:188-223): one block per topic (m,k) -- topics are
// independent of each other and of the Gaussian part, so they run beside block 0 of k_ctm_mstep's work.
__global__ __launch_bounds__(256) void k_ctm_mstep_topics(MstepArgs a)
{
    __shared__ double sh[4];
    MstepPtrs q;
    if (!mstep_replica(a, q)) return;
    const CtmDims& dm = a.dm;
    const CtmTopics& tp = a.tp;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double* sG = q.stats + 2 * dm.MK + dm.MK * dm.MK;
    int m = 0;
    while (m + 1 < dm.M && (int)blockIdx.x >= dm.koff[m + 1]) ++m;
    const int k = blockIdx.x - dm.koff[m];
    const int Vm = dm.V[m], go = dm.goff[m];
    if (!tp.immctm) {
        double part = 0.0;
        for (int v = tid; v < Vm; v += nt) {
            const double gm = a.gamma_from_stats ? q.alpha[m] + sG[go + k * Vm + v] : q.gamma[go + k * Vm + v];
            if (a.gamma_from_stats) q.gamma[go + k * Vm + v] = gm;
            part += gm;
        }
        part = wave_sum(part);
        if ((tid & 63) == 0) sh[tid >> 6] = part;
        __syncthreads();
        const double cs = sh[0] + sh[1] + sh[2] + sh[3];
        const double pcs = dev_digamma_ar(cs);
        for (int v = tid; v < Vm; v += nt) {
            const double gm = q.gamma[go + k * Vm + v];
            const double el = dev_digamma_ar(gm) - pcs;
            q.Elnphi[go + k * Vm + v] = el; q.Eeff[go + k * Vm + v] = el; q.expEeff[go + k * Vm + v] = ar_exp(el);
            const double ph = gm / cs;
            if (q.phi) q.phi[go + k * Vm + v] = ph;
            q.phieff[go + k * Vm + v] = ph;
        }
    } else {
        const int mg = tp.mgoff[m], SJ = tp.SJ[m], nf = tp.nfeat[m], ao = tp.aoff[m];
        const int* feat = tp.features + tp.foff[m];
        // gamma[m][k][i][j] = alpha[m][i] + sum_{v: f_vi = j} S[m][k][v]   (IMMCTM.jl:199-221).  The topic's statistics row and the feature
        // table are staged in LDS first (coalesced): the sum(J) threads that fold them walk all V terms each, in term order
        constexpr int kStage = 1024;
        __shared__ double sh_row[kStage];
        __shared__ int sh_feat[4 * kStage];
        const bool staged = a.gamma_from_stats && Vm <= kStage && nf * Vm <= 4 * kStage;
        if (staged) {
            for (int v = tid; v < Vm; v += nt) sh_row[v] = sG[go + k * Vm + v];
            for (int e = tid; e < nf * Vm; e += nt) sh_feat[e] = feat[e];
            __syncthreads();
        }
        // the topic's gamma / Elnphi rows (sum(J) values) stay in LDS between the three steps below: written to the model arrays once,
        // never read back from memory (each read-back was a global round trip inside a 10-block launch)
        constexpr int kRow = 512;
        __shared__ double sh_gam[kRow], sh_eln[kRow];
        const bool rows = SJ <= kRow;
        if (a.gamma_from_stats) for (int e = tid; e < SJ; e += nt) {
            int jj = e, i = 0;
            while (jj >= tp.J[ao + i]) { jj -= tp.J[ao + i]; ++i; }
            double s = q.alpha[ao + i];
            if (staged) { for (int v = 0; v < Vm; ++v) if (sh_feat[i * Vm + v] == jj) s += sh_row[v]; }
            else for (int v = 0; v < Vm; ++v) if (feat[i * Vm + v] == jj) s += sG[go + k * Vm + v];
            q.gamma[mg + k * SJ + e] = s;
            if (rows) sh_gam[e] = s;
        }
        else if (rows) for (int e = tid; e < SJ; e += nt) sh_gam[e] = q.gamma[mg + k * SJ + e];
        __syncthreads();
        const double* gam = rows ? sh_gam : q.gamma + mg + (size_t)k * SJ;
        // Elnphi[m][k][i][j] = psi(gamma) - psi(sum_j gamma)   (IMMCTM.jl:188-197)
        for (int e = tid; e < SJ; e += nt) {
            int jj = e, i = 0, jo = 0;
            while (jj >= tp.J[ao + i]) { jj -= tp.J[ao + i]; jo += tp.J[ao + i]; ++i; }
            double cs = 0.0;
            for (int j = 0; j < tp.J[ao + i]; ++j) cs += gam[jo + j];
            const double el = dev_digamma_ar(gam[e]) - dev_digamma_ar(cs);
            q.Elnphi[mg + k * SJ + e] = el;
            if (rows) sh_eln[e] = el;
        }
        __syncthreads();
        const double* eln = rows ? sh_eln : q.Elnphi + mg + (size_t)k * SJ;
        // effective [k][v] tables: Eeff = sum_i Elnphi[..][f_vi]; phieff = prod_i gamma[..][f_vi] / sum_j gamma[..][j]
        for (int v = tid; v < Vm; v += nt) {
            double se = 0.0, pp = 1.0; int jo = 0;
            for (int i = 0; i < nf; ++i) {
                const int Ji = tp.J[ao + i], f = (staged ? sh_feat[i * Vm + v] : feat[i * Vm + v]);
                double cs = 0.0;
                for (int j = 0; j < Ji; ++j) cs += gam[jo + j];
                se += eln[jo + f];
                pp *= gam[jo + f] / cs;
                jo += Ji;
            }
            q.Eeff[go + k * Vm + v] = se; q.expEeff[go + k * Vm + v] = ar_exp(se); q.phieff[go + k * Vm + v] = pp;
        }
    }
}

// update_α! (MMCTM.jl:252-269 / IMMCTM.jl:225-244): one block per Dirichlet parameter α[m] (MMCTM) / α[m][i] (IMMCTM);
// the block sums Elnϕ over the K_m topics and the V_m (J_mi) values, lane 0 runs the 1-D LD_MMA maximisation of
// α_objective (common.jl:38-46) from the current α with lower bound 1e-7 and xtol_rel = xtol_abs = 1e-5.
__global__ __launch_bounds__(64) void k_ctm_update_alpha(CtmDims dm, CtmTopics tp, const double* Elnphi, double* alpha, int GM, int nalpha,
                                                         int xtol_rule, int max_eval, const int* active)
{
    if (active && !active[blockIdx.y]) return;
    Elnphi += (size_t)blockIdx.y * GM; alpha += (size_t)blockIdx.y * nalpha;
    const int lane = threadIdx.x, a = blockIdx.x;
    int m = 0, i = 0, Km, n, stride, base;
    if (!tp.immctm) { m = a; Km = dm.K[m]; n = dm.V[m]; stride = n; base = dm.goff[m]; }
    else {
        while (a >= tp.aoff[m + 1]) ++m;
        i = a - tp.aoff[m];
        int jo = 0;
        for (int q = 0; q < i; ++q) jo += tp.J[tp.aoff[m] + q];
        Km = dm.K[m]; n = tp.J[a]; stride = tp.SJ[m]; base = tp.mgoff[m] + jo;
    }
    double s = 0.0;
    for (int j = lane; j < n; j += 64) { double c = 0.0; for (int k = 0; k < Km; ++k) c += Elnphi[base + k * stride + j]; s += c; }
    s = wave_sum(s);
    if (lane != 0) return;
    const double K = Km, V = n, lb = 1e-7, xtol = 1e-5;
    // minimise f = -α_objective; m = 0 constraints, one coordinate, sigma = 1 (infinite upper bound)
    auto eval = [&](double x, double& g) {
        g = -(K * V * (dev_digamma(V * x) - dev_digamma(x)) + s);
        return -(K * (lgamma(V * x) - V * lgamma(x)) + x * s);
    };
    double x = alpha[a], sigma = 1.0, rho = 1.0, dfdx, dfdx_cur, xcur = x, xprev = x, xprevprev = x;
    double fbest = eval(x, dfdx), fcur = fbest;
    int nev = 1, k = 0;
    bool capped = false;
    for (;;) {
        if (nev >= max_eval) break;
        if (++k > 1) xprevprev = xprev;
        xprev = xcur;
        for (;;) {
            const double g = dfdx, sigma2 = sigma * sigma, u = g * sigma2, v = fabs(g) * sigma + 0.5 * rho;
            const double q = u / (v * sigma);
            double dx = (u / v) / (-1.0 - sqrt(fabs(1.0 - q * q)));
            double xc = x + dx;
            if (xc < lb) xc = lb;
            if (xc > x + 0.9 * sigma) xc = x + 0.9 * sigma; else if (xc < x - 0.9 * sigma) xc = x - 0.9 * sigma;
            xcur = xc;
            dx = xc - x;
            const double dx2 = dx * dx, denominv = 1.0 / (sigma2 - dx2);
            const double gval = fbest + (g * (sigma2 * dx) + (fabs(g) * sigma + 0.5 * rho) * dx2) * denominv;
            const double wval = 0.5 * dx2 * denominv;
            fcur = eval(xcur, dfdx_cur); ++nev;
            const bool inner_done = gval >= fcur;
            if (fcur < fbest) { fbest = fcur; x = xcur; dfdx = dfdx_cur; }
            if (nev >= max_eval) { capped = true; break; }
            if (inner_done) break;
            if (fcur > gval) rho = fmin(10.0 * rho, 1.1 * (rho + (fcur - gval) / wval));
        }
        if (capped) break;
        const double ad = fabs(xcur - xprev);
        bool stop;
        if (xtol_rule == 0) stop = (ad < xtol * fabs(xcur)) || (ad < xtol);
        else stop = ad < xtol || ad < xtol * (fabs(xcur) + fabs(xprev)) * 0.5 || xcur == xprev;
        if (stop) break;
        rho = fmax(0.1 * rho, 1e-5);
        if (k > 1) {
            const double d2 = (xcur - xprev) * (xprev - xprevprev);
            sigma *= d2 < 0 ? 0.7 : (d2 > 0 ? 1.2 : 1.0);
        }
    }
    alpha[a] = x;
}

// effective tables from UPLOADED topic fields (fit_heldout copies γ and Elnϕ, MMCTM.jl:561-562), one block per topic:
// Elnphi != NULL: Eeff / exp(Eeff) from it; gamma != NULL (IMMCTM, whose ll normalises γ itself, IMMCTM.jl:417-420): phieff
__global__ __launch_bounds__(256) void k_ctm_tables_from_Elnphi(CtmDims dm, CtmTopics tp, const double* Elnphi, double* Eeff, double* expEeff,
                                                                const double* gamma, double* phieff)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    int m = 0;
    while (m + 1 < dm.M && (int)blockIdx.x >= dm.koff[m + 1]) ++m;
    const int k = blockIdx.x - dm.koff[m];
    const int Vm = dm.V[m], go = dm.goff[m];
    if (!tp.immctm) {
        if (Elnphi) for (int v = tid; v < Vm; v += nt) { const double el = Elnphi[go + k * Vm + v]; Eeff[go + k * Vm + v] = el; expEeff[go + k * Vm + v] = ar_exp(el); }
    } else {
        const int mg = tp.mgoff[m], SJ = tp.SJ[m], nf = tp.nfeat[m], ao = tp.aoff[m];
        const int* feat = tp.features + tp.foff[m];
        for (int v = tid; v < Vm; v += nt) {
            double se = 0.0, pp = 1.0; int jo = 0;
            for (int i = 0; i < nf; ++i) {
                const int Ji = tp.J[ao + i], f = feat[i * Vm + v];
                if (Elnphi) se += Elnphi[mg + k * SJ + jo + f];
                if (gamma) {
                    double cs = 0.0;
                    for (int j = 0; j < Ji; ++j) cs += gamma[mg + k * SJ + jo + j];
                    pp *= gamma[mg + k * SJ + jo + f] / cs;
                }
                jo += Ji;
            }
            if (Elnphi) { Eeff[go + k * Vm + v] = se; expEeff[go + k * Vm + v] = ar_exp(se); }
            if (gamma) phieff[go + k * Vm + v] = pp;
        }
    }
}

// props = softmax(lambda block) (MMCTM.jl:145-154) and per-modality ll numerators (MMCTM.jl:384-418); wave per document.
// llpart[block][M]
// gauss != 0: the launch carries one extra block (the last) that runs update_μ!/update_Σ! of the same pass -- the ll needs
// only lambda and phi, the next E-step needs mu / Sigma^-1, so the 50 us single-block inversion hides behind the document sweep
template <bool TAB_LDS, int L>
__global__ __launch_bounds__(kBlockS) void k_ctm_loglik(CtmDev c, const double* lam, const double* phieff, double* props, double* llpart,
                                                        int compute_ll, const int* active, MstepArgs ga, int gauss)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesS][kMaxM];
    constexpr int G = MMM_WAVE / L;           // documents per wave: L >= sum K lanes each (coordinates for the softmax, terms for the sweep)
    const CtmDims& dm = c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D, GT = dm.GT;
    if (active && !active[blockIdx.y]) return;
    const int ndoc_blocks = gauss ? gridDim.x - 1 : gridDim.x;
    if (gauss && blockIdx.x == 0) {      // block 0: dispatched first, so the serial inversion starts with the sweep, not after it
        MstepPtrs q;
        if (mstep_replica(ga, q)) ctm_gauss_mstep<false>(ga, q, smem);
        return;
    }
    const int bx = (int)blockIdx.x - gauss;
    lam += (size_t)blockIdx.y * D * MK; phieff += (size_t)blockIdx.y * GT; llpart += (size_t)blockIdx.y * ndoc_blocks * M;
    if (props) props += (size_t)blockIdx.y * D * MK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / L, l = lane % L;
    const double* sP = TAB_LDS ? smem : phieff;                   // [GT]: staged, or (wide tables) read through L2
    double* sPr = smem + (TAB_LDS ? GT : 0) + wid * 64 + g * L;   // the group's props
    const double* sLog = smem + (((TAB_LDS ? GT : 0) + kWavesS * 64 + 1) & ~1);      // [256] log table (dev_log_tab), 16-byte aligned
    if (TAB_LDS && compute_ll) { for (int i = tid; i < GT; i += kBlockS) smem[i] = phieff[i]; }
    if (compute_ll && tid < MMM_LOGTAB_N) smem[(((TAB_LDS ? GT : 0) + kWavesS * 64 + 1) & ~1) + tid] = g_mmm_logtab[tid];
    __syncthreads();
    int mod_l = 0;
    for (int m = 0; m < M; ++m) if (l >= dm.koff[m] && l < dm.koff[m + 1]) mod_l = m;
    // per-lane partial sums over the lane's terms of all its documents; reduced over the wave once, at the end
    double acc[kMaxM];
    for (int m = 0; m < kMaxM; ++m) acc[m] = 0.0;
    for (int base = (bx * kWavesS + wid) * G; base < D; base += ndoc_blocks * kWavesS * G) {
        const int d = base + g;
        const bool valid = d < D, act = valid && l < MK;
        const double x = act ? lam[(size_t)d * MK + l] : 0.0;
        double pr = 0.0;
        for (int m = 0; m < M; ++m) {
            const bool in = act && mod_l == m;
            const double mx = group_max<L>(in ? x : -1e300);
            const double e = in ? exp(x - mx) : 0.0;
            const double s = group_sum<L>(e);
            if (in) pr = e / s;
        }
        if (act && props) props[(size_t)d * MK + l] = pr;
        if (!compute_ll) continue;
        lds_wave_sync();
        sPr[l] = pr;
        lds_wave_sync();
        for (int m = 0; m < M; ++m) {
            const int Km = dm.K[m], Vm = dm.V[m], off = dm.koff[m];
            const double* tb = sP + dm.goff[m];
            const int64_t* dp = c.doc_ptr + (size_t)m * (D + 1);
            const int64_t start = valid ? dp[d] : 0;
            const int W = valid ? (int)(dp[d + 1] - start) : 0;
            double a = 0.0;
            for (int w = l; w < W; w += L) {
                const int2 t = c.tc[start + w];
                double p = 0.0;
                for (int k = 0; k < Km; ++k) p = fma(sPr[off + k], tb[k * Vm + t.x], p);
                a += (double)t.y * dev_log_tab(p, sLog);
            }
            acc[m] += a;
        }
    }
    if (compute_ll) {
        for (int m = 0; m < M; ++m) { const double tot = wave_sum(acc[m]); if (lane == 0) shw[wid][m] = tot; }
        __syncthreads();
        if (tid < M) { double s = 0.0; for (int w = 0; w < kWavesS; ++w) s += shw[w][tid]; llpart[(size_t)bx * M + tid] = s; }
    }
}

// The same sweep over ROWS OF COUNTS (round 3; handles whose theta phase runs over them, k_ctm_theta_dense): 16 lanes per document
// whatever sum K is, four documents per wave step, one modality after the other; a lane reads its term slots' 16-bit counts (no
// doc_ptr -> (term, count) round trip), its term's phi column as 16-byte pairs from a term-major copy in LDS, and keeps the document's
// props in registers.  props and the ll numerators are outside the feedback loop of the fit (the next pass reads lambda, not props),
// so their sums may be associated as this layout likes: they agree with k_ctm_loglik to rounding.
struct DenseRows { const unsigned short* rows[kMaxM]; int SL[kMaxM]; int tpoff[kMaxM + 1]; };      // tpoff: prefix sums of 16 SL_m

template <int KMX>
__global__ __launch_bounds__(kBlockS) void k_ctm_loglik_dense(CtmDev c, const double* lam, const double* phieff, double* props, double* llpart,
                                                              int compute_ll, const int* active, MstepArgs ga, int gauss, DenseRows dr)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesS][kMaxM];
    constexpr int L = 16, G = MMM_WAVE / L;
    const CtmDims& dm = c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D, GT = dm.GT;
    if (active && !active[blockIdx.y]) return;
    const int ndoc_blocks = gauss ? gridDim.x - 1 : gridDim.x;
    if (gauss && blockIdx.x == 0) {      // block 0: update_μ! / update_Σ! of the same pass beside the sweep (see k_ctm_loglik)
        MstepPtrs q;
        if (mstep_replica(ga, q)) ctm_gauss_mstep<false>(ga, q, smem);
        return;
    }
    const int bx = (int)blockIdx.x - gauss;
    lam += (size_t)blockIdx.y * D * MK; phieff += (size_t)blockIdx.y * GT; llpart += (size_t)blockIdx.y * ndoc_blocks * M;
    if (props) props += (size_t)blockIdx.y * D * MK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / L, l = lane % L;
    const int VT = dr.tpoff[M];                                     // term slots of all modalities
    double* sPhi = smem;                                            // [VT][KMX] phi, term-major; slots beyond V_m hold 1, topics beyond K_m hold 0
    double* sPr = sPhi + (size_t)VT * KMX + ((size_t)wid * G + g) * KMX;      // the group's props
    const double* sLog = smem + (((size_t)VT * KMX + (size_t)kWavesS * G * KMX + 1) & ~(size_t)1);      // [256] log table, 16-byte aligned
    if (compute_ll) {
        for (int m = 0; m < M; ++m) {
            const int Km = dm.K[m], Vm = dm.V[m], Vp = 16 * dr.SL[m];
            const double* src = phieff + dm.goff[m];
            for (int i = tid; i < Vp * KMX; i += kBlockS) {
                const int v = i / KMX, k = i % KMX;
                sPhi[(size_t)dr.tpoff[m] * KMX + i] = (k < Km) ? (v < Vm ? src[(size_t)k * Vm + v] : 1.0) : 0.0;
            }
        }
        if (tid < MMM_LOGTAB_N) smem[(((size_t)VT * KMX + (size_t)kWavesS * G * KMX + 1) & ~(size_t)1) + tid] = g_mmm_logtab[tid];
    }
    if (tid < kWavesS * kMaxM) (&shw[0][0])[tid] = 0.0;
    __syncthreads();
    // The wave walks its document steps once per modality (modality-major: iteration it = m * nsteps + step), and the NEXT iteration's
    // lambda values and counts are requested while this one computes -- unconditional loads (clamped indices, masks when the values are
    // taken over), a lane's part of a row as one load of <= 4 words, uniform base + 32-bit offset, first use pinned behind the slot loop
    // (the rules of k_lda_estep_dense).  One exposed round trip per wave instead of one per (step, modality).
    const int wslot = bx * kWavesS + wid, nslots = ndoc_blocks * kWavesS;
    // (wave-uniform: through readfirstlane, so that the per-modality dimensions below are read with scalar loads)
    const int nsteps = __builtin_amdgcn_readfirstlane(wslot * G < D ? (D - wslot * G + nslots * G - 1) / (nslots * G) : 0);
    const int T = nsteps * M;
    double xq = 0.0;
    unsigned wq[4] = {0u, 0u, 0u, 0u};
    auto request = [&](int it, double& x, unsigned* w) {
        const int mm = it / nsteps, step = it - mm * nsteps;
        const int dd = (wslot + step * nslots) * G + g;
        const unsigned dl = dd < D ? (unsigned)dd : 0u;
        const int Kq = dm.K[mm], lk = l < Kq ? l : Kq - 1;
        x = *at_byte(lam, (dl * (unsigned)MK + (unsigned)(dm.koff[mm] + lk)) * 8u);
        const int sls = (dr.SL[mm] + 1) & ~1;
        // lane-major rows: the lane's <= 8 slots are the first words of one 16-byte load (what lies behind them is not used; the rows are
        // allocated with 16 bytes to spare)
        const unsigned* row = at_byte((const unsigned*)dr.rows[mm], (dl * 16u + (unsigned)l) * (unsigned)sls * 2u);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = row[j];
    };
    if (T > 0) request(0, xq, wq);
    double a_mod = 0.0;
    for (int it = 0; it < T; ++it) {
        const int m = it / nsteps, step = it - m * nsteps;
        const int d = (wslot + step * nslots) * G + g;
        const bool valid = d < D;
        const int Km = dm.K[m], off = dm.koff[m];
        const bool in = l < Km;
        const double x = (valid && in) ? xq : 0.0;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = valid ? wq[j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(w[j]));       // taken over before the props store below (a wait behind it would cover the store)
        // props = softmax(lambda block) (MMCTM.jl:145-154)
        const double mx = group_max<L>(in ? x : -1e300);
        const double e = in ? exp(x - mx) : 0.0;
        const double pr = e / group_sum<L>(e);
        if (valid && in && props) *at_byte(props, ((unsigned)d * (unsigned)MK + (unsigned)(off + l)) * 8u) = pr;
        if (!compute_ll) { if (it + 1 < T) request(it + 1, xq, wq); continue; }
        lds_wave_sync();
        if (l < KMX) sPr[l] = pr;
        lds_wave_sync();
        double tv[KMX];
#pragma unroll
        for (int k = 0; k < KMX; ++k) tv[k] = sPr[k];
        if (it + 1 < T) request(it + 1, xq, wq);
        const int SLm = dr.SL[m];
        const double* tbm = sPhi + ((size_t)dr.tpoff[m] + l) * KMX;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < SLm) {
                const unsigned cq = (q & 1) ? w[q / 2] >> 16 : w[q / 2] & 0xffffu;
                const double* tb = tbm + (size_t)q * L * KMX;
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int k = 0; k + 1 < KMX; k += 2) { p0 = fma(tv[k], tb[k], p0); p1 = fma(tv[k + 1], tb[k + 1], p1); }
                if (KMX & 1) p0 = fma(tv[KMX - 1], tb[KMX - 1], p0);
                a = fma((double)cq, dev_log_tab(p0 + p1, sLog), a);       // a slot without count: 0 x log(p), p > 0
            }
        }
        a_mod += a;
        asm volatile("" : "+v"(a_mod) :: "memory");
        asm volatile("" : "+v"(xq) :: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(wq[j]) :: "memory");
        if (step == nsteps - 1) {
            const double tot = wave_sum(a_mod);
            if (lane == 0) shw[wid][m] = tot;
            a_mod = 0.0;
        }
    }
    if (compute_ll) {
        __syncthreads();
        if (tid < M) { double sm = 0.0; for (int w = 0; w < kWavesS; ++w) sm += shw[w][tid]; llpart[(size_t)bx * M + tid] = sm; }
    }
}

__global__ __launch_bounds__(64) void k_sum_columns(const double* part, int n, int stride, double* out, size_t out_stride, const int* active)
{
    const int j = blockIdx.x;
    if (active && !active[blockIdx.y]) return;
    part += (size_t)blockIdx.y * n * stride; out += (size_t)blockIdx.y * out_stride;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += part[(size_t)i * stride + j];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[j] = acc;
}

// k_sum_columns + k_ll_store in one launch (single GPU: no exchange between them)
// The stopping rule of fit! on the device (MMCTM.jl:481-489 + common.jl:48-51: after > 10 rows, stop when the largest relative change
// of the per-modality ll is < tol; a NaN propagates like Julia's `maximum` and never stops): the replica's `active` flag is
// cleared, every later launch of the fit skips the replica, and the host -- which reads the flags one pass late, so that it never
// stalls the stream -- stops enqueueing when none is left.  npass counts the rows a replica has written.
struct StopArgs { int enable; double tol; int* active_w; int* npass; };

__device__ __forceinline__ void ll_stop_rule(const StopArgs& st, int rep, int M, const double* row)
{
    if (st.npass) st.npass[rep] += 1;
    if (!st.enable || !st.active_w) return;
    double rel = 0.0;
    for (int q = 0; q < M; ++q) {
        const double a = row[q - M], b = row[q];          // previous row, this row
        const double rr = fabs(a - b) / fabs(b);
        if (rr > rel || rr != rr) rel = rr;
    }
    if (rel < st.tol) st.active_w[rep] = 0;
}

// grid (1, replicas), one wave per modality: column sums of the ll partials, division by N_m, history row, stopping rule
__global__ __launch_bounds__(64 * kMaxM) void k_ll_finish(const double* part, int n, int M, const double* Nm, double* num, size_t num_stride, double* dst,
                                                          size_t dst_stride, const int* active, StopArgs st)
{
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (active && !active[blockIdx.y]) return;
    part += (size_t)blockIdx.y * n * M;
    double acc = 0.0;
    for (int i = lane; i < n; i += 64) acc += part[(size_t)i * M + j];
    acc = wave_sum(acc);
    if (lane == 0) { num[blockIdx.y * num_stride + j] = acc; dst[blockIdx.y * dst_stride + j] = acc / Nm[j]; }
    __syncthreads();
    if (threadIdx.x == 0) ll_stop_rule(st, blockIdx.y, M, dst + blockIdx.y * dst_stride);
}

__global__ void k_ll_store(int M, const double* num, size_t num_stride, const double* Nm, double* dst, size_t dst_stride, const int* active, StopArgs st)
{
    if (active && !active[blockIdx.y]) return;
    if ((int)threadIdx.x < M) dst[blockIdx.y * dst_stride + threadIdx.x] = num[blockIdx.y * num_stride + threadIdx.x] / Nm[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) ll_stop_rule(st, blockIdx.y, M, dst + blockIdx.y * dst_stride);
}

// per-document ELBO pieces (MMCTM.jl:286-370): out[block][6] = {ElnPeta(without logdet/const), ElnPZ, ElnPX, ElnQeta, ElnQZ, count}
// theta is rebuilt on the fly from (lam_prev, expE_prev) when theta == NULL
template <bool TAB_LDS>
__global__ __launch_bounds__(kBlockS) void k_ctm_elbo_docs(CtmDev c, const double* invSigma, const double* mu, const double* lam, const double* nu,
                                                           const double* zeta, const double* theta, const double* Eeff, double* out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesS][5];
    const CtmDims& dm = c.dm;
    const int MK = dm.MK, M = dm.M, D = dm.D, GT = dm.GT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double* sS = smem; double* sEl = sS + MK * MK; double* scr = sEl + (TAB_LDS ? GT : 0) + wid * 64;
    const double* sE = TAB_LDS ? sEl : Eeff;
    for (int i = tid; i < MK * MK; i += kBlockS) sS[i] = invSigma[i];
    if (TAB_LDS) for (int i = tid; i < GT; i += kBlockS) sEl[i] = Eeff[i];
    __syncthreads();
    int mod_l = 0;
    for (int m = 0; m < M; ++m) if (lane >= dm.koff[m] && lane < dm.koff[m + 1]) mod_l = m;
    double t[5] = {0, 0, 0, 0, 0};
    for (int d = blockIdx.x * kWavesS + wid; d < D; d += gridDim.x * kWavesS) {
        const bool act = lane < MK;
        const double x = act ? lam[(size_t)d * MK + lane] : 0.0, v = act ? nu[(size_t)d * MK + lane] : 1.0;
        const double diff = act ? x - mu[lane] : 0.0;
        lds_wave_sync(); scr[lane] = diff; lds_wave_sync();
        double Sd = 0.0;
        if (act) for (int j = 0; j < MK; ++j) Sd = fma(sS[j * MK + lane], scr[j], Sd);
        // ElnPη without the constants: -1/2 (tr(diag(ν) invΣ) + diff' invΣ diff)   (MMCTM.jl:286-300)
        t[0] += wave_sum(act ? -0.5 * (v * sS[lane * MK + lane] + diff * Sd) : 0.0);
        // ElnQη without the constant: -1/2 Σ log ν   (MMCTM.jl:352-358)
        t[3] += wave_sum(act ? -0.5 * log(v) : 0.0);
        // ElnPZ (MMCTM.jl:302-316), ElnPX (318-336), ElnQZ (360-370)
        const double Nl = act ? c.Ndm[(size_t)d * M + mod_l] : 0.0;
        const double zl = act ? zeta[(size_t)d * M + mod_l] : 1.0;
        double pz = act ? -(Nl / zl) * exp(x + 0.5 * v) : 0.0;
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
                    lin += n * p * lam[(size_t)d * MK + off + k];       // Σ λ_i sumθ_i
                    px += n * p * sE[dm.goff[m] + k * Vm + tc.x];
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

// topic-side ELBO pieces (MMCTM.jl:271-284,338-350; IMMCTM.jl:247-262,316-330) and logdet(invSigma): out = {ElnPphi, ElnQphi, logdet}
__global__ __launch_bounds__(256) void k_ctm_elbo_topics(CtmDims dm, CtmTopics tp, const double* gamma, const double* Elnphi, const double* invSigma, double* out,
                                                         double* big_scratch)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double s_logdet; __shared__ int s_sing, s_piv; __shared__ double sh[4];
    const int MK = dm.MK, M = dm.M, tid = threadIdx.x, nt = blockDim.x;
    if (big_scratch) {       // sum K > 64: device memory (separate code paths keep the LDS addressing of the other)
        double* A = big_scratch; double* Ai = A + MK * MK;
        for (int e = tid; e < MK * MK; e += nt) { const int i = e % MK, j = e / MK; A[i * MK + j] = invSigma[e]; }
        __syncthreads();
        block_inverse_big(MK, A, Ai, &s_logdet, &s_sing);
    } else {
        double* A = smem; double* Ai = smem + MK * MK;
        for (int e = tid; e < MK * MK; e += nt) { const int i = e % MK, j = e / MK; A[i * MK + j] = invSigma[e]; }
        __syncthreads();
        block_inverse(MK, A, Ai, &s_logdet, &s_sing, &s_piv);
    }
    __syncthreads();
    // one (m,k[,i]) Dirichlet per loop trip, handled by the whole block
    double P = 0.0, Q = 0.0;
    for (int m = 0; m < M; ++m) {
        const int Km = dm.K[m];
        const int nblk = tp.immctm ? tp.nfeat[m] : 1;
        for (int k = 0; k < Km; ++k) {
            int jo = 0;
            for (int i = 0; i < nblk; ++i) {
                const int n = tp.immctm ? tp.J[tp.aoff[m] + i] : dm.V[m];
                const int base = tp.immctm ? tp.mgoff[m] + k * tp.SJ[m] + jo : dm.goff[m] + k * n;
                const double al = tp.immctm ? tp.alpha[tp.aoff[m] + i] : tp.alpha[m];
                double se = 0.0, sg = 0.0, lg = 0.0, ge = 0.0;
                for (int v = tid; v < n; v += nt) {
                    const double gm = gamma[base + v], el = Elnphi[base + v];
                    se += el; sg += gm; lg += lgamma(gm); ge += (gm - 1.0) * el;
                }
                double vals[4] = {se, sg, lg, ge};
                for (int q = 0; q < 4; ++q) {
                    double w = wave_sum(vals[q]);
                    __syncthreads();
                    if ((tid & 63) == 0) sh[tid >> 6] = w;
                    __syncthreads();
                    vals[q] = sh[0] + sh[1] + sh[2] + sh[3];
                }
                // ElnPϕ: -(n lgamma(α) - lgamma(n α)) + (α-1) Σ Elnϕ ; ElnQϕ: -(Σ lgamma(γ) - lgamma(Σγ)) + Σ (γ-1) Elnϕ
                P += -((double)n * lgamma(al) - lgamma((double)n * al)) + (al - 1.0) * vals[0];
                Q += -(vals[2] - lgamma(vals[1])) + vals[3];
                jo += n;
            }
        }
    }
    if (tid == 0) { out[0] = P; out[1] = Q; out[2] = s_logdet; }
}

__global__ void k_fill(double* p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// gamma statistics from a resident theta (stage update_γ!): sums[goff+k*V+v] += theta n, global f64 atomics
__global__ void k_ctm_gamma_from_theta(CtmDev c, int m, const double* theta, double* sums)
{
    const CtmDims& dm = c.dm;
    const int64_t e0 = dm.estart[m];
    const int64_t e1 = c.doc_ptr[(size_t)m * (dm.D + 1) + dm.D];
    const int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= e1) return;
    const int2 t = c.tc[e];
    const int Km = dm.K[m], Vm = dm.V[m];
    for (int k = 0; k < Km; ++k)
        unsafeAtomicAdd(&sums[dm.goff[m] + k * Vm + t.x], theta[dm.toff[m] + (size_t)(e - e0) * Km + k] * (double)t.y);
}

// copy n doubles per replica, active replicas only (grid.y = replicas)
__global__ void k_copy_rep(double* dst, const double* src, size_t n, const int* active)
{
    if (active && !active[blockIdx.y]) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[blockIdx.y * n + i] = src[blockIdx.y * n + i];
}

} // namespace

// =====================================================================================================================
// A handle holds R >= 1 independent models ("replicas": the restarts of scripts/run_mmctm.jl:77-134) over ONE resident
// corpus.  Every per-model array is R contiguous copies; batched launches put the replica on grid.y.  The classic
// single-model API works on the selected replica (`sel`, 0 by default).
struct mmm_ctm {
    mmm_ctx* ctx = nullptr;
    mmm_tuning_opts tune{};        // the caller's choices at create time (mmm_ctx_set_tuning)
    CtmDims dm{};
    CtmTopics tp{};
    bool immctm = false;
    int R = 1, sel = 0;
    int L = 64, GM = 0 /* model-layout gamma size */;
    int Ls = 64;                   // lanes per document in the solve phase: L, or sum K for the packed builds (6 / 12), or 2 / 4 (cpl > 1)
    int cpl = 1;                   // coordinates per lane in the solve phase (k_ctm_solve_cpl: sum K = 10, 14, 28)
    int lam_occ = 4;               // waves per SIMD of the persistent solve build (3 for sum K = 28)
    bool persist = false;          // solve phase by k_ctm_solve_cpl (persistent waves, document slots refilled): cpl > 1, or cpl = 1 with Ls = L
    int64_t nnz = 0, theta_n = 0;
    long long nnzm[kMaxM] = {0};
    double Dglobal = 0;
    SolveOpts opt{};
    DevBuf<int64_t> doc_ptr; DevBuf<int2> tc; DevBuf<double> Ndm; DevBuf<int> features; DevBuf<double> alpha;
    DevBuf<double> lambda, lambda_prev, nu, zeta, props, sumth;                 // [R][D*MK] / [R][D*M]
    DevBuf<double> theta;                                                        // ONE replica (the selected one), on demand
    DevBuf<double> mu, Sigma, invSigma, gamma, Elnphi, phi, Eeff, expEeff, expEeff_prev, phieff;   // [R][...]
    DevBuf<double> partial, mompart, stats, llpart, llnum, Nm, elbopart, ll_hist;
    DevBuf<int> nev_nu, nev_lam, status, active, npass;
    // dense corpora: the fused pass's theta phase over rows of 16-bit counts, one launch per modality (k_ctm_theta_dense)
    bool tdense = false; int tSL[kMaxM] = {0};
    DevBuf<unsigned short> trows[kMaxM];      // [D][16][SL_m rounded up to even]: lane-major rows of counts (a lane's part of a row is one load)
    bool big = false;              // 64 < sum K <= 256: the generic kernels of ctm_big.cuh (one wave per document, several coordinates per lane)
    DevBuf<double> big_scratch;    // [R][2 MK^2]: Sigma and its inverse during the Gaussian M-step / the ELBO's logdet
    int stop_enable = 0; double stop_tol = 0.0;     // set by fit_scope around a pass: the ll kernels apply the stopping rule
    int* pin_flags = nullptr;                       // pinned [2][2R]: snapshots of (active | status) the host reads one pass late
    std::vector<int> h_active, n_hist;        // per replica
    // theta (the largest array) exists once, not per replica.  Per replica we know how to rebuild it:
    // 0 = constructor value 1/K, 1 = update_θ! on (lambda_prev, expEeff_prev) -- the theta the last pass used --,
    // 2 = explicit (update_θ! stage call or upload; parked in theta_spill[r] when another replica takes the buffer)
    std::vector<char> theta_state;
    std::vector<std::unique_ptr<DevBuf<double>>> theta_spill;
    int theta_rep = -1;                       // replica whose theta is in the theta buffer (-1: none)
    int cap_hist = 0;
    int grid_e = 1, waves_e = 8, grid_s = 1, grid_m = 1, grid_v = 1, waves_s = 4;
    // wide tables (sum_m K_m V_m beyond LDS): theta phase without table / slabs in LDS + k_ctm_stats_terms over posting lists
    bool wide = false;
    int stats_waves = 1, nterms = 0;
    DevBuf<int64_t> term_ptr;      // [sum V + 1], modality-major
    DevBuf<int2> tpost;            // (document, count) per posting
    DevBuf<double> aexp;           // [R][D][MK]
    int nmom = 0, nalpha = 0; size_t s_stats = 0, s_llnum = 0;
    std::vector<double> hNm;
    CtmDev dev() const { return CtmDev{dm, doc_ptr.p, tc.p, Ndm.p}; }
    size_t sDMK() const { return (size_t)dm.D * dm.MK; }
};

namespace {

// which replicas a launch covers: one (stage API, on the selected replica) or all active ones (batched fit)
struct Scope { int rep0, nrep; const int* active; };
inline Scope one(const mmm_ctm* m) { return Scope{m->sel, 1, nullptr}; }
inline Scope all(const mmm_ctm* m) { return Scope{0, m->R, m->active.p}; }

template <int L, int PH, int MKT = 0, int KMX = 16, int OCC = 4, bool WIDE = false, bool PACK = false>
int launch_estep_L(mmm_ctm* m, const CtmEArgs& a, size_t lds, int grid, int waves, int nrep)
{
    mmm_ctx* ctx = m->ctx;
    auto k = k_ctm_estep<L, PH, MKT, KMX, OCC, WIDE, PACK>;
    if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid, nrep), dim3(waves * MMM_WAVE), lds, ctx->stream, a);
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

size_t estep_lds(const mmm_ctm* m, int flags)     // theta phase
{
    const int G = MMM_WAVE / m->L;
    if (m->wide) return sizeof(double) * (size_t)m->waves_e * G * 2 * m->L;
    size_t n = (size_t)m->dm.GT + (size_t)m->waves_e * G * ((flags & F_SLAB) ? 1 : 2) * m->L;      // (the fused pass: one scratch row per group)
    int slabn = 0;      // a wave's slab holds one modality at a time
    for (int i = 0; i < m->dm.M; ++i) slabn = std::max(slabn, m->dm.K[i] * m->dm.V[i]);
    if (flags & F_SLAB) n += (size_t)m->waves_e * slabn;
    return n * sizeof(double);
}

size_t solve_lds(const mmm_ctm* m)
{
    if (m->persist) {       // [MK][Ls * CPLP] padded invSigma + [waves][64 / Ls][MK + 2] difference vectors
        const int cplp = m->cpl == 1 ? 1 : (m->cpl + 1) & ~1;
        return sizeof(double) * ((size_t)m->dm.MK * m->Ls * cplp + (size_t)m->waves_s * (MMM_WAVE / m->Ls) * (m->dm.MK + 2) + (size_t)m->Ls * m->cpl);
    }
    const int scrw = m->Ls != m->L ? (MMM_WAVE / m->Ls + 1) * 2 * m->Ls : 2 * MMM_WAVE;
    return sizeof(double) * ((size_t)m->dm.MK * m->dm.MK + m->dm.MK + (size_t)m->waves_s * scrw);
}

size_t theta_dense_lds(const mmm_ctm* m, int i, int kmx)
{
    const int NW = m->waves_e;
    return sizeof(double) * ((size_t)16 * m->tSL[i] * kmx + (size_t)NW * 16 * m->tSL[i] * kmx + (size_t)NW * 4 * kmx + (size_t)NW * MMM_WAVE * kmx);
}

inline int theta_dense_kmx(int Km) { return Km <= 8 ? 8 : (Km <= 10 ? 10 : 16); }

// the theta phase of the fused pass over rows of counts: one launch per modality (k_ctm_theta_dense)
int launch_theta_dense(mmm_ctm* m, const CtmEArgs& a, int nrep)
{
    mmm_ctx* ctx = m->ctx;
    for (int i = 0; i < m->dm.M; ++i) {
        const int kmx = theta_dense_kmx(m->dm.K[i]);
        const size_t lds = theta_dense_lds(m, i, kmx);
        auto go = [&](auto kern) -> int {
            if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(m->grid_e, nrep), dim3(m->waves_e * MMM_WAVE), lds, ctx->stream, a, i, (const unsigned short*)m->trows[i].p);
            MMM_LAUNCH_CHECK(ctx);
            return MMM_OK;
        };
        int rc = MMM_ERR_UNSUPPORTED;
        switch (kmx * 10 + m->tSL[i]) {
            case 82: rc = go(k_ctm_theta_dense<8, 2>); break;    case 83: rc = go(k_ctm_theta_dense<8, 3>); break;
            case 86: rc = go(k_ctm_theta_dense<8, 6>); break;    case 88: rc = go(k_ctm_theta_dense<8, 8>); break;
            case 102: rc = go(k_ctm_theta_dense<10, 2>); break;  case 103: rc = go(k_ctm_theta_dense<10, 3>); break;
            case 106: rc = go(k_ctm_theta_dense<10, 6>); break;
            case 162: rc = go(k_ctm_theta_dense<16, 2>); break;  case 163: rc = go(k_ctm_theta_dense<16, 3>); break;
        }
        if (rc) return rc == MMM_ERR_UNSUPPORTED ? mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "no dense theta build for K = %d, %d slots", m->dm.K[i], m->tSL[i]) : rc;
    }
    return MMM_OK;
}


template <int PH>
int launch_phase(mmm_ctm* m, const CtmEArgs& a, size_t lds, int grid, int waves, int nrep)
{
    if (m->big) {      // sum K > 64: the generic kernels (ctm_big.cuh), one wave per document
        mmm_ctx* ctx = m->ctx;
        const int nblk = std::max(1, std::min((m->dm.D + 3) / 4, mmm_geo_cus(ctx) * 4));
        if constexpr (PH == 0) {
            const size_t l = sizeof(double) * 4 * (64 + 64 * 64);
            MMM_HIP(ctx, hipFuncSetAttribute((const void*)k_ctm_theta_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l));
            hipLaunchKernelGGL(k_ctm_theta_big, dim3(nblk, nrep), dim3(256), l, ctx->stream, a);
        } else {
            const size_t l = sizeof(double) * 4 * (size_t)m->dm.MK;
            hipLaunchKernelGGL(k_ctm_solve_big, dim3(nblk, nrep), dim3(256), l, ctx->stream, a);
        }
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    if constexpr (PH == 1) {      // solve phase: compile-time sum K for the shapes of the BASELINE configs (cfg 5 / 3 / 4)
        const bool small = (int64_t)grid * waves * nrep <= (int64_t)3 * 4 * mmm_geo_cus(m->ctx);      // cannot fill 4 waves per SIMD anyway
        if (m->persist) {          // persistent waves with refilled document slots (several coordinates per lane, or one)
            mmm_ctx* ctx = m->ctx;
            auto go = [&](auto kern) -> int {
                if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(grid, nrep), dim3(waves * MMM_WAVE), lds, ctx->stream, a);
                MMM_LAUNCH_CHECK(ctx);
                return MMM_OK;
            };
            // builds: sum K = 10 (BASELINE config 5) 2 lanes x 5 coordinates at 2 waves per SIMD, the chains of the coordinates interleaved by the
            // scheduler (245 VGPRs, no scratch); sum K = 28 (config 4) 16 lanes, 14 of them x 2 coordinates, at 3 waves per SIMD.  The other
            // layouts that were built and measured -- 8 x 4 and 32 x 1 for sum K = 28, 2 x 7 and 8 x 2 for sum K = 14, the two solves as two
            // launches with documents claimed on demand -- did not beat these and are gone from the source.
            if (m->dm.MK == 10 && m->Ls == 2) return go(k_ctm_solve_cpl<10, 2, 2, false>);
            if (m->dm.MK == 28 && m->Ls == 16) return go(k_ctm_solve_cpl<28, 16, 3, false>);
            return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "no multi-coordinate solve build for sum K = %d", m->dm.MK);
        }
        if (m->Ls != m->L) {       // packed groups: sum K lanes per document
            if (m->Ls == 6) return launch_estep_L<16, PH, 6, 16, 4, false, true>(m, a, lds, grid, waves, nrep);
            if (m->Ls == 10) return launch_estep_L<16, PH, 10, 16, 4, false, true>(m, a, lds, grid, waves, nrep);
            if (m->Ls == 12) return launch_estep_L<16, PH, 12, 16, 4, false, true>(m, a, lds, grid, waves, nrep);
            return mmm_fail(m->ctx, MMM_ERR_UNSUPPORTED, "no packed solve build for sum K = %d", m->Ls);
        }
        if (m->L == 16 && m->dm.MK == 10) return small ? launch_estep_L<16, PH, 10, 16, 3>(m, a, lds, grid, waves, nrep) : launch_estep_L<16, PH, 10>(m, a, lds, grid, waves, nrep);
        if (m->L == 16 && m->dm.MK == 14) return small ? launch_estep_L<16, PH, 14, 16, 3>(m, a, lds, grid, waves, nrep) : launch_estep_L<16, PH, 14>(m, a, lds, grid, waves, nrep);
        if (m->L == 32 && m->dm.MK == 28) return small ? launch_estep_L<32, PH, 28, 16, 3>(m, a, lds, grid, waves, nrep) : launch_estep_L<32, PH, 28>(m, a, lds, grid, waves, nrep);
    }
    if constexpr (PH == 0) {      // theta phase: a modality with more than 16 topics takes the build unrolled to 32
        int kmax = 0;
        for (int i = 0; i < m->dm.M; ++i) kmax = std::max(kmax, m->dm.K[i]);
        if (m->wide) {
            // 33..64 topics in one modality (sum K <= 64, so L = 64): the 64-topic build -- correct, far from tuned (its four 64-entry
            // register arrays live in scratch); the reference has no limit (MMCTM.jl:29-91)
            if (kmax > 32) return launch_estep_L<64, PH, 0, 64, 4, true>(m, a, lds, grid, waves, nrep);
            if (kmax > 16) return m->L == 32 ? launch_estep_L<32, PH, 0, 32, 4, true>(m, a, lds, grid, waves, nrep) : launch_estep_L<64, PH, 0, 32, 4, true>(m, a, lds, grid, waves, nrep);
            if (m->L == 16) return launch_estep_L<16, PH, 0, 16, 4, true>(m, a, lds, grid, waves, nrep);
            if (m->L == 32) return launch_estep_L<32, PH, 0, 16, 4, true>(m, a, lds, grid, waves, nrep);
            return launch_estep_L<64, PH, 0, 16, 4, true>(m, a, lds, grid, waves, nrep);
        }
        if (kmax > 16) return m->L == 32 ? launch_estep_L<32, PH, 0, 32>(m, a, lds, grid, waves, nrep) : launch_estep_L<64, PH, 0, 32>(m, a, lds, grid, waves, nrep);
        // the topic loops are unrolled to KMX: builds with KMX = 10 / 8 for the BASELINE shapes (K = [10,10,8], [10], [7,7]) instead of 16 --
        // the padded topics cost instructions (a product, two sums, a select and an exec-masked atomic each), not results
        const bool kfit = !mmm_off(m->tune, MMM_OFF_CTM_KFIT);
        if (kfit && kmax <= 8 && m->L == 16) return launch_estep_L<16, PH, 0, 8>(m, a, lds, grid, waves, nrep);
        if (kfit && kmax <= 10 && m->L == 16) return launch_estep_L<16, PH, 0, 10>(m, a, lds, grid, waves, nrep);
        if (kfit && kmax <= 10 && m->L == 32) return launch_estep_L<32, PH, 0, 10>(m, a, lds, grid, waves, nrep);
    }
    if (m->L == 16) return launch_estep_L<16, PH>(m, a, lds, grid, waves, nrep);
    if (m->L == 32) return launch_estep_L<32, PH>(m, a, lds, grid, waves, nrep);
    return launch_estep_L<64, PH>(m, a, lds, grid, waves, nrep);
}

// lam_in / expE: per-replica arrays (base of replica 0); lam_out likewise (may alias lam_in: in-place update)
// the side stream of the context (created on first use), or NULL if the runtime refuses
hipStream_t side_stream(mmm_ctx* ctx)
{
    if (ctx->side) return ctx->side;
    hipStream_t s = nullptr; hipEvent_t a = nullptr, b = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipEventDestroy(a);
        (void)hipStreamDestroy(s);
        return nullptr;
    }
    ctx->side = s; ctx->ev_fork = a; ctx->ev_join = b;
    return s;
}

// launches on the context go to another stream while this lives
struct StreamSwap {
    mmm_ctx* ctx; hipStream_t saved;
    StreamSwap(mmm_ctx* c, hipStream_t s) : ctx(c), saved(c->stream) { c->stream = s; }
    ~StreamSwap() { ctx->stream = saved; }
};

int run_estep(mmm_ctm* m, Scope sc, int flags, const double* lam_in, double* lam_out, const double* expE, double* lam_keep = nullptr,
              double* expE_keep = nullptr, bool fork_after_theta = false)
{
    const CtmDims& dm = m->dm;
    const size_t r0 = sc.rep0, DMK = m->sDMK(), MK = dm.MK;
    CtmEArgs a{m->dev(), m->invSigma.p + r0 * MK * MK, m->mu.p + r0 * MK, expE ? expE + r0 * dm.GT : nullptr, lam_in + r0 * DMK,
               lam_out ? lam_out + r0 * DMK : nullptr, m->nu.p + r0 * DMK, m->zeta.p + r0 * dm.D * dm.M,
               (flags & (F_THETA_STORED | F_THETA_STORE)) ? m->theta.p : nullptr, m->sumth.p + r0 * DMK,
               m->wide ? nullptr : m->partial.p + r0 * m->grid_e * dm.GT, m->wide ? m->aexp.p + r0 * dm.D * MK : nullptr,
               m->nev_nu.p + r0 * dm.D, m->nev_lam.p + r0 * dm.D, m->opt, flags, sc.active,
               lam_keep ? lam_keep + r0 * DMK : nullptr, expE_keep ? expE_keep + r0 * dm.GT : nullptr};
    int rc;
    if (flags & (F_ZETA | F_THETA_COMPUTE | F_THETA_STORED | F_SLAB)) {
        const size_t lds = estep_lds(m, flags);
        if (lds > 160 * 1024) return mmm_fail(m->ctx, MMM_ERR_UNSUPPORTED, "CTM theta phase needs %zu B of LDS (> 160 KiB)", lds);
        ProfSpan span(m->ctx, 1);   // mmm_ctx_profile_select(1): theta phase
        if (m->tdense && (flags & F_SLAB) && (flags & F_THETA_COMPUTE) && !(flags & (F_THETA_STORE | F_THETA_STORED))) {
            if ((rc = launch_theta_dense(m, a, sc.nrep))) return rc;
        } else if ((rc = launch_phase<0>(m, a, lds, m->grid_e, m->waves_e, sc.nrep))) return rc;
    }
    if (fork_after_theta) MMM_HIP(m->ctx, hipEventRecord(m->ctx->ev_fork, m->ctx->stream));
    if (flags & (F_NU | F_LAMBDA)) {
        ProfSpan span(m->ctx);      // mmm_ctx_profile_*: event pair around the dominant kernel (the two LD_MMA solves)
        if ((rc = launch_phase<1>(m, a, solve_lds(m), m->grid_v, m->waves_s, sc.nrep))) return rc;
    }
    return MMM_OK;
}

// part: [R][nslab][n] -> out: per replica at out + r*out_stride
int reduce_partials(mmm_ctm* m, Scope sc, const double* part, int nslab, int n, double* out, size_t out_stride)
{
    hipLaunchKernelGGL(k_reduce_partials, dim3((n + 15) / 16, sc.nrep), dim3(16, 64), 0, m->ctx->stream, part + (size_t)sc.rep0 * nslab * n, nslab, n,
                       out + sc.rep0 * out_stride, out_stride, sc.active);
    MMM_LAUNCH_CHECK(m->ctx);
    return MMM_OK;
}

MstepArgs mstep_args(mmm_ctm* m, Scope sc, int do_mu, int do_sigma, int do_gamma, int gamma_from_stats)
{
    const size_t r0 = sc.rep0, MK = m->dm.MK, GT = m->dm.GT, GM = m->GM;
    MstepArgs a{m->dm, m->tp, m->stats.p + r0 * m->s_stats, m->Dglobal, m->mu.p + r0 * MK, m->Sigma.p + r0 * MK * MK, m->invSigma.p + r0 * MK * MK,
                m->gamma.p + r0 * GM, m->Elnphi.p + r0 * GM, m->immctm ? nullptr : m->phi.p + r0 * GM, m->Eeff.p + r0 * GT, m->expEeff.p + r0 * GT,
                m->phieff.p + r0 * GT, m->status.p + r0, do_mu, do_sigma, do_gamma, gamma_from_stats, m->s_stats, m->GM, sc.active, m->nalpha,
                m->big ? m->big_scratch.p + r0 * 2 * MK * MK : nullptr};
    a.tp.alpha += r0 * m->nalpha;      // host-side copy of the argument struct: fine
    return a;
}

int run_mstep(mmm_ctm* m, Scope sc, int do_mu, int do_sigma, int do_gamma, int gamma_from_stats)
{
    mmm_ctx* ctx = m->ctx;
    const size_t MK = m->dm.MK;
    const MstepArgs a = mstep_args(m, sc, do_mu, do_sigma, do_gamma, gamma_from_stats);
    const size_t lds = m->big ? 0 : sizeof(double) * 2 * MK * MK;
    if (do_mu || do_sigma) {
        if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)k_ctm_mstep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_ctm_mstep, dim3(1, sc.nrep), dim3(256), lds, ctx->stream, a);
    }
    if (do_gamma) hipLaunchKernelGGL(k_ctm_mstep_topics, dim3(m->dm.MK, sc.nrep), dim3(256), 0, ctx->stream, a);
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

// status words of the scope: singular Sigma in any replica -> error
int check_status(mmm_ctm* m, Scope sc)
{
    std::vector<int> h((size_t)sc.nrep, 0);
    MMM_HIP(m->ctx, hipMemcpyAsync(h.data(), m->status.p + sc.rep0, sizeof(int) * sc.nrep, hipMemcpyDeviceToHost, m->ctx->stream));
    MMM_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    { int rc = mmm_p2p_check(m->ctx); if (rc) return rc; }
    for (int i = 0; i < sc.nrep; ++i)
        if (h[i]) return mmm_fail(m->ctx, MMM_ERR_NUMERIC, "update_Σ!: Sigma is singular (inv failed) in replica %d", sc.rep0 + i);
    return MMM_OK;
}

// the theta buffer is about to be given to the selected replica: park an explicit theta of its current owner
int claim_theta(mmm_ctm* m)
{
    const int own = m->theta_rep;
    if (own < 0 || own == m->sel || m->theta_state[own] != 2) return MMM_OK;
    if (!m->theta_spill[own]) {
        m->theta_spill[own].reset(new DevBuf<double>());
        MMM_HIP(m->ctx, m->theta_spill[own]->alloc((size_t)m->theta_n));
    }
    if (m->theta_n) MMM_HIP(m->ctx, hipMemcpyAsync(m->theta_spill[own]->p, m->theta.p, sizeof(double) * m->theta_n, hipMemcpyDeviceToDevice, m->ctx->stream));
    m->theta_rep = -1;
    return MMM_OK;
}

// theta of the selected replica into the theta buffer (MMCTM.jl:183-198)
int materialise_theta(mmm_ctm* m)
{
    if (m->theta_rep == m->sel) return MMM_OK;
    int rc = claim_theta(m);
    if (rc) return rc;
    const int st = m->theta_state[m->sel];
    if (st == 2) {
        MMM_CHECK(m->ctx, m->theta_spill[m->sel], "theta of replica %d was never stored", m->sel);
        if (m->theta_n) MMM_HIP(m->ctx, hipMemcpyAsync(m->theta.p, m->theta_spill[m->sel]->p, sizeof(double) * m->theta_n, hipMemcpyDeviceToDevice, m->ctx->stream));
    } else if (st == 0) {
        for (int i = 0; i < m->dm.M; ++i) {
            const size_t n = (size_t)m->nnzm[i] * m->dm.K[i];
            if (n) hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->ctx->stream, m->theta.p + m->dm.toff[i], n, 1.0 / m->dm.K[i]);
        }
        MMM_LAUNCH_CHECK(m->ctx);
    } else {
        const bool prev = st == 1;
        rc = run_estep(m, one(m), F_THETA_COMPUTE | F_THETA_STORE, prev ? m->lambda_prev.p : m->lambda.p, nullptr, prev ? m->expEeff_prev.p : m->expEeff.p);
        if (rc) return rc;
    }
    m->theta_rep = m->sel;
    return MMM_OK;
}


// props (+ ll written to dst + r*dst_stride for every replica of the scope)
// gauss_mu / gauss_sigma: also run update_μ! / update_Σ! of the pass in an extra block of the same launch (see k_ctm_loglik)
int run_loglik(mmm_ctm* m, Scope sc, double* dst_dev, size_t dst_stride, bool compute_ll, int gauss_mu = 0, int gauss_sigma = 0)
{
    mmm_ctx* ctx = m->ctx;
    const int M = m->dm.M;
    const size_t r0 = sc.rep0;
    const int gauss = (gauss_mu || gauss_sigma) ? 1 : 0;
    const size_t lds = sizeof(double) * std::max((m->wide ? (size_t)0 : (size_t)m->dm.GT) + kWavesS * 64 + 1 + MMM_LOGTAB_N,      // table | props | log table
                                                 gauss ? 2 * (size_t)m->dm.MK * m->dm.MK : (size_t)0);
    if (m->big) {      // sum K > 64: the Gaussian M-step as its own launch (device-memory inversion), then the generic props / ll sweep
        if (gauss) { int rc = run_mstep(m, sc, gauss_mu, gauss_sigma, 0, 0); if (rc) return rc; }
        hipLaunchKernelGGL(k_ctm_loglik_big, dim3(m->grid_s, sc.nrep), dim3(kBlockS), sizeof(double) * kWavesS * 64, ctx->stream, m->dev(), m->lambda.p + r0 * m->sDMK(),
                           m->phieff.p + r0 * m->dm.GT, m->props.p + r0 * m->sDMK(), m->llpart.p + r0 * m->grid_s * M, compute_ll ? 1 : 0, sc.active);
    } else if (m->tdense && !mmm_off(m->tune, MMM_OFF_CTM_LL_ROWS)) {      // dense corpora: the sweep over rows of counts
        DenseRows dr{};
        int kmx = 8, vt = 0;
        for (int i = 0; i < M; ++i) { dr.rows[i] = m->trows[i].p; dr.SL[i] = m->tSL[i]; dr.tpoff[i] = vt; vt += 16 * m->tSL[i]; kmx = std::max(kmx, theta_dense_kmx(m->dm.K[i])); }
        dr.tpoff[M] = vt;
        const size_t ldsd = sizeof(double) * std::max((size_t)vt * kmx + (size_t)kWavesS * 4 * kmx + 2 + MMM_LOGTAB_N, gauss ? 2 * (size_t)m->dm.MK * m->dm.MK : (size_t)0);
        auto go = [&](auto kern) -> int {
            if (ldsd > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
            hipLaunchKernelGGL(kern, dim3(m->grid_s + gauss, sc.nrep), dim3(kBlockS), ldsd, ctx->stream, m->dev(), m->lambda.p + r0 * m->sDMK(),
                               m->phieff.p + r0 * m->dm.GT, m->props.p + r0 * m->sDMK(), m->llpart.p + r0 * m->grid_s * M, compute_ll ? 1 : 0, sc.active,
                               mstep_args(m, sc, gauss_mu, gauss_sigma, 0, 0), gauss, dr);
            return MMM_OK;
        };
        int rc = kmx == 8 ? go(k_ctm_loglik_dense<8>) : (kmx == 10 ? go(k_ctm_loglik_dense<10>) : go(k_ctm_loglik_dense<16>));
        if (rc) return rc;
    } else {
        auto kll = m->wide ? (m->L == 16 ? k_ctm_loglik<false, 16> : (m->L == 32 ? k_ctm_loglik<false, 32> : k_ctm_loglik<false, 64>))
                           : (m->L == 16 ? k_ctm_loglik<true, 16> : (m->L == 32 ? k_ctm_loglik<true, 32> : k_ctm_loglik<true, 64>));
        if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kll, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kll, dim3(m->grid_s + gauss, sc.nrep), dim3(kBlockS), lds, ctx->stream, m->dev(), m->lambda.p + r0 * m->sDMK(),
                           m->phieff.p + r0 * m->dm.GT, m->props.p + r0 * m->sDMK(), m->llpart.p + r0 * m->grid_s * M, compute_ll ? 1 : 0, sc.active,
                           mstep_args(m, sc, gauss_mu, gauss_sigma, 0, 0), gauss);
    }
    MMM_LAUNCH_CHECK(ctx);
    if (!compute_ll) return MMM_OK;
    // inside a fit (fit_scope sets stop_enable / sc.active): the stopping rule and the pass counter ride on the row's last kernel
    const StopArgs st{(m->stop_enable && sc.active) ? 1 : 0, m->stop_tol, sc.active ? m->active.p + r0 : nullptr, sc.active ? m->npass.p + r0 : nullptr};
    if (!mmm_comm_active(ctx)) {        // nothing to exchange: column sums and the division by N_m in one launch
        hipLaunchKernelGGL(k_ll_finish, dim3(1, sc.nrep), dim3(64 * M), 0, ctx->stream, m->llpart.p + r0 * m->grid_s * M, m->grid_s, M, m->Nm.p,
                           m->llnum.p + r0 * m->s_llnum, m->s_llnum, dst_dev, dst_stride, sc.active, st);
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    hipLaunchKernelGGL(k_sum_columns, dim3(M, sc.nrep), dim3(64), 0, ctx->stream, m->llpart.p + r0 * m->grid_s * M, m->grid_s, M, m->llnum.p + r0 * m->s_llnum,
                       m->s_llnum, sc.active);
    MMM_LAUNCH_CHECK(ctx);
    int rc = mmm_allreduce_sum(ctx, m->llnum.p + r0 * m->s_llnum, (size_t)sc.nrep * m->s_llnum);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ll_store, dim3(1, sc.nrep), dim3(64), 0, ctx->stream, M, m->llnum.p + r0 * m->s_llnum, m->s_llnum, m->Nm.p, dst_dev, dst_stride, sc.active, st);
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

// ll history: [R][cap_hist][M]
int ensure_hist(mmm_ctm* m, int extra)
{
    const int M = m->dm.M;
    int need = 0;
    for (int r = 0; r < m->R; ++r) need = std::max(need, m->n_hist[r] + extra);
    if (need <= m->cap_hist) return MMM_OK;
    const int cap = std::max(2 * m->cap_hist, need + 64);
    DevBuf<double> nb;
    MMM_HIP(m->ctx, nb.alloc((size_t)m->R * cap * M));
    for (int r = 0; r < m->R; ++r)
        if (m->n_hist[r]) MMM_HIP(m->ctx, hipMemcpyAsync(nb.p + (size_t)r * cap * M, m->ll_hist.p + (size_t)r * m->cap_hist * M, sizeof(double) * m->n_hist[r] * M,
                                                         hipMemcpyDeviceToDevice, m->ctx->stream));
    MMM_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    m->ll_hist.swap(nb);
    m->cap_hist = cap;
    return MMM_OK;
}

// two per-replica copies in one launch (blocks [0, nb1) take the first array)
__global__ void k_copy2_rep(double* dst1, const double* src1, size_t n1, int nb1, double* dst2, const double* src2, size_t n2, const int* active)
{
    if (active && !active[blockIdx.y]) return;
    const bool first = (int)blockIdx.x < nb1;
    const size_t n = first ? n1 : n2;
    const size_t i = (size_t)(first ? blockIdx.x : blockIdx.x - nb1) * blockDim.x + threadIdx.x;
    if (i < n) (first ? dst1 : dst2)[blockIdx.y * n + i] = (first ? src1 : src2)[blockIdx.y * n + i];
}

int copy2_rep(mmm_ctm* m, Scope sc, double* dst1, const double* src1, size_t n1, double* dst2, const double* src2, size_t n2)
{
    const int nb1 = (int)((n1 + 255) / 256), nb2 = (int)((n2 + 255) / 256);
    if (nb1 + nb2 == 0) return MMM_OK;
    hipLaunchKernelGGL(k_copy2_rep, dim3(nb1 + nb2, sc.nrep), dim3(256), 0, m->ctx->stream, dst1 + sc.rep0 * n1, src1 + sc.rep0 * n1, n1, nb1,
                       dst2 + sc.rep0 * n2, src2 + sc.rep0 * n2, n2, sc.active);
    MMM_LAUNCH_CHECK(m->ctx);
    return MMM_OK;
}

int copy_rep(mmm_ctm* m, Scope sc, double* dst, const double* src, size_t n)
{
    if (!n) return MMM_OK;
    hipLaunchKernelGGL(k_copy_rep, dim3((unsigned)((n + 255) / 256), sc.nrep), dim3(256), 0, m->ctx->stream, dst + sc.rep0 * n, src + sc.rep0 * n, n, sc.active);
    MMM_LAUNCH_CHECK(m->ctx);
    return MMM_OK;
}

// one pass of the body of fit! (MMCTM.jl:462-479 / IMMCTM.jl:440-451) for every replica of the scope.  All replicas of the
// scope must have the same history length (they do: a batched fit advances its active replicas in lock step).
int run_update_alpha(mmm_ctm* m, Scope sc)
{
    hipLaunchKernelGGL(k_ctm_update_alpha, dim3(m->nalpha, sc.nrep), dim3(64), 0, m->ctx->stream, m->dm, m->tp, m->Elnphi.p + (size_t)sc.rep0 * m->GM,
                       m->alpha.p + (size_t)sc.rep0 * m->nalpha, m->GM, m->nalpha, m->opt.xtol_rule, m->opt.max_eval, sc.active);
    MMM_LAUNCH_CHECK(m->ctx);
    return MMM_OK;
}

// fit_flags: MMM_FIT_UPDATE_SIGMA | MMM_FIT_AUTO_ALPHA (updateΣ / autoα of fit!, MMCTM.jl:457-458)
int fused_pass(mmm_ctm* m, Scope sc, int fit_flags)
{
    const int update_sigma = fit_flags & MMM_FIT_UPDATE_SIGMA;
    mmm_ctx* ctx = m->ctx;
    const CtmDims& dm = m->dm;
    int rc;
    // keep lambda_{t-1} and the exp table of this pass (theta_t is rebuilt from them on demand): the theta phase writes them beside its
    // own reads; the wide-table path has no such hook and copies
    if (m->wide && (rc = copy2_rep(m, sc, m->lambda_prev.p, m->lambda.p, m->sDMK(), m->expEeff_prev.p, m->expEeff.p, (size_t)dm.GT))) return rc;
    // for d in 1:D fitdoc!(model, d)   (lambda is updated in place: the theta phase has consumed it before the solve phase)
    // One GPU: what only depends on the theta phase -- the reduction of the gamma statistics and the topic M-step -- runs on a side stream
    // BESIDE the solve phase (which leaves a wave slot per SIMD free) and is joined before the log-likelihood launch: two launches and
    // their boundaries off the pass's critical path.  Same kernels, same sums; mmm_tuning_opts.side_stream = -1 / 1: never / whenever possible.
    // Measured (5 regions x 10 passes each): cfg 5 (IMMCTM, topic M-step 13 us) 0.467 -> 0.458 ms per pass, min 0.407 -> 0.399; cfg 4 (MMCTM, topic
    // M-step 5 us) 1.082 -> 1.091: there the fork / join cost what the two short launches take -- so by default only the IMMCTM forks.
    const bool overlap = (m->tune.side_stream == 0 ? m->immctm : m->tune.side_stream > 0) && ctx->nranks == 1 && !m->wide && !m->big &&
                         !(fit_flags & MMM_FIT_AUTO_ALPHA) && !ctx->profiling && side_stream(ctx) != nullptr;
    rc = run_estep(m, sc, F_ZETA | F_THETA_COMPUTE | F_NU | F_LAMBDA | F_SLAB, m->lambda.p, m->lambda.p, m->expEeff.p,
                   m->wide ? nullptr : m->lambda_prev.p, m->wide ? nullptr : m->expEeff_prev.p, overlap);
    if (rc) return rc;
    const int nb1 = (m->nmom + 15) / 16, nb2 = (dm.GT + 15) / 16;
    // the side stream is joined on EVERY way out of this function once it has been forked: an early return (a failed launch, exchange or
    // copy further down) must not leave the reduction / topic M-step in flight while the caller reads state or retries the pass
    struct SideJoin {
        mmm_ctx* c; bool forked = false, joined = false;
        ~SideJoin() { if (forked && !joined) (void)hipStreamSynchronize(c->side); }
    } side_join{ctx};
    if (overlap) {
        side_join.forked = true;
        StreamSwap sw(ctx, ctx->side);
        MMM_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        hipLaunchKernelGGL(k_reduce_partials, dim3(nb2, sc.nrep), dim3(16, 64), 0, ctx->stream, m->mompart.p + sc.rep0 * m->grid_m * m->nmom, m->grid_m,
                           m->nmom, m->stats.p + sc.rep0 * m->s_stats, m->s_stats, sc.active, 0, m->partial.p + sc.rep0 * m->grid_e * dm.GT, m->grid_e, dm.GT,
                           m->stats.p + sc.rep0 * m->s_stats + m->nmom);
        MMM_LAUNCH_CHECK(ctx);
        if ((rc = run_mstep(m, sc, 0, 0, 1, 1))) return rc;
        MMM_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
    }
    // sufficient statistics: [sum lambda | sum nu | sum lambda lambda' | gamma sums]
    const size_t r0 = sc.rep0;
    ProfSpan* mid_span = new ProfSpan(ctx, 2);      // mmm_ctx_profile_select(2): moments, reduction, all-reduce, topic M-step
    struct SpanGuard { ProfSpan*& p; ~SpanGuard() { delete p; } } mid_guard{mid_span};
    if (sizeof(double) * 64 * dm.MK > 48 * 1024) MMM_HIP(m->ctx, hipFuncSetAttribute((const void*)k_ctm_moments, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 64 * dm.MK)));
    hipLaunchKernelGGL(k_ctm_moments, dim3(m->grid_m, sc.nrep), dim3(256), sizeof(double) * 64 * dm.MK, ctx->stream, dm.D, dm.MK, m->lambda.p + r0 * m->sDMK(),
                       m->nu.p + r0 * m->sDMK(), m->mompart.p + r0 * m->grid_m * m->nmom, sc.active);
    MMM_LAUNCH_CHECK(ctx);
    if (m->wide) {   // gamma sums by the term-major sweep, moments reduced on their own
        int kmax = 0;
        for (int i = 0; i < dm.M; ++i) kmax = std::max(kmax, dm.K[i]);
        auto ks = kmax > 32 ? k_ctm_stats_terms<64> : (kmax > 16 ? k_ctm_stats_terms<32> : k_ctm_stats_terms<16>);
        hipLaunchKernelGGL(ks, dim3(m->nterms, sc.nrep), dim3(m->stats_waves * MMM_WAVE), 0, ctx->stream, dm, m->term_ptr.p, m->tpost.p,
                           m->aexp.p + r0 * dm.D * dm.MK, m->expEeff.p + r0 * dm.GT, m->stats.p + r0 * m->s_stats + m->nmom, m->s_stats, sc.active);
        MMM_LAUNCH_CHECK(ctx);
        if ((rc = reduce_partials(m, sc, m->mompart.p, m->grid_m, m->nmom, m->stats.p, m->s_stats))) return rc;
    } else {   // moments and gamma sums reduced by one launch (with the side stream: the moments alone -- blocks beyond nb1 do the gamma part)
        hipLaunchKernelGGL(k_reduce_partials, dim3(overlap ? nb1 : nb1 + nb2, sc.nrep), dim3(16, 64), 0, ctx->stream, m->mompart.p + r0 * m->grid_m * m->nmom, m->grid_m,
                           m->nmom, m->stats.p + r0 * m->s_stats, m->s_stats, sc.active, nb1, m->partial.p + r0 * m->grid_e * dm.GT, m->grid_e, dm.GT,
                           m->stats.p + r0 * m->s_stats + m->nmom);
        MMM_LAUNCH_CHECK(ctx);
    }
    if ((rc = mmm_allreduce_sum(ctx, m->stats.p + r0 * m->s_stats, (size_t)sc.nrep * m->s_stats))) return rc;
    // update_μ!, update_Σ!, update_γ! (+Elnϕ), update_ϕ!
    // (the Gaussian part runs as an extra block of the log-likelihood launch below, beside the document sweep)
    const bool fuse = !mmm_off(m->tune, MMM_OFF_CTM_FUSED_GAUSS);
    const int do_sig = (update_sigma || m->immctm) ? 1 : 0;
    if ((rc = run_mstep(m, sc, fuse ? 0 : 1, fuse ? 0 : do_sig, overlap ? 0 : 1, 1))) return rc;
    if (overlap) { MMM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); side_join.joined = true; }
    if ((fit_flags & MMM_FIT_AUTO_ALPHA) && (rc = run_update_alpha(m, sc))) return rc;      // MMCTM.jl:472-474
    delete mid_span; mid_span = nullptr;
    // update_props! and the log-likelihoods
    if ((rc = ensure_hist(m, 1))) return rc;
    ProfSpan ll_span(ctx, 3);                       // mmm_ctx_profile_select(3): Gaussian M-step block + props + log-likelihood launches
    const int M = dm.M;
    int nh = 0;       // the replicas still running share one history length (fit_scope checks it)
    for (int i = 0; i < sc.nrep; ++i)
        if (!sc.active || m->h_active[sc.rep0 + i]) nh = std::max(nh, m->n_hist[sc.rep0 + i]);
    if ((rc = run_loglik(m, sc, m->ll_hist.p + ((size_t)sc.rep0 * m->cap_hist + nh) * M, (size_t)m->cap_hist * M, true, fuse ? 1 : 0,
                         fuse ? do_sig : 0))) return rc;
    for (int i = 0; i < sc.nrep; ++i) {
        const int r = sc.rep0 + i;
        if (sc.active && !m->h_active[r]) continue;
        m->n_hist[r]++; m->theta_state[r] = 1;
        if (m->theta_rep == r) m->theta_rep = -1;
    }
    return MMM_OK;
}

// one frozen-topic pass for the selected replica: the document loop of transform (MMCTM.jl:521-528; unsmoothed theta reads phi)
// or of fit_heldout / predict_modality_η (MMCTM.jl:565-569: fitdoc!), optional update_μ!/update_Σ! (fit_gaussian, :530-533),
// update_props! and the log-likelihoods.  Topics (gamma, Elnphi, phi) are not touched.
int frozen_pass(mmm_ctm* m, Scope sc, int flags)
{
    mmm_ctx* ctx = m->ctx;
    const CtmDims& dm = m->dm;
    int rc;
    const double* table = (flags & MMM_INFER_UNSMOOTHED) ? m->phieff.p : m->expEeff.p;
    if ((rc = copy_rep(m, sc, m->lambda_prev.p, m->lambda.p, m->sDMK()))) return rc;
    if ((rc = copy_rep(m, sc, m->expEeff_prev.p, table, (size_t)dm.GT))) return rc;
    rc = run_estep(m, sc, F_ZETA | F_THETA_COMPUTE | F_NU | F_LAMBDA, m->lambda.p, m->lambda.p, table);
    if (rc) return rc;
    if (flags & MMM_INFER_FIT_GAUSSIAN) {
        const size_t r0 = sc.rep0;
        if (sizeof(double) * 64 * dm.MK > 48 * 1024) MMM_HIP(m->ctx, hipFuncSetAttribute((const void*)k_ctm_moments, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 64 * dm.MK)));
        hipLaunchKernelGGL(k_ctm_moments, dim3(m->grid_m, sc.nrep), dim3(256), sizeof(double) * 64 * dm.MK, ctx->stream, dm.D, dm.MK, m->lambda.p + r0 * m->sDMK(),
                           m->nu.p + r0 * m->sDMK(), m->mompart.p + r0 * m->grid_m * m->nmom, sc.active);
        MMM_LAUNCH_CHECK(ctx);
        if ((rc = reduce_partials(m, sc, m->mompart.p, m->grid_m, m->nmom, m->stats.p, m->s_stats))) return rc;
        if ((rc = mmm_allreduce_sum(ctx, m->stats.p + r0 * m->s_stats, (size_t)sc.nrep * m->s_stats))) return rc;
        if ((rc = run_mstep(m, sc, 1, 1, 0, 0))) return rc;
    }
    if ((rc = ensure_hist(m, 1))) return rc;
    int nh = 0;
    for (int i = 0; i < sc.nrep; ++i)
        if (!sc.active || m->h_active[sc.rep0 + i]) nh = std::max(nh, m->n_hist[sc.rep0 + i]);
    if ((rc = run_loglik(m, sc, m->ll_hist.p + ((size_t)sc.rep0 * m->cap_hist + nh) * dm.M, (size_t)m->cap_hist * dm.M, true))) return rc;
    for (int i = 0; i < sc.nrep; ++i) {
        const int r = sc.rep0 + i;
        if (sc.active && !m->h_active[r]) continue;
        m->n_hist[r]++; m->theta_state[r] = 1;
        if (m->theta_rep == r) m->theta_rep = -1;
    }
    return MMM_OK;
}

int prep(mmm_ctm* m)
{
    if (int rc = mmm_ctx_usable(m->ctx, "CTM call")) return rc;
    MMM_HIP(m->ctx, hipSetDevice(m->ctx->device));
    return MMM_OK;
}

int upload_active(mmm_ctm* m)
{
    MMM_HIP(m->ctx, hipMemcpyAsync(m->active.p, m->h_active.data(), sizeof(int) * m->R, hipMemcpyHostToDevice, m->ctx->stream));
    return MMM_OK;
}

int create_impl(mmm_ctx* ctx, int R, int D, int M, const int* K, const int* V, const double* alpha, const int64_t* doc_ptr, const int32_t* term,
                const int32_t* count, const int* n_feat, const int* J, const int32_t* features, const double* gamma0,
                const mmm_solver_opts* opts, mmm_ctm** out)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, out && K && V && alpha && doc_ptr && gamma0, "mmm_ctm_create: NULL argument");
    MMM_CHECK(ctx, D >= 0 && M >= 1 && M <= kMaxM && R >= 1, "mmm_ctm_create: bad sizes D=%d M=%d (M <= %d) R=%d", D, M, kMaxM, R);
    *out = nullptr;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    mmm_ctm* m = new mmm_ctm();
    m->ctx = ctx; m->R = R;
    CtmDims& dm = m->dm;
    dm.D = D; dm.M = M; dm.koff[0] = 0; dm.goff[0] = 0;
    int64_t toff = 0;
    for (int i = 0; i < M; ++i) {
        if (K[i] < 1 || K[i] > kKmax || V[i] < 1) { int rc = mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_ctm_create: K[%d]=%d must be in 1..%d and V[%d]=%d >= 1", i, K[i], kKmax, i, V[i]); delete m; return rc; }
        dm.K[i] = K[i]; dm.V[i] = V[i];
        dm.koff[i + 1] = dm.koff[i] + K[i]; dm.goff[i + 1] = dm.goff[i] + K[i] * V[i];
        dm.estart[i] = doc_ptr[(size_t)i * (D + 1)];
        dm.toff[i] = toff;
        m->nnzm[i] = doc_ptr[(size_t)i * (D + 1) + D] - dm.estart[i];
        toff += m->nnzm[i] * K[i];
    }
    dm.MK = dm.koff[M]; dm.GT = dm.goff[M];
    if (dm.MK > 64 * kBigSlots) { int rc = mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_ctm_create: sum(K)=%d must be <= %d", dm.MK, 64 * kBigSlots); delete m; return rc; }
    m->L = dm.MK <= 16 ? 16 : (dm.MK <= 32 ? 32 : 64);
    m->big = dm.MK > 64;      // more coordinates than lanes: the generic kernels of ctm_big.cuh (and the wide-table data flow)
    const int64_t nnz = doc_ptr[(size_t)(M - 1) * (D + 1) + D];
    m->nnz = nnz; m->theta_n = toff;
    // validate + pack the corpus
    std::vector<int2> tc((size_t)nnz);
    std::vector<double> Ndm((size_t)D * M, 0.0);
    m->hNm.assign(M, 0.0);
    if (doc_ptr[0] != 0) { int rc = mmm_fail(ctx, MMM_ERR_ARG, "mmm_ctm_create: doc_ptr[0] != 0"); delete m; return rc; }
    for (int i = 0; i < M; ++i) {
        const int64_t* dp = doc_ptr + (size_t)i * (D + 1);
        if (i > 0 && dp[0] != doc_ptr[(size_t)(i - 1) * (D + 1) + D]) { int rc = mmm_fail(ctx, MMM_ERR_ARG, "mmm_ctm_create: doc_ptr of modality %d does not continue modality %d", i, i - 1); delete m; return rc; }
        for (int d = 0; d < D; ++d) {
            if (dp[d + 1] < dp[d]) { int rc = mmm_fail(ctx, MMM_ERR_ARG, "mmm_ctm_create: doc_ptr not monotone (m=%d d=%d)", i, d); delete m; return rc; }
            double n = 0.0;
            for (int64_t e = dp[d]; e < dp[d + 1]; ++e) {
                if (term[e] < 0 || term[e] >= V[i] || count[e] < 0) { int rc = mmm_fail(ctx, MMM_ERR_ARG, "mmm_ctm_create: entry %lld out of range", (long long)e); delete m; return rc; }
                tc[(size_t)e] = make_int2(term[e], count[e]);
                n += count[e];
            }
            Ndm[(size_t)d * M + i] = n;
            if (n > 0) m->hNm[i] += n;        // MMCTM.jl:409-414: only documents with N > 0 enter the ll
        }
    }
    // topics descriptor
    CtmTopics& tp = m->tp;
    m->immctm = (n_feat != nullptr);
    tp.immctm = m->immctm ? 1 : 0;
    int nalpha = M, GM = dm.GT;
    std::vector<int> featv;
    if (m->immctm) {
        if (!J || !features) { int rc = mmm_fail(ctx, MMM_ERR_ARG, "mmm_ctm_create: IMMCTM needs J and features"); delete m; return rc; }
        tp.aoff[0] = 0; tp.mgoff[0] = 0;
        long long fo = 0;
        for (int i = 0; i < M; ++i) {
            tp.nfeat[i] = n_feat[i];
            tp.aoff[i + 1] = tp.aoff[i] + n_feat[i];
            if (tp.aoff[i + 1] > 4 * kMaxM) { int rc = mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_ctm_create: more than %d features in total", 4 * kMaxM); delete m; return rc; }
            int sj = 0;
            for (int q = 0; q < n_feat[i]; ++q) { tp.J[tp.aoff[i] + q] = J[tp.aoff[i] + q]; sj += J[tp.aoff[i] + q]; }
            tp.SJ[i] = sj;
            tp.mgoff[i + 1] = tp.mgoff[i] + K[i] * sj;
            tp.foff[i] = fo;
            for (int q = 0; q < n_feat[i]; ++q) for (int v = 0; v < V[i]; ++v) {
                const int f = features[fo + (long long)q * V[i] + v];
                if (f < 0 || f >= J[tp.aoff[i] + q]) { int rc = mmm_fail(ctx, MMM_ERR_ARG, "mmm_ctm_create: feature value out of range"); delete m; return rc; }
            }
            fo += (long long)n_feat[i] * V[i];
        }
        featv.assign(features, features + fo);
        nalpha = tp.aoff[M]; GM = tp.mgoff[M];
    } else {
        for (int i = 0; i <= M; ++i) tp.mgoff[i] = dm.goff[i];
    }
    m->GM = GM;
    m->tune = ctx->tune;
    mmm_solver_opts so; mmm_solver_opts_default(&so);
    if (opts) so = *opts;
    m->opt = SolveOpts{so.xtol_rel, so.xtol_abs, so.nu_lower, so.xtol_rule, so.max_eval > 0 ? so.max_eval : 2000};
    // launch geometry
    const int G = MMM_WAVE / m->L;
    m->waves_e = 8;
    while (m->waves_e > 1 && estep_lds(m, F_SLAB) > 150 * 1024) m->waves_e >>= 1;
    // table + one slab beyond LDS: the wide path (theta phase through L2, gamma statistics by k_ctm_stats_terms).  ctm_build =
    // MMM_BUILD_WIDE forces it for any shape (tests, A/B)
    int kmax_all = 0;
    for (int i = 0; i < dm.M; ++i) kmax_all = std::max(kmax_all, dm.K[i]);
    if (estep_lds(m, F_SLAB) > 160 * 1024 || ctx->tune.ctm_build == MMM_BUILD_WIDE || kmax_all > 32 || m->big) { m->wide = true; m->waves_e = 8; }
    const int dpb = m->waves_e * G;
    const int per_cu = m->wide ? 2 : std::max(1, (int)((160 * 1024) / estep_lds(m, F_SLAB)));
    const int ncu = mmm_geo_cus(ctx);      // the CU count the geometry (and so the association of the cross-document sums) is derived from
    m->grid_e = std::max(1, std::min((D + dpb - 1) / dpb, ncu * std::min(per_cu, 2)));
    // Dense corpora: the fused pass's theta phase over rows of 16-bit counts (k_ctm_theta_dense), when every modality has at most 16 topics
    // and 128 terms, no document lists a term twice, every count fits 16 bits and at least half of the D x V_m entries are present.
    // ctm_build = MMM_BUILD_DENSE / _SPARSE forces / forbids it (tests, A/B); by default corpora of at least 32 documents per CU take it (below, a block of
    // the 16-lane layout has less than one wave step and the slab kernel is as fast).
    {
        const int dmode = ctx->tune.ctm_build == MMM_BUILD_DENSE ? 1 : (ctx->tune.ctm_build == MMM_BUILD_AUTO ? -1 : 0);
        bool ok = !m->wide && !m->big && dmode != 0 && D > 0 && (int64_t)D * dm.MK * 8 < ((int64_t)1 << 32) && D < (1 << 24);      // (32-bit byte offsets into lambda and into the rows)
        int64_t present = 0, cells = 0;
        for (int i = 0; i < M && ok; ++i) {
            const int sl = dm.V[i] <= 32 ? 2 : (dm.V[i] <= 48 ? 3 : (dm.V[i] <= 96 ? 6 : (dm.V[i] <= 128 ? 8 : 0)));
            if (sl == 0 || dm.K[i] > 16 || theta_dense_kmx(dm.K[i]) * sl > 64) { ok = false; break; }      // (the statistics must stay in registers)
            m->tSL[i] = sl;
            present += m->nnzm[i]; cells += (int64_t)D * dm.V[i];
        }
        if (ok && 2 * present < cells) ok = false;
        if (ok && dmode < 0 && D < 32 * ncu) ok = false;
        std::vector<std::vector<unsigned short>> rows((size_t)M);
        for (int i = 0; i < M && ok; ++i) {
            const int sls = (m->tSL[i] + 1) & ~1, Vp = 16 * sls;     // lane-major: the slots of lane l (terms l, 16 + l, ...) are contiguous, an even number
            rows[i].assign((size_t)D * Vp + 8, 0);      // (+ 16 bytes: k_ctm_loglik_dense reads 16 bytes from a lane's first slot)
            std::vector<int> seen((size_t)dm.V[i], -1);
            const int64_t* dp = doc_ptr + (size_t)i * (D + 1);
            for (int d = 0; d < D && ok; ++d)
                for (int64_t e = dp[d]; e < dp[d + 1]; ++e) {
                    if (seen[(size_t)term[e]] == d || count[e] >= 65536) { ok = false; break; }
                    seen[(size_t)term[e]] = d;
                    rows[i][(size_t)d * Vp + (term[e] & 15) * sls + (term[e] >> 4)] = (unsigned short)count[e];
                }
        }
        if (ok) {
            m->waves_e = 8;
            for (int i = 0; i < M; ++i) if (theta_dense_lds(m, i, theta_dense_kmx(dm.K[i])) > 160 * 1024) ok = false;
        }
        if (ok) {
            m->tdense = true;
            m->grid_e = std::max(1, std::min((D + 31) / 32, ncu));
            for (int i = 0; i < M; ++i) {
                hipError_t e_ = m->trows[i].alloc(rows[i].size());
                if (e_ != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(trows): %s", hipGetErrorString(e_)); delete m; return rc; }
                if (hipMemcpy(m->trows[i].p, rows[i].data(), sizeof(unsigned short) * rows[i].size(), hipMemcpyHostToDevice) != hipSuccess) {
                    int rc = mmm_fail(ctx, MMM_ERR_HIP, "upload of the rows of counts failed"); delete m; return rc;
                }
            }
        }
    }
    if (ctx->tune.grid_blocks > 0) m->grid_e = ctx->tune.grid_blocks;
    m->grid_s = std::max(1, std::min((D + kWavesS - 1) / kWavesS, ncu * 4));
    m->waves_s = 4;
    // solve phase: packed document groups (sum K lanes per document) for the shapes with a build (MMM_OFF_CTM_PACKED: the 16-lane rows)
    m->Ls = m->L;
    {
        if (!mmm_off(ctx->tune, MMM_OFF_CTM_PACKED) && (dm.MK == 6 || dm.MK == 10 || dm.MK == 12)) m->Ls = dm.MK;
        // several coordinates per lane (k_ctm_solve_cpl): sum K = 10 -> 2 lanes x 5 coordinates (32 document slots per wave; BASELINE config 5:
        // solve phase 334 -> 263 us); sum K = 28 -> 16 lanes, 14 of them x 2 coordinates (round 3; 4 slots per wave, 134 VGPRs at 3 waves per
        // SIMD, no scratch: BASELINE config 4 solve phase 1,096 -> 974 us at pass 20, 794 -> 757 us at pass 60 -- and the sums of a document
        // are associated exactly as the 32-lane butterfly associates them, so not a bit changes).  MMM_OFF_CTM_CPL switches the path off
        // (one coordinate per lane, lock step: k_ctm_estep<L, 1>).
        if (!mmm_off(ctx->tune, MMM_OFF_CTM_CPL)) {
            if (dm.MK == 10) { m->Ls = 2; m->cpl = 5; }
            else if (dm.MK == 28) { m->Ls = 16; m->cpl = 2; m->lam_occ = 3; }      // 14 of 16 lanes x 2 coordinates, 3 waves per SIMD
            if (m->cpl > 1) m->persist = true;
        }
    }
    if (m->big) { m->Ls = 64; m->cpl = kBigSlots; m->persist = false; }      // ctm_big.cuh: lane l holds coordinates l + 64 q
    const int Gs = MMM_WAVE / m->Ls;
    m->grid_v = std::max(1, std::min((D + m->waves_s * Gs - 1) / (m->waves_s * Gs), ncu * 8));
    // k_ctm_solve_cpl: as many waves as are resident at once (2 per SIMD), each with a contiguous range of documents that its
    // slots work through (a finished slot takes the range's next document)
    if (m->persist) m->grid_v = std::max(1, std::min((D + m->waves_s * Gs - 1) / (m->waves_s * Gs), ncu * (m->Ls == 16 ? m->lam_occ : 2)));
    // moment sums: whole 32-document tiles per block (a short last tile is padded to 32 and costs as much as a full one), at most 1024 blocks
    {
        const int tiles_per_block = std::max(1, (D + 32 * 1024 - 1) / (32 * 1024));
        m->grid_m = std::max(1, (D + 32 * tiles_per_block - 1) / (32 * tiles_per_block));
    }
    if (ctx->tune.moment_blocks > 0) m->grid_m = ctx->tune.moment_blocks;
    const size_t MK = dm.MK, DMK = (size_t)D * MK, Rz = (size_t)R;
    m->nmom = 2 * dm.MK + dm.MK * dm.MK; m->nalpha = nalpha;
    m->s_stats = (size_t)m->nmom + dm.GT + 16;
    m->s_llnum = (size_t)M + 8;
    m->h_active.assign(R, 1); m->n_hist.assign(R, 0); m->theta_state.assign(R, 0); m->theta_spill.resize(R);
#define A(buf, n) do { hipError_t e_ = m->buf.alloc(n); if (e_ != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(" #buf "): %s", hipGetErrorString(e_)); delete m; return rc; } } while (0)
    A(doc_ptr, (size_t)M * (D + 1)); A(tc, (size_t)nnz); A(Ndm, (size_t)D * M); A(features, featv.size()); A(alpha, Rz * nalpha);
    A(lambda, Rz * DMK); A(lambda_prev, Rz * DMK); A(nu, Rz * DMK); A(sumth, Rz * DMK); A(zeta, Rz * D * M); A(props, Rz * DMK); A(theta, (size_t)toff);
    A(mu, Rz * MK); A(Sigma, Rz * MK * MK); A(invSigma, Rz * MK * MK); A(gamma, Rz * GM); A(Elnphi, Rz * GM); A(phi, Rz * GM);
    A(Eeff, Rz * dm.GT); A(expEeff, Rz * dm.GT); A(expEeff_prev, Rz * dm.GT); A(phieff, Rz * dm.GT);
    A(partial, m->wide ? 1 : Rz * m->grid_e * dm.GT); A(mompart, Rz * m->grid_m * m->nmom); A(stats, Rz * m->s_stats);
    A(llpart, Rz * m->grid_s * M); A(llnum, Rz * m->s_llnum); A(Nm, (size_t)M); A(elbopart, (size_t)m->grid_s * 5 + 16 + 2 * MK);
    A(nev_nu, Rz * D); A(nev_lam, Rz * D); A(status, Rz); A(active, Rz); A(npass, Rz);
    A(big_scratch, m->big ? Rz * 2 * (size_t)dm.MK * dm.MK : 1);
#undef A
    hipStream_t st = ctx->stream;
    MMM_HIP(ctx, hipMemcpyAsync(m->doc_ptr.p, doc_ptr, sizeof(int64_t) * M * (D + 1), hipMemcpyHostToDevice, st));
    if (nnz) MMM_HIP(ctx, hipMemcpyAsync(m->tc.p, tc.data(), sizeof(int2) * nnz, hipMemcpyHostToDevice, st));
    if (D) MMM_HIP(ctx, hipMemcpyAsync(m->Ndm.p, Ndm.data(), sizeof(double) * D * M, hipMemcpyHostToDevice, st));
    if (!featv.empty()) MMM_HIP(ctx, hipMemcpyAsync(m->features.p, featv.data(), sizeof(int) * featv.size(), hipMemcpyHostToDevice, st));
    for (size_t r = 0; r < Rz; ++r) MMM_HIP(ctx, hipMemcpyAsync(m->alpha.p + r * nalpha, alpha, sizeof(double) * nalpha, hipMemcpyHostToDevice, st));
    MMM_HIP(ctx, hipMemcpyAsync(m->gamma.p, gamma0, sizeof(double) * Rz * GM, hipMemcpyHostToDevice, st));
    std::vector<int64_t> tptr;
    std::vector<int2> tpost;
    if (m->wide) {      // posting lists per (modality, term), documents ascending: the summation order of k_ctm_stats_terms
        int nterms = 0;
        std::vector<int> voff(M + 1, 0);
        for (int i = 0; i < M; ++i) { voff[i + 1] = voff[i] + V[i]; }
        nterms = voff[M];
        tptr.assign((size_t)nterms + 1, 0);
        for (int i = 0; i < M; ++i) {
            const int64_t* dp = doc_ptr + (size_t)i * (D + 1);
            for (int64_t e = dp[0]; e < dp[D]; ++e) tptr[(size_t)voff[i] + term[e] + 1]++;
        }
        for (int t = 0; t < nterms; ++t) tptr[(size_t)t + 1] += tptr[(size_t)t];
        tpost.resize((size_t)nnz);
        std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
        for (int i = 0; i < M; ++i) {
            const int64_t* dp = doc_ptr + (size_t)i * (D + 1);
            for (int d = 0; d < D; ++d)
                for (int64_t e = dp[d]; e < dp[d + 1]; ++e) tpost[(size_t)fill[(size_t)voff[i] + term[e]]++] = make_int2(d, count[e]);
        }
        m->nterms = nterms;
        hipError_t e1 = m->term_ptr.alloc((size_t)nterms + 1), e2 = m->tpost.alloc((size_t)nnz), e3 = m->aexp.alloc(Rz * DMK);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(postings): out of memory"); delete m; return rc; }
        MMM_HIP(ctx, hipMemcpyAsync(m->term_ptr.p, tptr.data(), sizeof(int64_t) * ((size_t)nterms + 1), hipMemcpyHostToDevice, st));
        if (nnz) MMM_HIP(ctx, hipMemcpyAsync(m->tpost.p, tpost.data(), sizeof(int2) * (size_t)nnz, hipMemcpyHostToDevice, st));
        const int64_t avg = nnz / std::max(1, nterms);
        m->stats_waves = 1;
        while (m->stats_waves < 8 && avg / (m->stats_waves * 2) >= 128) m->stats_waves *= 2;
    }
    MMM_HIP(ctx, hipMemsetAsync(m->status.p, 0, sizeof(int) * R, st));
    MMM_HIP(ctx, hipMemsetAsync(m->nev_nu.p, 0, sizeof(int) * std::max<size_t>(Rz * D, 1), st));
    MMM_HIP(ctx, hipMemsetAsync(m->nev_lam.p, 0, sizeof(int) * std::max<size_t>(Rz * D, 1), st));
    int rc = upload_active(m);
    if (rc) { delete m; return rc; }
    MMM_HIP(ctx, hipStreamSynchronize(st));
    tp.features = m->features.p; tp.alpha = m->alpha.p;
    // global N per modality and D (sum over ranks)
    {
        std::vector<double> h(m->hNm);
        h.push_back((double)D);
        MMM_HIP(ctx, hipMemcpyAsync(m->llnum.p, h.data(), sizeof(double) * (M + 1), hipMemcpyHostToDevice, st));
        if ((rc = mmm_allreduce_sum(ctx, m->llnum.p, M + 1))) { delete m; return rc; }
        MMM_HIP(ctx, hipMemcpyAsync(h.data(), m->llnum.p, sizeof(double) * (M + 1), hipMemcpyDeviceToHost, st));
        MMM_HIP(ctx, hipStreamSynchronize(st));
        m->Dglobal = h[M];
        MMM_HIP(ctx, hipMemcpyAsync(m->Nm.p, h.data(), sizeof(double) * M, hipMemcpyHostToDevice, st));
        MMM_HIP(ctx, hipStreamSynchronize(st));
    }
    // constructor state (MMCTM.jl:44-86): mu = 0, Sigma = invSigma = I, theta = 1/K, Elnphi from gamma0, phi = gamma0
    // (deepcopy, MMCTM.jl:80), lambda = 0, nu = 1, zeta = update_ζ!
    std::vector<double> eye(Rz * MK * MK, 0.0);
    for (size_t r = 0; r < Rz; ++r) for (size_t i = 0; i < MK; ++i) eye[r * MK * MK + i * MK + i] = 1.0;
    MMM_HIP(ctx, hipMemsetAsync(m->mu.p, 0, sizeof(double) * Rz * MK, st));
    MMM_HIP(ctx, hipMemcpyAsync(m->Sigma.p, eye.data(), sizeof(double) * eye.size(), hipMemcpyHostToDevice, st));
    MMM_HIP(ctx, hipMemcpyAsync(m->invSigma.p, eye.data(), sizeof(double) * eye.size(), hipMemcpyHostToDevice, st));
    MMM_HIP(ctx, hipStreamSynchronize(st));
    if (DMK) {
        MMM_HIP(ctx, hipMemsetAsync(m->lambda.p, 0, sizeof(double) * Rz * DMK, st));
        MMM_HIP(ctx, hipMemsetAsync(m->lambda_prev.p, 0, sizeof(double) * Rz * DMK, st));
        MMM_HIP(ctx, hipMemsetAsync(m->props.p, 0, sizeof(double) * Rz * DMK, st));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((Rz * DMK + 255) / 256)), dim3(256), 0, st, m->nu.p, Rz * DMK, 1.0);
    }
    MMM_LAUNCH_CHECK(ctx);
    if ((rc = materialise_theta(m))) { delete m; return rc; }
    Scope sc{0, R, nullptr};
    if ((rc = run_mstep(m, sc, 0, 0, 1, 0))) { delete m; return rc; }                       // update_Elnϕ! on gamma0 (+ tables)
    if (!m->immctm) MMM_HIP(ctx, hipMemcpyAsync(m->phi.p, m->gamma.p, sizeof(double) * Rz * GM, hipMemcpyDeviceToDevice, st));   // MMCTM.jl:80
    if ((rc = run_estep(m, sc, F_ZETA, m->lambda.p, nullptr, nullptr))) { delete m; return rc; }
    MMM_HIP(ctx, hipStreamSynchronize(st));
    *out = m;
    mmm_ctx_model_created(ctx);
    return MMM_OK;
}

} // namespace

extern "C" {

int mmm_ctm_create(mmm_ctx* ctx, int D, int M, const int* K, const int* V, const double* alpha, const int64_t* doc_ptr, const int32_t* term,
                   const int32_t* count, const int* n_feat, const int* J, const int32_t* features, const double* gamma0,
                   const mmm_solver_opts* opts, mmm_ctm** out)
{
    return create_impl(ctx, 1, D, M, K, V, alpha, doc_ptr, term, count, n_feat, J, features, gamma0, opts, out);
}

int mmm_ctm_create_batch(mmm_ctx* ctx, int R, int D, int M, const int* K, const int* V, const double* alpha, const int64_t* doc_ptr,
                         const int32_t* term, const int32_t* count, const int* n_feat, const int* J, const int32_t* features,
                         const double* gamma0, const mmm_solver_opts* opts, mmm_ctm** out)
{
    return create_impl(ctx, R, D, M, K, V, alpha, doc_ptr, term, count, n_feat, J, features, gamma0, opts, out);
}

int mmm_ctm_replicas(const mmm_ctm* m) { return m ? m->R : 0; }

int mmm_ctm_select(mmm_ctm* m, int r)
{
    if (!m) return MMM_ERR_ARG;
    MMM_CHECK(m->ctx, r >= 0 && r < m->R, "mmm_ctm_select: replica %d out of range (0..%d)", r, m->R - 1);
    m->sel = r;
    return MMM_OK;
}

int mmm_ctm_destroy(mmm_ctm* m)
{
    if (!m) return MMM_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->pin_flags) (void)hipHostFree(m->pin_flags);
    mmm_ctx* ctx = m->ctx;
    delete m;
    mmm_ctx_model_destroyed(ctx);
    return MMM_OK;
}

// pointer to the selected replica's copy of a field
static int ctm_field(mmm_ctm* m, int field, double** p, size_t* n)
{
    const size_t MK = m->dm.MK, D = m->dm.D, r = m->sel, GT = m->dm.GT, GM = m->GM;
    switch (field) {
        case MMM_CTM_MU: *p = m->mu.p + r * MK; *n = MK; break;
        case MMM_CTM_SIGMA: *p = m->Sigma.p + r * MK * MK; *n = MK * MK; break;
        case MMM_CTM_INVSIGMA: *p = m->invSigma.p + r * MK * MK; *n = MK * MK; break;
        case MMM_CTM_GAMMA: *p = m->gamma.p + r * GM; *n = GM; break;
        case MMM_CTM_ELNPHI: *p = m->Elnphi.p + r * GM; *n = GM; break;
        case MMM_CTM_PHI: *p = m->immctm ? m->phieff.p + r * GT : m->phi.p + r * GM; *n = m->immctm ? GT : GM; break;
        case MMM_CTM_LAMBDA: *p = m->lambda.p + r * D * MK; *n = D * MK; break;
        case MMM_CTM_NU: *p = m->nu.p + r * D * MK; *n = D * MK; break;
        case MMM_CTM_ZETA: *p = m->zeta.p + r * D * m->dm.M; *n = D * m->dm.M; break;
        case MMM_CTM_PROPS: *p = m->props.p + r * D * MK; *n = D * MK; break;
        case MMM_CTM_THETA: *p = m->theta.p; *n = (size_t)m->theta_n; break;
        case MMM_CTM_ALPHA: *p = m->alpha.p + r * m->nalpha; *n = (size_t)m->nalpha; break;
        default: return mmm_fail(m->ctx, MMM_ERR_ARG, "unknown CTM field %d", field);
    }
    return MMM_OK;
}

int mmm_ctm_get(mmm_ctm* m, int field, double* host, size_t n)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc) return rc;
    double* p; size_t cnt;
    if ((rc = ctm_field(m, field, &p, &cnt))) return rc;
    MMM_CHECK(ctx, host && n == cnt, "mmm_ctm_get(field %d): expected %zu doubles, got %zu", field, cnt, n);
    if (field == MMM_CTM_THETA && (rc = materialise_theta(m))) return rc;
    if (n) MMM_HIP(ctx, hipMemcpyAsync(host, p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

// stage calls that read theta: bring the selected replica's theta into the buffer first
static int begin_stage(mmm_ctm* m)
{
    int rc = prep(m);
    if (rc) return rc;
    return materialise_theta(m);
}

int mmm_ctm_set(mmm_ctm* m, int field, const double* host, size_t n)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc) return rc;
    double* p; size_t cnt;
    if ((rc = ctm_field(m, field, &p, &cnt))) return rc;
    MMM_CHECK(ctx, host && n == cnt, "mmm_ctm_set(field %d): expected %zu doubles, got %zu", field, cnt, n);
    if (field == MMM_CTM_THETA && (rc = claim_theta(m))) return rc;
    if (n) MMM_HIP(ctx, hipMemcpyAsync(p, host, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (field == MMM_CTM_THETA) { m->theta_rep = m->sel; m->theta_state[m->sel] = 2; }
    const size_t r = m->sel, GT = m->dm.GT;
    if (field == MMM_CTM_ELNPHI || (field == MMM_CTM_GAMMA && m->immctm)) {
        // the uploaded Elnphi is what update_θ! reads from now on (MMCTM.jl:190); IMMCTM's ll normalises the uploaded gamma
        const bool e = field == MMM_CTM_ELNPHI;
        hipLaunchKernelGGL(k_ctm_tables_from_Elnphi, dim3(m->dm.MK), dim3(256), 0, ctx->stream, m->dm, m->tp, e ? m->Elnphi.p + r * m->GM : nullptr,
                           m->Eeff.p + r * GT, m->expEeff.p + r * GT, e ? nullptr : m->gamma.p + r * m->GM, m->phieff.p + r * GT);
        MMM_LAUNCH_CHECK(ctx);
    }
    if (field == MMM_CTM_PHI && !m->immctm)   // ... and the uploaded phi what the ll and unsmoothed_update_θ! read
        MMM_HIP(ctx, hipMemcpyAsync(m->phieff.p + r * GT, m->phi.p + r * m->GM, sizeof(double) * GT, hipMemcpyDeviceToDevice, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_ctm_update_zeta(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prep(m);
    if (rc) return rc;
    return run_estep(m, one(m), F_ZETA, m->lambda.p, nullptr, nullptr);
}

int mmm_ctm_update_theta(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prep(m);
    if (rc) return rc;
    if ((rc = claim_theta(m))) return rc;
    if ((rc = run_estep(m, one(m), F_THETA_COMPUTE | F_THETA_STORE, m->lambda.p, nullptr, m->expEeff.p))) return rc;
    m->theta_rep = m->sel; m->theta_state[m->sel] = 2;
    return MMM_OK;
}

int mmm_ctm_update_nu(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = begin_stage(m);
    if (rc) return rc;
    return run_estep(m, one(m), F_NU, m->lambda.p, nullptr, nullptr);
}

int mmm_ctm_update_lambda(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = begin_stage(m);
    if (rc) return rc;
    // in place on the current lambda: the solver reads its start point before it stores the result
    return run_estep(m, one(m), F_THETA_STORED | F_LAMBDA, m->lambda.p, m->lambda.p, nullptr);
}

static int moments_to_stats(mmm_ctm* m)
{
    const CtmDims& dm = m->dm;
    const Scope sc = one(m);
    const size_t r0 = sc.rep0;
    if (sizeof(double) * 64 * dm.MK > 48 * 1024) MMM_HIP(m->ctx, hipFuncSetAttribute((const void*)k_ctm_moments, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 64 * dm.MK)));
    hipLaunchKernelGGL(k_ctm_moments, dim3(m->grid_m, 1), dim3(256), sizeof(double) * 64 * dm.MK, m->ctx->stream, dm.D, dm.MK, m->lambda.p + r0 * m->sDMK(),
                       m->nu.p + r0 * m->sDMK(), m->mompart.p + r0 * m->grid_m * m->nmom, (const int*)nullptr);
    MMM_LAUNCH_CHECK(m->ctx);
    int rc = reduce_partials(m, sc, m->mompart.p, m->grid_m, m->nmom, m->stats.p, m->s_stats);
    if (rc) return rc;
    return mmm_allreduce_sum(m->ctx, m->stats.p + r0 * m->s_stats, (size_t)m->nmom);
}

int mmm_ctm_update_mu(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prep(m);
    if (rc || (rc = moments_to_stats(m))) return rc;
    return run_mstep(m, one(m), 1, 0, 0, 0);
}

int mmm_ctm_update_Sigma(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prep(m);
    if (rc || (rc = moments_to_stats(m))) return rc;
    if ((rc = run_mstep(m, one(m), 0, 1, 0, 0))) return rc;      // uses the stored mu, as update_Σ! does (MMCTM.jl:207)
    return check_status(m, one(m));
}

int mmm_ctm_update_gamma(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = begin_stage(m);
    if (rc) return rc;
    const CtmDims& dm = m->dm;
    double* sums = m->stats.p + (size_t)m->sel * m->s_stats + m->nmom;
    MMM_HIP(ctx, hipMemsetAsync(sums, 0, sizeof(double) * dm.GT, ctx->stream));
    for (int i = 0; i < dm.M; ++i) {
        const int64_t n = m->nnzm[i];
        if (n > 0) hipLaunchKernelGGL(k_ctm_gamma_from_theta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, m->dev(), i, m->theta.p, sums);
    }
    MMM_LAUNCH_CHECK(ctx);
    if ((rc = mmm_allreduce_sum(ctx, sums, (size_t)dm.GT))) return rc;
    return run_mstep(m, one(m), 0, 0, 1, 1);
}

int mmm_ctm_update_Elnphi(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = begin_stage(m);
    if (rc) return rc;
    return run_mstep(m, one(m), 0, 0, 1, 0);
}

int mmm_ctm_update_alpha(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prep(m);
    if (rc) return rc;
    return run_update_alpha(m, one(m));
}

int mmm_ctm_update_props(mmm_ctm* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prep(m);
    if (rc) return rc;
    return run_loglik(m, one(m), nullptr, 0, false);
}

int mmm_ctm_update_phi(mmm_ctm* m)
{
    // phi = gamma / sum gamma is refreshed together with Elnphi by the M-step kernel
    return mmm_ctm_update_Elnphi(m);
}

int mmm_ctm_loglik(mmm_ctm* m, double* ll)
{
    if (!m || !ll) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc) return rc;
    double* dst = m->elbopart.p;            // scratch
    if ((rc = run_loglik(m, one(m), dst, 0, true))) return rc;
    MMM_HIP(ctx, hipMemcpyAsync(ll, dst, sizeof(double) * m->dm.M, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_ctm_objectives(mmm_ctm* m, int d, double* lambda_val, double* lambda_grad, double* nu_val, double* nu_grad)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc || (rc = materialise_theta(m))) return rc;
    MMM_CHECK(ctx, d >= 0 && d < m->dm.D, "mmm_ctm_objectives: document %d out of range", d);
    if (m->big) return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_ctm_objectives: not available for sum K > 64 (diagnostic entry point)");
    const size_t MK = m->dm.MK, r = m->sel;
    DevBuf<double> tmp;
    MMM_HIP(ctx, tmp.alloc(2 + 2 * MK));
    hipLaunchKernelGGL(k_ctm_objectives, dim3(1), dim3(64), 0, ctx->stream, m->dev(), d, m->invSigma.p + r * MK * MK, m->mu.p + r * MK, m->lambda.p + r * m->sDMK(),
                       m->nu.p + r * m->sDMK(), m->zeta.p + r * m->dm.D * m->dm.M, m->theta.p, tmp.p);
    MMM_LAUNCH_CHECK(ctx);
    std::vector<double> h(2 + 2 * MK);
    MMM_HIP(ctx, hipMemcpyAsync(h.data(), tmp.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (lambda_val) *lambda_val = h[0];
    if (nu_val) *nu_val = h[1];
    if (lambda_grad) memcpy(lambda_grad, h.data() + 2, sizeof(double) * MK);
    if (nu_grad) memcpy(nu_grad, h.data() + 2 + MK, sizeof(double) * MK);
    return MMM_OK;
}

int mmm_ctm_doc_sums(mmm_ctm* m, int d, double* sumtheta, double* Ndivzeta)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = begin_stage(m);
    if (rc) return rc;
    MMM_CHECK(ctx, d >= 0 && d < m->dm.D, "mmm_ctm_doc_sums: document %d out of range", d);
    const size_t MK = m->dm.MK;
    DevBuf<double> tmp;
    MMM_HIP(ctx, tmp.alloc(2 * MK));
    hipLaunchKernelGGL(k_ctm_doc_sums, dim3(1), dim3(256), 0, ctx->stream, m->dev(), d, m->zeta.p + (size_t)m->sel * m->dm.D * m->dm.M, m->theta.p, tmp.p);
    MMM_LAUNCH_CHECK(ctx);
    std::vector<double> h(2 * MK);
    MMM_HIP(ctx, hipMemcpyAsync(h.data(), tmp.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (sumtheta) memcpy(sumtheta, h.data(), sizeof(double) * MK);
    if (Ndivzeta) memcpy(Ndivzeta, h.data() + MK, sizeof(double) * MK);
    return MMM_OK;
}

// update_ζ!(model, d), update_θ!(model, d), update_ν!(model, d), update_λ!(model, d): the stage kernels process every document of the shard
// in one launch, so the per-document form runs the stage and then puts every OTHER document's values back.
int mmm_ctm_update_doc(mmm_ctm* m, int stage, int d)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = begin_stage(m);
    if (rc) return rc;
    MMM_CHECK(ctx, d >= 0 && d < m->dm.D, "mmm_ctm_update_doc: document %d out of range", d);
    MMM_CHECK(ctx, stage >= MMM_STAGE_ZETA && stage <= MMM_STAGE_LAMBDA, "mmm_ctm_update_doc: unknown stage %d", stage);
    if (stage == MMM_STAGE_THETA && (rc = claim_theta(m))) return rc;
    const int field = stage == MMM_STAGE_ZETA ? MMM_CTM_ZETA : stage == MMM_STAGE_THETA ? MMM_CTM_THETA : stage == MMM_STAGE_NU ? MMM_CTM_NU : MMM_CTM_LAMBDA;
    double* p; size_t cnt;
    if ((rc = ctm_field(m, field, &p, &cnt))) return rc;
    DevBuf<double> save;
    MMM_HIP(ctx, save.alloc(cnt));
    if (cnt) MMM_HIP(ctx, hipMemcpyAsync(save.p, p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
    rc = stage == MMM_STAGE_ZETA ? mmm_ctm_update_zeta(m) : stage == MMM_STAGE_THETA ? mmm_ctm_update_theta(m) : stage == MMM_STAGE_NU ? mmm_ctm_update_nu(m)
                                                                                                                                       : mmm_ctm_update_lambda(m);
    if (rc) return rc;
    if ((rc = ctm_field(m, field, &p, &cnt))) return rc;
    if (stage == MMM_STAGE_THETA) {
        hipLaunchKernelGGL(k_ctm_copy_doc_theta, dim3(1), dim3(256), 0, ctx->stream, m->dev(), d, p, save.p);
        MMM_LAUNCH_CHECK(ctx);
    } else {
        const size_t w = stage == MMM_STAGE_ZETA ? m->dm.M : m->dm.MK;
        MMM_HIP(ctx, hipMemcpyAsync(save.p + (size_t)d * w, p + (size_t)d * w, sizeof(double) * w, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (cnt) MMM_HIP(ctx, hipMemcpyAsync(p, save.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_ctm_solver_stats(mmm_ctm* m, int64_t* n_eval_nu, int64_t* n_eval_lambda, int64_t* n_capped, int* per_doc_nu, int* per_doc_lambda)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc) return rc;
    const int D = m->dm.D;
    std::vector<int> a((size_t)D), b((size_t)D);
    if (D) {
        MMM_HIP(ctx, hipMemcpyAsync(a.data(), m->nev_nu.p + (size_t)m->sel * D, sizeof(int) * D, hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipMemcpyAsync(b.data(), m->nev_lam.p + (size_t)m->sel * D, sizeof(int) * D, hipMemcpyDeviceToHost, ctx->stream));
    }
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int64_t sa = 0, sb = 0, cap = 0;
    for (int d = 0; d < D; ++d) { if (a[d] < 0) { ++cap; sa -= a[d]; } else sa += a[d]; if (b[d] < 0) { ++cap; sb -= b[d]; } else sb += b[d]; }
    if (n_eval_nu) *n_eval_nu = sa;
    if (n_eval_lambda) *n_eval_lambda = sb;
    if (n_capped) *n_capped = cap;
    if (per_doc_nu && D) memcpy(per_doc_nu, a.data(), sizeof(int) * D);
    if (per_doc_lambda && D) memcpy(per_doc_lambda, b.data(), sizeof(int) * D);
    return MMM_OK;
}

int mmm_ctm_geometry(const mmm_ctm* m, int out[8])
{
    if (!m || !out) return MMM_ERR_ARG;
    out[0] = m->L; out[1] = m->grid_e; out[2] = m->waves_e; out[3] = m->grid_m; out[4] = m->wide ? 1 : (m->tdense ? 2 : 0); out[5] = m->Ls; out[6] = m->cpl; out[7] = 0;
    return MMM_OK;
}

namespace {
__global__ void k_debug_math(int op, size_t n, const double* a, const double* b, double* out)
{
    __shared__ __attribute__((aligned(16))) double sTabs[MMM_EXPTAB_N + MMM_LOGTAB_N];
    stage_solve_tabs(sTabs);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // n is a multiple of 64 for the collectives: no early return
    const double x = i < n ? a[i] : 0.0, y = (i < n && b) ? b[i] : 1.0;
    double r = 0.0;
    switch (op) {
    case 9: r = ar_exp_tab(x, sTabs); break;
    case 10: r = ar_log_tab(x, sTabs + MMM_EXPTAB_N); break;
    case 0: r = ar_exp(x); break;
    case 1: r = ar_log(x); break;
    case 2: r = dev_digamma_ar(x); break;
    case 3: r = dev_div(x, y); break;
    case 4: r = dev_sqrt(x); break;
    case 5: r = group_sum<16>(x); break;
    case 6: r = group_sum<32>(x); break;
    case 7: r = group_sum<64>(x); break;
    case 8: r = wave_sum(x); break;
    }
    if (i < n) out[i] = r;
}
}

int mmm_debug_math(mmm_ctx* ctx, int op, size_t n, const double* a, const double* b, double* out)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, a && out && op >= 0 && op <= 10, "mmm_debug_math: bad arguments");
    MMM_CHECK(ctx, op < 5 || op > 8 || n % 64 == 0, "mmm_debug_math: the collectives need n %% 64 == 0");
    if (!n) return MMM_OK;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<double> da, db, dout;
    MMM_HIP(ctx, da.alloc(n)); MMM_HIP(ctx, dout.alloc(n));
    MMM_HIP(ctx, hipMemcpyAsync(da.p, a, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (b) { MMM_HIP(ctx, db.alloc(n)); MMM_HIP(ctx, hipMemcpyAsync(db.p, b, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream)); }
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, n, da.p, b ? db.p : nullptr, dout.p);
    MMM_LAUNCH_CHECK(ctx);
    MMM_HIP(ctx, hipMemcpyAsync(out, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_ctm_iterate(mmm_ctm* m, int n_iter, int update_sigma)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prep(m);
    if (rc) return rc;
    MMM_CHECK(m->ctx, n_iter >= 0, "mmm_ctm_iterate: n_iter < 0");
    for (int i = 0; i < n_iter; ++i) if ((rc = fused_pass(m, one(m), update_sigma))) return rc;
    return MMM_OK;
}

int mmm_ctm_ll_history(mmm_ctm* m, double* ll, int max_n, int* n)
{
    if (!m || !n) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc) return rc;
    const int M = m->dm.M, nh = m->n_hist[m->sel], cnt = std::min(max_n, nh);
    if (cnt > 0 && ll) MMM_HIP(ctx, hipMemcpyAsync(ll, m->ll_hist.p + ((size_t)m->sel * m->cap_hist + (nh - cnt)) * M, sizeof(double) * cnt * M, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n = cnt;
    return MMM_OK;
}

int mmm_ctm_elbo(mmm_ctm* m, double* elbo, double terms[7])
{
    if (!m || !elbo) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc || (rc = materialise_theta(m))) return rc;
    const CtmDims& dm = m->dm;
    const size_t MKz = dm.MK, r = m->sel;
    const size_t lds = sizeof(double) * (MKz * MKz + (m->wide ? 0 : dm.GT) + kWavesS * 64);
    auto kel = m->wide ? k_ctm_elbo_docs<false> : k_ctm_elbo_docs<true>;
    if (lds > 48 * 1024 && !m->big) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double* acc = m->elbopart.p + (size_t)m->grid_s * 5;     // [0..4] doc sums, [5..7] topic side
    if (m->big)
        hipLaunchKernelGGL(k_ctm_elbo_docs_big, dim3(m->grid_s), dim3(kBlockS), sizeof(double) * kWavesS * MKz, ctx->stream, m->dev(), m->invSigma.p + r * MKz * MKz,
                           m->mu.p + r * MKz, m->lambda.p + r * m->sDMK(), m->nu.p + r * m->sDMK(), m->zeta.p + r * dm.D * dm.M, m->theta.p, m->Eeff.p + r * dm.GT,
                           m->elbopart.p);
    else
    hipLaunchKernelGGL(kel, dim3(m->grid_s), dim3(kBlockS), lds, ctx->stream, m->dev(), m->invSigma.p + r * MKz * MKz, m->mu.p + r * MKz,
                       m->lambda.p + r * m->sDMK(), m->nu.p + r * m->sDMK(), m->zeta.p + r * dm.D * dm.M, m->theta.p, m->Eeff.p + r * dm.GT, m->elbopart.p);
    hipLaunchKernelGGL(k_sum_columns, dim3(5, 1), dim3(64), 0, ctx->stream, m->elbopart.p, m->grid_s, 5, acc, (size_t)0, (const int*)nullptr);
    const size_t lds2 = m->big ? 0 : sizeof(double) * 2 * MKz * MKz;
    if (lds2 > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)k_ctm_elbo_topics, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    CtmTopics tpr = m->tp;
    tpr.alpha += r * m->nalpha;
    hipLaunchKernelGGL(k_ctm_elbo_topics, dim3(1), dim3(256), lds2, ctx->stream, dm, tpr, m->gamma.p + r * m->GM, m->Elnphi.p + r * m->GM,
                       m->invSigma.p + r * MKz * MKz, acc + 5, m->big ? m->big_scratch.p + r * 2 * MKz * MKz : (double*)nullptr);
    MMM_LAUNCH_CHECK(ctx);
    if ((rc = mmm_allreduce_sum(ctx, acc, 5))) return rc;
    double h[8];
    MMM_HIP(ctx, hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = mmm_p2p_check(ctx))) return rc;      // (the ELBO's sums went through the ranks' exchange: a peer that never came is an error)
    const double MK = dm.MK, Dg = m->Dglobal, l2pi = log(2.0 * M_PI);
    double t[7];
    t[0] = h[5];                                              // ElnPϕ  MMCTM.jl:271-284
    t[1] = h[0] + 0.5 * Dg * (h[7] - MK * l2pi);              // ElnPη  MMCTM.jl:286-300
    t[2] = h[1];                                              // ElnPZ  MMCTM.jl:302-316
    t[3] = h[2];                                              // ElnPX  MMCTM.jl:318-336
    t[4] = h[6];                                              // ElnQϕ  MMCTM.jl:338-350
    t[5] = h[3] - 0.5 * Dg * MK * (l2pi + 1.0);               // ElnQη  MMCTM.jl:352-358
    t[6] = h[4];                                              // ElnQZ  MMCTM.jl:360-370
    if (terms) memcpy(terms, t, sizeof t);
    *elbo = t[0] + t[1] + t[2] + t[3] - t[4] - t[5] - t[6];
    return MMM_OK;
}

// fit! for the replicas of a scope, in lock step: every pass advances all still-active replicas; a replica leaves when
// its stopping rule fires (MMCTM.jl:485 + common.jl:48-51).  ll_hist: [nrep][maxiter][M].
static int fit_scope(mmm_ctm* m, Scope sc, int maxiter, double tol, int update_sigma, double* ll_hist, int* n_iter, int* converged,
                     int infer_flags = -1)
{
    mmm_ctx* ctx = m->ctx;
    const int M = m->dm.M, nrep = sc.nrep, rep0 = sc.rep0;
    int rc;
    std::vector<int> base(nrep), done(nrep, 0);
    for (int i = 0; i < nrep; ++i) { base[i] = m->n_hist[rep0 + i]; converged[i] = 0; }
    for (int i = 1; i < nrep; ++i) MMM_CHECK(ctx, base[i] == base[0], "mmm_ctm_fit_batch: replicas have different histories (%d vs %d passes)", base[i], base[0]);
    // The stopping rule runs on the device (k_ll_finish / k_ll_store clear a replica's `active` flag; every launch of the scope skips
    // inactive replicas), so the host never has to wait for a pass before enqueueing the next: it reads the flags -- and the status
    // words of update_Σ! -- from in-stream snapshots taken every few passes and examined one snapshot late, and stops enqueueing
    // when no replica is left.  Cost: a few passes of no-op launches after the last replica has stopped.
    for (int i = 0; i < nrep; ++i) m->h_active[rep0 + i] = 1;
    if ((rc = upload_active(m))) return rc;
    sc.active = m->active.p + rep0;
    MMM_HIP(ctx, hipMemsetAsync(m->npass.p + rep0, 0, sizeof(int) * nrep, ctx->stream));
    if (!m->pin_flags) MMM_HIP(ctx, hipHostMalloc((void**)&m->pin_flags, sizeof(int) * 4 * (size_t)m->R, hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) if (!ctx->pin_ev[i]) MMM_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
    constexpr int kSnapEvery = 4;       // => at most 2 * kSnapEvery no-op passes after the last replica has stopped
    int pass = 0, slot = 0;
    bool have_prev = false;
    m->stop_tol = tol;
    while (pass < maxiter) {
        m->stop_enable = (pass + 1 > 10) ? 1 : 0;          // the rule needs > 10 rows (MMCTM.jl:481)
        rc = infer_flags < 0 ? fused_pass(m, sc, update_sigma) : frozen_pass(m, sc, infer_flags);
        m->stop_enable = 0;
        if (rc) return rc;
        ++pass;
        if (pass <= 10 || (pass < maxiter && (pass - 11) % kSnapEvery != 0)) continue;      // a snapshot costs the host two copies and an event
        int* snap = m->pin_flags + (size_t)slot * 2 * m->R;
        MMM_HIP(ctx, hipMemcpyAsync(snap, m->active.p + rep0, sizeof(int) * nrep, hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipMemcpyAsync(snap + m->R, m->status.p + rep0, sizeof(int) * nrep, hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipEventRecord(ctx->pin_ev[slot], ctx->stream));
        if (have_prev) {
            MMM_HIP(ctx, hipEventSynchronize(ctx->pin_ev[slot ^ 1]));
            const int* prev = m->pin_flags + (size_t)(slot ^ 1) * 2 * m->R;
            int nactive = 0, bad = 0;
            for (int i = 0; i < nrep; ++i) { nactive += prev[i] != 0; bad |= prev[m->R + i]; }
            if (nactive == 0 || bad) break;
        }
        have_prev = true; slot ^= 1;
    }
    // settle: status (singular Σ), pass counts, final flags
    if ((rc = check_status(m, Scope{rep0, nrep, nullptr}))) return rc;
    std::vector<int> hn(nrep), ha(nrep);
    MMM_HIP(ctx, hipMemcpyAsync(hn.data(), m->npass.p + rep0, sizeof(int) * nrep, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipMemcpyAsync(ha.data(), m->active.p + rep0, sizeof(int) * nrep, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int maxdone = 0;
    for (int i = 0; i < nrep; ++i) {
        done[i] = hn[i]; converged[i] = ha[i] ? 0 : 1;
        m->h_active[rep0 + i] = ha[i];
        m->n_hist[rep0 + i] = base[i] + done[i];           // the host counted the no-op passes of replicas that had stopped
        maxdone = std::max(maxdone, done[i]);
        n_iter[i] = done[i];
    }
    if (ll_hist && maxdone > 0) {
        std::vector<double> ll((size_t)nrep * maxdone * M);
        MMM_HIP(ctx, hipMemcpy2DAsync(ll.data(), sizeof(double) * maxdone * M, m->ll_hist.p + ((size_t)rep0 * m->cap_hist + base[0]) * M,
                                      sizeof(double) * m->cap_hist * M, sizeof(double) * maxdone * M, (size_t)nrep, hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < nrep; ++i) memcpy(ll_hist + (size_t)i * maxiter * M, ll.data() + (size_t)i * maxdone * M, sizeof(double) * done[i] * M);
    }
    return MMM_OK;
}

int mmm_ctm_fit(mmm_ctm* m, int maxiter, double tol, int update_sigma, double* ll_hist, int* n_iter, int* converged, double* elbo)
{
    if (!m || !n_iter || !converged) return MMM_ERR_ARG;
    MMM_CHECK(m->ctx, maxiter >= 1, "mmm_ctm_fit: maxiter < 1");
    int rc = prep(m);
    if (rc || (rc = fit_scope(m, one(m), maxiter, tol, update_sigma, ll_hist, n_iter, converged))) return rc;
    if (elbo) return mmm_ctm_elbo(m, elbo, nullptr);
    return MMM_OK;
}

int mmm_ctm_infer(mmm_ctm* m, int flags, int maxiter, double tol, double* ll_hist, int* n_iter, int* converged)
{
    if (!m || !n_iter || !converged) return MMM_ERR_ARG;
    MMM_CHECK(m->ctx, maxiter >= 1, "mmm_ctm_infer: maxiter < 1");
    MMM_CHECK(m->ctx, (flags & ~(MMM_INFER_UNSMOOTHED | MMM_INFER_FIT_GAUSSIAN)) == 0, "mmm_ctm_infer: unknown flags %d", flags);
    MMM_CHECK(m->ctx, !((flags & MMM_INFER_UNSMOOTHED) && m->immctm), "mmm_ctm_infer: IMMCTM has no phi field, hence no unsmoothed_update_θ! (IMMCTM.jl)");
    int rc = prep(m);
    if (rc) return rc;
    return fit_scope(m, one(m), maxiter, tol, 1, ll_hist, n_iter, converged, flags);
}

int mmm_ctm_fit_batch(mmm_ctm* m, int maxiter, double tol, int update_sigma, double* ll_hist, int* n_iter, int* converged, double* elbo)
{
    if (!m || !n_iter || !converged) return MMM_ERR_ARG;
    MMM_CHECK(m->ctx, maxiter >= 1, "mmm_ctm_fit_batch: maxiter < 1");
    int rc = prep(m);
    if (rc || (rc = fit_scope(m, all(m), maxiter, tol, update_sigma, ll_hist, n_iter, converged))) return rc;
    if (elbo) {
        const int keep = m->sel;
        for (int r = 0; r < m->R; ++r) {
            m->sel = r;
            if ((rc = mmm_ctm_elbo(m, elbo + r, nullptr))) { m->sel = keep; return rc; }
        }
        m->sel = keep;
    }
    return MMM_OK;
}

} // extern "C"
