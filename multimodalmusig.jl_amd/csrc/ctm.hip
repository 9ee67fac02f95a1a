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

enum { F_ZETA = 1, F_THETA_COMPUTE = 2, F_THETA_STORED = 4, F_THETA_STORE = 8, F_NU = 16, F_LAMBDA = 32, F_SLAB = 64, F_ORDER_LAM = 128 };

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

#include "ctm_estep.cuh"

#include "ctm_big.cuh"

#include "ctm_mstep.cuh"

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

// shard sizes (documents, on a 256-CU device) below which the solve phase takes more lanes per document (create_impl; measured: profiles/r05_solve_layouts.jsonl)
constexpr int kSolve10Lanes8Below = 75000;      // sum K = 10: 2 lanes x 5 coordinates -> 8 lanes x 2
constexpr int kSolve28Lanes32Below = 9000;
constexpr int kSolve28Waves4From = 40000;         // sum K = 28 as 16 x 2: three -> four persistent waves per SIMD        // sum K = 28: 16 lanes x 2 coordinates -> 32 lanes x 1

// LDS of the Gaussian M-step block: the augmented matrix twice (block_inverse_pipelined, sum K <= 32) or A and its inverse (block_inverse_wide)
inline size_t gauss_lds_doubles(int MK) { return (size_t)(MK <= 32 ? 4 : 2) * MK * MK; }

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
            // round 5, small shards (what one GPU of an N-GPU strong run holds): more lanes per document, so that the chip still has a wave per
            // SIMD and a document's chain of evaluations is shorter -- sum K = 10: 8 lanes, 5 of them x 2 coordinates (8 slots per wave instead
            // of 32); sum K = 28: 32 lanes x 1 coordinate (2 slots instead of 4)
            if (m->dm.MK == 10 && m->Ls == 8) return go(k_ctm_solve_cpl<10, 8, 3, false>);
            if (m->dm.MK == 28 && m->Ls == 32) return go(k_ctm_solve_cpl<28, 32, 4, false>);
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
        if (!mmm_off(m->tune, MMM_OFF_CTM_SOLVE_ORDER)) a.flags |= F_ORDER_LAM;      // the lambda solves' documents by the previous pass' evaluation counts (order_range)
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
                m->big ? m->big_scratch.p + r0 * 2 * MK * MK : nullptr, mmm_off(m->tune, MMM_OFF_CTM_PIPE_GAUSS) ? 1 : 0};
    a.tp.alpha += r0 * m->nalpha;      // host-side copy of the argument struct: fine
    return a;
}

int run_mstep(mmm_ctm* m, Scope sc, int do_mu, int do_sigma, int do_gamma, int gamma_from_stats)
{
    mmm_ctx* ctx = m->ctx;
    const size_t MK = m->dm.MK;
    const MstepArgs a = mstep_args(m, sc, do_mu, do_sigma, do_gamma, gamma_from_stats);
    const size_t lds = m->big ? 0 : sizeof(double) * gauss_lds_doubles((int)MK);
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
                                                 gauss ? gauss_lds_doubles(m->dm.MK) : (size_t)0);
    if (m->big) {      // sum K > 64: the Gaussian M-step as its own launch (device-memory inversion), then the generic props / ll sweep
        if (gauss) { int rc = run_mstep(m, sc, gauss_mu, gauss_sigma, 0, 0); if (rc) return rc; }
        hipLaunchKernelGGL(k_ctm_loglik_big, dim3(m->grid_s, sc.nrep), dim3(kBlockS), sizeof(double) * kWavesS * 64, ctx->stream, m->dev(), m->lambda.p + r0 * m->sDMK(),
                           m->phieff.p + r0 * m->dm.GT, m->props.p + r0 * m->sDMK(), m->llpart.p + r0 * m->grid_s * M, compute_ll ? 1 : 0, sc.active);
    } else if (m->tdense && !mmm_off(m->tune, MMM_OFF_CTM_LL_ROWS)) {      // dense corpora: the sweep over rows of counts
        DenseRows dr{};
        int kmx = 8, vt = 0;
        for (int i = 0; i < M; ++i) { dr.rows[i] = m->trows[i].p; dr.SL[i] = m->tSL[i]; dr.tpoff[i] = vt; vt += 16 * m->tSL[i]; kmx = std::max(kmx, theta_dense_kmx(m->dm.K[i])); }
        dr.tpoff[M] = vt;
        const size_t ldsd = sizeof(double) * std::max((size_t)vt * kmx + (size_t)kWavesS * 4 * kmx + 2 + MMM_LOGTAB_N, gauss ? gauss_lds_doubles(m->dm.MK) : (size_t)0);
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
    // (one block short of four per CU: the log-likelihood launch carries the Gaussian M-step as one block more, and a 1,025th block waits for a
    // slot -- a second round of one block: + 3.3 us at config 5, + 4.6 us at config 4)
    m->grid_s = std::max(1, std::min((D + kWavesS - 1) / kWavesS, ncu * 4 - 1));
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
        const int want = ctx->tune.solve_lanes;      // 0: by shape and corpus size
        const bool lockstep10 = dm.MK == 10 && want == 16;      // sum K = 10 with solve_lanes = 16: the packed / 16-lane lock-step build (A/B)
        if (!mmm_off(ctx->tune, MMM_OFF_CTM_CPL) && !lockstep10) {
            // Which layout: the full-size configurations keep the layouts of rounds 2-3 (most documents per wave trip); a SHARD -- D small
            // enough that those layouts leave SIMDs without a wave or slots without a document -- takes more lanes per document
            // (profiles/r05_shard_sizes.jsonl: sum K = 10 at 12,505 documents 2 x 5 -> 8 x 2; sum K = 28 at 6,249 documents 16 x 2 -> 32 x 1).
            if (dm.MK == 10) {
                const bool wide8 = want == 8 || (want == 0 && (int64_t)D * R < kSolve10Lanes8Below * (ncu / 256.0));      // (a restart batch fills the chip with its replicas)
                if (wide8) { m->Ls = 8; m->cpl = 2; m->lam_occ = 3; } else { m->Ls = 2; m->cpl = 5; m->lam_occ = 2; }
                m->persist = true;
            } else if (dm.MK == 28) {
                const bool wide32 = want == 32 || (want == 0 && (int64_t)D * R < kSolve28Lanes32Below * (ncu / 256.0));
                // (16 x 2 needs 125 VGPRs: four waves per SIMD fit.  With three or more documents per slot the fourth wave pays -- config 4 at
                // 50,000 documents 0.875 -> 0.861 ms per pass; at 25,000 / 12,500 documents it costs 4 % / 8 %: tools/ab_tuning.py, solve_waves)
                const bool four = (int64_t)D * R >= kSolve28Waves4From * (ncu / 256.0);
                if (wide32) { m->Ls = 32; m->cpl = 1; m->lam_occ = 4; } else { m->Ls = 16; m->cpl = 2; m->lam_occ = four ? 4 : 3; }
                m->persist = true;
            }
        }
    }
    if (m->big) { m->Ls = 64; m->cpl = kBigSlots; m->persist = false; }      // ctm_big.cuh: lane l holds coordinates l + 64 q
    const int Gs = MMM_WAVE / m->Ls;
    m->grid_v = std::max(1, std::min((D + m->waves_s * Gs - 1) / (m->waves_s * Gs), ncu * 8));
    // k_ctm_solve_cpl: persistent waves, each with a contiguous range of documents that its slots work through (a finished slot takes the
    // range's next document): as many as are resident at once (lam_occ per SIMD), fewer when every document has a slot of its own.  (Round 5
    // measured whole numbers of waves per SIMD with more documents than slots each for small shards: slower -- 6,249 documents of config 4:
    // 200 us with one wave per SIMD against 188; the phase is issue-bound per SIMD and a half-filled wave costs a full trip.)
    if (m->persist) {
        const int blocks_full = (D + m->waves_s * Gs - 1) / (m->waves_s * Gs);      // every document a slot of its own
        m->grid_v = std::max(1, std::min(blocks_full, ncu * m->lam_occ));
        if (ctx->tune.solve_waves > 0) m->grid_v = std::max(1, std::min(blocks_full, ncu * ctx->tune.solve_waves));
    }
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
// in one launch, so the per-document form runs the stage and then puts every OTHER document's values back -- the field AND the solver's
// per-document evaluation counters (only document d was logically solved) -- also when the stage fails.  Cost: O(D) per call (a D-sized
// temporary and three device copies), i.e. the reference's `for d in 1:D update_ν!(model, d) end` costs O(D^2) here: the loop belongs to
// mmm_ctm_update_nu(m), which is that loop as ONE launch; the per-document form exists for the reference's tests (test/mmctm.jl:92-199).
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
    const size_t D = (size_t)m->dm.D;
    int* nev = stage == MMM_STAGE_NU ? m->nev_nu.p + (size_t)m->sel * D : (stage == MMM_STAGE_LAMBDA ? m->nev_lam.p + (size_t)m->sel * D : nullptr);
    DevBuf<double> save;
    DevBuf<int> save_nev;
    MMM_HIP(ctx, save.alloc(cnt));
    if (nev) MMM_HIP(ctx, save_nev.alloc(D));
    if (cnt) MMM_HIP(ctx, hipMemcpyAsync(save.p, p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
    if (nev) MMM_HIP(ctx, hipMemcpyAsync(save_nev.p, nev, sizeof(int) * D, hipMemcpyDeviceToDevice, ctx->stream));
    const int rc_stage = stage == MMM_STAGE_ZETA ? mmm_ctm_update_zeta(m) : stage == MMM_STAGE_THETA ? mmm_ctm_update_theta(m)
                         : stage == MMM_STAGE_NU ? mmm_ctm_update_nu(m) : mmm_ctm_update_lambda(m);
    // document d's new values into the saved copy (not after a failed stage: then everything goes back as it was), the saved copy back
    if ((rc = ctm_field(m, field, &p, &cnt))) return rc;
    if (!rc_stage) {
        if (stage == MMM_STAGE_THETA) {
            hipLaunchKernelGGL(k_ctm_copy_doc_theta, dim3(1), dim3(256), 0, ctx->stream, m->dev(), d, p, save.p);
            MMM_LAUNCH_CHECK(ctx);
        } else {
            const size_t w = stage == MMM_STAGE_ZETA ? m->dm.M : m->dm.MK;
            MMM_HIP(ctx, hipMemcpyAsync(save.p + (size_t)d * w, p + (size_t)d * w, sizeof(double) * w, hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (nev) MMM_HIP(ctx, hipMemcpyAsync(save_nev.p + d, nev + d, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (cnt) MMM_HIP(ctx, hipMemcpyAsync(p, save.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
    if (nev) MMM_HIP(ctx, hipMemcpyAsync(nev, save_nev.p, sizeof(int) * D, hipMemcpyDeviceToDevice, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rc_stage;
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
    // (a stored value: evaluations | MMM_NEV_NONFINITE, negated when the cap was hit -- nev_code in ctm_estep.cuh)
    int64_t sa = 0, sb = 0, cap = 0;
    for (int d = 0; d < D; ++d) {
        for (int* v : {&a[d], &b[d]}) {
            const bool capped = *v < 0;
            const int n = (capped ? -*v : *v) & ~MMM_NEV_NONFINITE;
            if (capped) ++cap;
            (v == &a[d] ? sa : sb) += n;
            *v = capped ? -n : n;
        }
    }
    if (n_eval_nu) *n_eval_nu = sa;
    if (n_eval_lambda) *n_eval_lambda = sb;
    if (n_capped) *n_capped = cap;
    if (per_doc_nu && D) memcpy(per_doc_nu, a.data(), sizeof(int) * D);
    if (per_doc_lambda && D) memcpy(per_doc_lambda, b.data(), sizeof(int) * D);
    return MMM_OK;
}

int mmm_ctm_events(mmm_ctm* m, int64_t out[4])
{
    if (!m || !out) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    if (rc) return rc;
    const int D = m->dm.D, M = m->dm.M, nh = m->n_hist[m->sel];
    std::vector<int> a((size_t)D), b((size_t)D);
    std::vector<double> ll((size_t)nh * M);
    if (D) {
        MMM_HIP(ctx, hipMemcpyAsync(a.data(), m->nev_nu.p + (size_t)m->sel * D, sizeof(int) * D, hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipMemcpyAsync(b.data(), m->nev_lam.p + (size_t)m->sel * D, sizeof(int) * D, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (nh) MMM_HIP(ctx, hipMemcpyAsync(ll.data(), m->ll_hist.p + (size_t)m->sel * m->cap_hist * M, sizeof(double) * nh * M, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int d = 0; d < D; ++d)
        for (int v : {a[d], b[d]}) {
            if (v < 0) { ++out[0]; v = -v; }
            if (v & MMM_NEV_NONFINITE) ++out[1];
        }
    for (double v : ll) if (!std::isfinite(v)) ++out[2];
    return MMM_OK;
}

#ifdef MMM_DIAG_STAMPS
int mmm_diag_gauss_stamps(unsigned long long out[96])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gauss_stamps), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : -2;
}
#endif

int mmm_ctm_geometry(const mmm_ctm* m, int out[8])
{
    if (!m || !out) return MMM_ERR_ARG;
    out[0] = m->L; out[1] = m->grid_e; out[2] = m->waves_e; out[3] = m->grid_m; out[4] = m->wide ? 1 : (m->tdense ? 2 : 0); out[5] = m->Ls; out[6] = m->cpl;
    out[7] = m->persist ? m->grid_v * m->waves_s : 0;
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

} // extern "C"

namespace {
// the ELBO launches of the selected replica, its eight sums copied to `h8` in stream order (no synchronisation: a batch enqueues every
// replica's, then waits once -- the scratch the sums pass through is reused in stream order)
int elbo_enqueue(mmm_ctm* m, double* h8)
{
    mmm_ctx* ctx = m->ctx;
    int rc = materialise_theta(m);
    if (rc) return rc;
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
    MMM_HIP(ctx, hipMemcpyAsync(h8, acc, sizeof(double) * 8, hipMemcpyDeviceToHost, ctx->stream));
    return MMM_OK;
}

// the seven terms (MMCTM.jl:264-370) from the eight sums
void elbo_finish(const mmm_ctm* m, const double* h, double* elbo, double terms[7])
{
    const CtmDims& dm = m->dm;
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
}
} // namespace

extern "C" {

int mmm_ctm_elbo(mmm_ctm* m, double* elbo, double terms[7])
{
    if (!m || !elbo) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prep(m);
    double h[8];
    if (rc || (rc = elbo_enqueue(m, h))) return rc;
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = mmm_p2p_check(ctx))) return rc;      // (the ELBO's sums went through the ranks' exchange: a peer that never came is an error)
    elbo_finish(m, h, elbo, terms);
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
    if (elbo) {      // every replica's ELBO launches enqueued back to back (pinned landing area), ONE wait: 256 restarts paid 255 round trips before
        mmm_ctx* ctx = m->ctx;
        const int keep = m->sel;
        double* pin = nullptr;
        MMM_HIP(ctx, hipHostMalloc((void**)&pin, sizeof(double) * 8 * (size_t)m->R, hipHostMallocDefault));
        for (int r = 0; r < m->R && !rc; ++r) { m->sel = r; rc = elbo_enqueue(m, pin + 8 * (size_t)r); }
        m->sel = keep;
        // (the wait also on the way out of a failed enqueue: copies into the landing area may be in flight)
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = mmm_fail(ctx, MMM_ERR_HIP, "mmm_ctm_fit_batch: waiting for the ELBO sums failed");
        if (!rc) rc = mmm_p2p_check(ctx);
        if (!rc) for (int r = 0; r < m->R; ++r) elbo_finish(m, pin + 8 * (size_t)r, elbo + r, nullptr);
        (void)hipHostFree(pin);
        if (rc) return rc;
    }
    return MMM_OK;
}

} // extern "C"
