// lda.hip -- LDA variational EM on gfx950 (replaces the hot path of src/LDA.jl)
//
// Data layout in HBM (per model handle)
//   corpus     doc_ptr int64[D+1]; tc int2[nnz] = (term0, count) interleaved -> one 8-byte load per nonzero
//   topics     lambda, Elnbeta, expElnbeta, beta: [k][v] (= the reference's V x K column-major), V*K doubles,
//              each a ring of 3 slots indexed by (iteration mod 3)
//   documents  gamma, Elntheta: [d][k] (= K x D column-major), rings of 3; theta [d][k] on demand
//   phi        [nnz][K] (= per-doc K x W_d blocks, k fastest); materialised only when asked for
//   ctl        device control block: ticket, stop flag, iteration counter t, ll-history length
//
// One outer iteration t (the body of fit!, LDA.jl:201-209) = TWO launches (k_lda_estep, k_lda_reduce_ll_mstep) for plain LDA with
// V <= 256 on one GPU or over the mailboxes, THREE otherwise (k_lda_estep, k_lda_reduce[_ll], k_lda_mstep / k_ilda_mstep):
//
// k_lda_estep<KP, L, LL, VT, SINGLE> (dominant): a wave handles 64/L documents at a time, L lanes per document (L = 16 for
//   K <= 15), lanes over the document's nonzero terms.
//   * Elntheta_k = psi(gamma_k) - psi(sum gamma) on the first K+1 lanes of the group (LDA.jl:78-80)
//   * phi_kw = a_k B_vk / sum_k a_k B_vk with a_k = exp(Elntheta_k) (K exps per document) and B = exp(Elnbeta)
//     (V*K exps per iteration, staged in LDS): algebraically exp(Elntheta_k + Elnbeta_vk) of LDA.jl:71-74
//     without one exp per (term, topic)
//   * lambda scatter (LDA.jl:103-105) into the wave's private LDS slab [K][V] with ds_add_f64; slabs are summed
//     per block in fixed order and written as one partial per block
//   * gamma of iteration t+1 (LDA.jl:85-87 uses the previous phi) = alpha + sum_w phi_kw n_w is formed in the
//     same pass with DPP row reductions, so phi never round-trips through HBM inside the loop; phi is
//     materialised on demand from (Elntheta_t, Elnbeta_{t-1}), which reproduces the stored phi
//   * LL = true: also the log-likelihood numerator of iteration t-1 (LDA.jl:174-188: needs beta_{t-1}, only known after
//     M-step t-1; "lagged ll") -- used by the frozen-topic passes (and MMM_LDA_LL_IN_ESTEP=1)
// k_lda_reduce_ll_mstep: the two kernels below in one launch -- the reduce blocks of a topic exchange their partial column sums
//   through seq-tagged cells in device memory and run the M-step of their own entries; see the comment at the kernel.
// k_lda_reduce_ll / k_lda_reduce: blocks of (16 entries x 64 slab lanes) sum the per-block partials in fixed order
//   (deterministic); with the mailbox transport they send each statistic to the peer GPUs as it is produced.  The "_ll"
//   launch carries extra blocks that evaluate the lagged log-likelihood beside the reduction (default on one GPU and
//   with the mailboxes).
// k_lda_mstep: one wave per topic (receives the peers' statistics in rank order when folded): lambda = eta + sums,
//   Elnbeta, exp table, beta (LDA.jl:96-112); an extra block finishes ll_{t-1}, applies the convergence test of
//   common.jl:53-56 (device-side stop flag: later launches exit at once) and advances t.  ILDA: k_ilda_mstep instead.
#include <memory>
#include "dev_math.h"
#include "mmm_logtab.h"
#include "mmm_exptab.h"
#include "mmm_internal.h"

namespace {

constexpr int kWavesPerBlock = 4;                    // stage kernels
constexpr int kBlock = kWavesPerBlock * MMM_WAVE;
constexpr int kMaxWavesE = 12;                       // fused E-step kernel: up to 768 threads

struct LdaDev {
    int D, V, K;
    const int64_t* doc_ptr;
    const int2* tc;
    double alpha, eta;
    const int2* ell;      // [D][V] rows padded with (-1, 0), or NULL: lets the ll blocks fetch a document's terms without first
                          // waiting for doc_ptr (built when V <= 128 and no document lists a term twice)
    const int* dense;     // [D][16][Vp / 16] rows of counts, LANE-major: the Vp / 16 slots of lane l (terms l, 16 + l, ...) are contiguous, so a lane
                          // requests its part of a row with one load (row_slot()); or NULL.  The ll blocks read these instead of ell
    int Vp;               // slots per row (16 x slots per lane; 16-bit rows keep an even number of slots per lane)
    const unsigned short* dense16;   // the same rows as 16-bit counts (every count < 65536), or NULL: 2 bytes per term slot
};

// position of term slot w (lane w % 16, the lane's slot w / 16) in a lane-major row of 16 x slp slots
__device__ __forceinline__ int row_slot(int w, int slp) { return (w & 15) * slp + (w >> 4); }
// 16-bit lane-major rows: slot c (0..7) of a lane's part, held as four 32-bit words (ONE 16-byte load; the rows are allocated with 16 bytes to spare)
__device__ __forceinline__ int row16_count(unsigned w0, unsigned w1, unsigned w2, unsigned w3, int c)
{
    const unsigned word = c < 4 ? (c < 2 ? w0 : w1) : (c < 6 ? w2 : w3);
    return (int)((c & 1) ? word >> 16 : word & 0xffffu);
}

struct LdaCtl {
    unsigned int ticket;
    int stop;        // 1 once the convergence criterion was met (later launches exit at once)
    int stop_iter;   // iteration at which it was met
    int t;           // completed iterations
    int n_hist;      // ll values written
    int wait_timeout;   // a device-side wait of the merged reduce + M-step launch gave up (sync_ctl reports it)
    int pad[2];
};

struct Ring { double* s[3]; };

// 16-byte cells {low half | seq} {high half | seq} in device memory: how blocks of one launch hand each other a double without
// a fence or a flag (k_lda_reduce_ll_mstep; the mailbox format of p2p.hip)
__device__ __forceinline__ void cell_store(unsigned long long* c, double v, unsigned int seq)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v), tag = (unsigned long long)seq << 32;
    __hip_atomic_store(c, (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(c + 1, (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double cell_wait(const unsigned long long* c, unsigned int seq, LdaCtl* ctl)
{
    unsigned long long w0 = 0, w1 = 0;
    for (int it = 0; it < (1 << 22); ++it) {
        w0 = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1 = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned int)(w0 >> 32) == seq && (unsigned int)(w1 >> 32) == seq)
            return __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
        __builtin_amdgcn_s_sleep(1);
    }
    ctl->wait_timeout = 1;
    return 0.0;
}

// sum of the cells c[2 (lane + 64 j)], j = 0.. while lane + 64 j < n (n <= 512), added in that order: every polling round asks for ALL the
// cells the lane still misses at once, so the lane is done one memory round trip after its last cell arrives (waiting for them one
// after the other costs a round trip per cell: 157 ll blocks = 3 cells per lane)
__device__ __forceinline__ double cells_wait_sum(const unsigned long long* c, int n, int lane, unsigned int seq, LdaCtl* ctl)
{
    constexpr int MAXJ = 8;
    double val[MAXJ];
    unsigned pending = 0;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) { val[j] = 0.0; if (lane + 64 * j < n) pending |= 1u << j; }
    for (int it = 0; it < (1 << 22) && pending; ++it) {
        unsigned long long w0[MAXJ], w1[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if (pending & (1u << j)) {
                w0[j] = __hip_atomic_load(c + 2 * (lane + 64 * j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w1[j] = __hip_atomic_load(c + 2 * (lane + 64 * j) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if ((pending & (1u << j)) && (unsigned int)(w0[j] >> 32) == seq && (unsigned int)(w1[j] >> 32) == seq) {
                val[j] = __longlong_as_double((long long)((w0[j] & 0xffffffffull) | (w1[j] << 32)));
                pending &= ~(1u << j);
            }
        }
        if (pending) __builtin_amdgcn_s_sleep(1);
    }
    if (pending) ctl->wait_timeout = 1;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) v += val[j];
    return v;
}

constexpr int kIldaMaxI = 8, kIldaMaxSJ = 512;

struct EstepArgs {
    LdaDev c;
    const LdaCtl* ctl;
    Ring gamma, Elntheta, expElnbeta, beta;
    double* partial;   // [gridDim][K*pstride]
    double* llpart;    // [gridDim]
    int do_ll;
    int t;             // this pass (1-based); the host's count, valid unless ctl->stop is set
    int pstride;       // row stride of a block's partial: V, or V rounded up to 16 for k_lda_reduce_ll_mstep (pad entries stay 0)
    const double* aexp; // single-step build: a = exp(Elntheta_t) [D][K], formed (with Elntheta_t) by the PREVIOUS pass's merged launch (its
                       // prologue blocks, k_lda_reduce_ll_mstep) -- the kernel then starts at the term phase; NULL: it forms them itself
};

#ifdef MMM_DIAG_STAMPS
// diagnostic build only (make diag): s_memtime stamps of block 0 / wave 0 through the fused E-step kernel.
__device__ unsigned long long g_lda_stamps[16];
#define MMM_STAMP(i)                                                                                         \
    do {                                                                                                     \
        unsigned long long t_;                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_lda_stamps[i] = t_;                                       \
        if ((i) == 0 || (i) == 7) {                                                                          \
            unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                                        \
            if (blockIdx.x == 0 && threadIdx.x == 0) g_lda_stamps[8 + ((i) != 0)] = r_;                     \
            if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_lda_stamps[10 + ((i) != 0)] = r_;        \
        }                                                                                                    \
    } while (0)
// stamps of the merged reduce + ll + M-step launch: s_memrealtime (100 MHz) of one chosen wave per role
__device__ unsigned long long g_red_stamps[32];
// start / numerator-posted time of every ll block of the last merged launch (s_memrealtime): where the launch's tail comes from
__device__ unsigned long long g_ll_times[2][512];
#define MMM_LLSTAMP(which, lb)                                                                               \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                                            \
        if (threadIdx.x == 0 && threadIdx.y == 0 && (lb) < 512) g_ll_times[which][lb] = r_;                  \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define MMM_RSTAMP(cond, i)                                                                                  \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
        unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                                            \
        if (cond) g_red_stamps[i] = r_;                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#else
#define MMM_STAMP(i) do { } while (0)
#define MMM_RSTAMP(cond, i) do { } while (0)
#define MMM_LLSTAMP(which, lb) do { } while (0)
#endif

#include "lda_estep.cuh"

#include "lda_reduce.cuh"

#include "lda_stage.cuh"

} // namespace

// ---------------------------------------------------------------------------------------------------------
struct mmm_lda {
    mmm_ctx* ctx = nullptr;
    mmm_tuning_opts tune{};      // the caller's choices at create time (mmm_ctx_set_tuning)
    int D = 0, V = 0, K = 0, KP = 0, L = 16;
    int64_t nnz = 0;
    double alpha = 0, eta = 0;
    double Nglobal = 0, Dglobal = 0;
    DevBuf<int64_t> doc_ptr; DevBuf<int2> tc, tc_ell;
    DevBuf<unsigned short> cnt16;   // the rows as 16-bit counts, when every count fits (then cnt_dense is not built)
    DevBuf<int> cnt_dense;      // dense rows [D][16 SL] of counts (k_lda_estep_dense), or empty
    bool dense = false; int SL = 0, SLs = 0; size_t lds_d = 0; bool attr_d = false;     // SL: term slots per lane of a 16-lane group; SLs: as stored (LdaDev::Vp / 16)
    bool drows = false;         // cnt_dense exists (dense-row E-step build, or rows for the single-step build and the ll blocks)
    DevBuf<double> lambda[3], Elnbeta[3], expElnbeta[3], beta[3], gamma[3], Elntheta[3];
    DevBuf<double> theta, phi;
    DevBuf<double> partial, stats[2], scratch, llpart, llpart2, elbopart, ll_hist;
    int n_ll = 0;               // ll blocks of the reduce launch
    DevBuf<LdaCtl> ctl;
    // host mirror of the device control block (exact after sync_ctl)
    int t = 0, n_hist = 0, cap_hist = 0;
    bool inflight = false;      // fused passes enqueued since the last sync_ctl
    bool phi_valid = false;     // phi buffer == phi of the current state
    bool phi_from_prev = false; // current phi is implied by (Elntheta_t, Elnbeta_{t-1}) (after fused passes)
    bool gnext_valid = false;   // gamma[(t+1)%3] holds gamma_{t+1}
    bool ll_pending = false;    // the ll of pass t has not been recorded yet
    bool theta_valid = false;
    bool attr_e[2] = {false, false}, attr_m = false, attr_mm[3] = {false, false, false};
    int cap_mm[3] = {-1, -1, -1}, cap_m = -1;      // blocks of the reduce / merged launches that can be resident together (residency_cap)
    DevBuf<unsigned long long> cells;   // k_lda_reduce_ll_mstep: [2 * (512 + 512)] exchange cells
    DevBuf<unsigned long long> fcells;  // ILDA: [2 * 512 * 16] fold cells
    unsigned int kseq = 0;              // sequence number of its launches
    bool stop_seen = false;     // the device stop flag may be set
    bool lag_ll = true;         // the passes in flight evaluate the ll one pass late (training); false: frozen-topic passes
    bool phi_table_beta = false; // phi of the current state is exp(Elntheta) .* beta normalised (unsmoothed_update_ϕ!, LDA.jl:226)
    bool single_step = false;   // one step per wave: the grid covers every document
    bool wide = false;          // K*V tables larger than LDS: k_lda_estep_wide + k_lda_stats_terms, tables through L2
    DevBuf<int64_t> term_ptr;           // wide: the postings of term v are tpost[term_ptr[v] .. term_ptr[v+1])
    DevBuf<int2> tpost;                 // (document, count), documents ascending within a term
    DevBuf<double> aexp;                // wide: exp(Elntheta_t), D x KP, written by the document sweep for the term sweep
    DevBuf<double> aexp_next;           // single-step build: exp(Elntheta_{t+1}), D x K, written by the merged launch of pass t (its prologue blocks)
    int aexp_for = -1;                  // the pass whose Elntheta / aexp_next the last merged launch of THIS call has formed (prepare_call resets it)
    bool pro_used = false;              // some merged launch of this handle has run a next pass's prologue (mmm_lda_prologue_moved)
    DevBuf<double> tabT;                // wide: [2][V][KP] term-major copies of the pass's exp(Elnbeta) and beta tables
    int stats_waves = 1;                // waves per term block of k_lda_stats_terms
    bool attr_big = false, attr_bigs = false;
    int grid_e = 1, waves_e = 8, grid_s = 1;
    size_t lds_e = 0, lds_tab = 0;
    // ILDA (src/ILDA.jl): feature-factorised topics; the V x K rings then hold the effective tables
    bool ilda = false;
    IldaDesc ids{};
    DevBuf<int> features;
    DevBuf<double> ilam[3], iEln[3], ibeta[3];      // model layout, ring like the V x K tables
    LdaDev dev() const { return LdaDev{D, V, K, doc_ptr.p, tc.p, alpha, eta, tc_ell.p, (drows && !cnt16.p) ? cnt_dense.p : nullptr, 16 * SLs, cnt16.p}; }
    int cur() const { return t % 3; }
    Ring ring(DevBuf<double>* b) const { return Ring{{b[0].p, b[1].p, b[2].p}}; }
};

namespace {

int pick_kp(int K)
{
    static const int opts[] = {2, 4, 6, 8, 10, 12, 16, 20, 24, 32};
    for (int o : opts) if (K <= o) return o;
    if (K <= 64 * kLdaSlots) return (K + 1) & ~1;      // 33..256 topics: the rolled-loop kernels (k_lda_estep_big, k_lda_stats_big), wide path only
    return -1;
}

#define MMM_KP_SWITCH(m, ...)                                                                                   \
    switch ((m)->KP) {                                                                                          \
        case 2: { constexpr int KPV = 2; __VA_ARGS__ } break;                                                          \
        case 4: { constexpr int KPV = 4; __VA_ARGS__ } break;                                                          \
        case 6: { constexpr int KPV = 6; __VA_ARGS__ } break;                                                          \
        case 8: { constexpr int KPV = 8; __VA_ARGS__ } break;                                                          \
        case 10: { constexpr int KPV = 10; __VA_ARGS__ } break;                                                        \
        case 12: { constexpr int KPV = 12; __VA_ARGS__ } break;                                                        \
        case 16: { constexpr int KPV = 16; __VA_ARGS__ } break;                                                        \
        case 20: { constexpr int KPV = 20; __VA_ARGS__ } break;                                                        \
        case 24: { constexpr int KPV = 24; __VA_ARGS__ } break;                                                        \
        case 32: { constexpr int KPV = 32; __VA_ARGS__ } break;                                                        \
        default: return mmm_fail((m)->ctx, MMM_ERR_UNSUPPORTED, "LDA: K=%d has no build on this path (the LDS builds stop at 32 topics)", (m)->K);    \
    }

template <typename Kern>
int set_lds(mmm_ctx* ctx, Kern kern, size_t lds)
{
    if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return MMM_OK;
}

// How many blocks of a launch whose blocks WAIT for each other (cells) can be resident at once: occupancy x CUs.  Launches of
// k_lda_reduce_ll_mstep / k_lda_reduce_ll never exceed it, so every block a waiting block waits for is on the chip -- the waits
// cannot deadlock whatever order the dispatcher picks.  mmm_tuning_opts.resident_cap lowers the figure (tests).
template <class Kern>
int residency_cap(mmm_lda* m, Kern kern, size_t lds, int* cap)
{
    mmm_ctx* ctx = m->ctx;
    int nb = 0;
    MMM_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 1024, lds));
    // the blocks of such a launch wait for each other, so every one of them must be on the chip at once.  nb x #CU holds when the device is
    // ours alone; a few CUs are left out of the count so that a short kernel of another stream or process does not turn a wait into a timeout
    *cap = nb * std::max(1, ctx->num_cu - 4);
    if (m->tune.resident_cap > 0) *cap = std::min(*cap, m->tune.resident_cap);
    return MMM_OK;
}

template <int KPV, int LV, bool LLV, int VT, bool SG>
int go_estep3(mmm_lda* m, const EstepArgs& a)
{
    mmm_ctx* ctx = m->ctx;
    auto k = k_lda_estep<KPV, LV, LLV, VT, SG>;
    if (!m->attr_e[LLV]) { int rc = set_lds(ctx, k, m->lds_e); if (rc) return rc; m->attr_e[LLV] = true; }
    hipLaunchKernelGGL(k, dim3(m->grid_e), dim3(m->waves_e * MMM_WAVE), m->lds_e, ctx->stream, a);
    return MMM_OK;
}

template <int KPV, int LV, bool LLV, int VT>
int go_estep2(mmm_lda* m, const EstepArgs& a)
{
    if constexpr (LV == 16 && KPV <= 12) { if (m->single_step) return go_estep3<KPV, LV, LLV, VT, true>(m, a); }
    return go_estep3<KPV, LV, LLV, VT, false>(m, a);
}

template <int KPV, int LV, bool LLV>
int go_estep(mmm_lda* m, const EstepArgs& a)
{
    // the 96-term SNV vocabulary (data/brca-eu_snv_counts.tsv; every BASELINE config) gets compile-time strides
    if constexpr (LV == 16 && (KPV == 8 || KPV == 10)) { if (m->V == 96) return go_estep2<KPV, LV, LLV, 96>(m, a); }
    return go_estep2<KPV, LV, LLV, 0>(m, a);
}

template <int KPV, int SLV>
int go_dense(mmm_lda* m, const EstepArgs& a)
{
    if constexpr (KPV * SLV <= 64) {
        mmm_ctx* ctx = m->ctx;
        const bool c16 = m->cnt16.p != nullptr;
        auto k = c16 ? k_lda_estep_dense<KPV, SLV, true> : k_lda_estep_dense<KPV, SLV, false>;
        if (!m->attr_d) { int rc = set_lds(ctx, k, m->lds_d); if (rc) return rc; m->attr_d = true; }
        hipLaunchKernelGGL(k, dim3(m->grid_e), dim3(m->waves_e * MMM_WAVE), m->lds_d, ctx->stream, a, (const int*)m->cnt_dense.p, (const unsigned short*)m->cnt16.p);
        return MMM_OK;
    } else return mmm_fail(m->ctx, MMM_ERR_UNSUPPORTED, "LDA: no dense-row build for KP=%d SL=%d", KPV, SLV);
}

// term slots per lane of the dense-row build that covers V terms (0: none)
int dense_slots(int V) { return V <= 32 ? 2 : (V <= 48 ? 3 : (V <= 96 ? 6 : (V <= 128 ? 8 : 0))); }

int launch_estep(mmm_lda* m, const EstepArgs& a)
{
    mmm_ctx* ctx = m->ctx;
    int rc = MMM_OK;
    if (m->dense && !a.do_ll) {
        MMM_KP_SWITCH(m, {
            if constexpr (KPV >= 4 && KPV <= 16) {
                switch (m->SL) {
                    case 2: rc = go_dense<KPV, 2>(m, a); break;
                    case 3: rc = go_dense<KPV, 3>(m, a); break;
                    case 6: rc = go_dense<KPV, 6>(m, a); break;
                    default: rc = go_dense<KPV, 8>(m, a); break;
                }
            }
        })
        if (rc) return rc;
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    if (m->wide) {
        const size_t n = (size_t)m->V * m->KP;
        const int slot = (a.t + 2) % 3;
        hipLaunchKernelGGL(k_lda_tables_by_term, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, m->V, m->K, m->KP,
                           (const double*)a.expElnbeta.s[slot], a.do_ll ? (const double*)a.beta.s[slot] : (const double*)nullptr, m->tabT.p, m->tabT.p + n);
        if (m->KP > 32) {
            const size_t per = sizeof(double) * ((size_t)m->KP * MMM_WAVE + 2 * m->KP);
            const int nw = (int)std::max<size_t>(1, std::min<size_t>(kWavesPerBlock, (150 * 1024) / per));
            const size_t lds = per * nw;
            if (!m->attr_big) { if ((rc = set_lds(ctx, k_lda_estep_big, lds))) return rc; m->attr_big = true; }
            // (the grid stays m->grid_e: the ll partials are summed over that many blocks; the blocks stride over the documents)
            hipLaunchKernelGGL(k_lda_estep_big, dim3(m->grid_e), dim3(nw * MMM_WAVE), lds, ctx->stream, a, m->aexp.p, m->tabT.p, m->tabT.p + n, m->KP);
        } else
        MMM_KP_SWITCH(m, { hipLaunchKernelGGL(k_lda_estep_wide<KPV>, dim3(m->grid_e), dim3(kBlock), 0, ctx->stream, a, m->aexp.p, m->tabT.p, m->tabT.p + n); })
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    MMM_KP_SWITCH(m, {
        if (m->L == 16) { if constexpr (KPV <= 16) rc = a.do_ll ? go_estep<KPV, 16, true>(m, a) : go_estep<KPV, 16, false>(m, a); }
        else if (m->L == 32) { if constexpr (KPV >= 16) rc = a.do_ll ? go_estep<KPV, 32, true>(m, a) : go_estep<KPV, 32, false>(m, a); }
        else { if constexpr (KPV == 32) rc = a.do_ll ? go_estep<KPV, 64, true>(m, a) : go_estep<KPV, 64, false>(m, a); }
    })
    if (rc) return rc;
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

int launch_phi(mmm_lda* m, const double* Elntheta, const double* expElnbeta)
{
    mmm_ctx* ctx = m->ctx;
    if (m->K > 64) {        // 65..256 topics: a_k in LDS, rolled loops
        hipLaunchKernelGGL(k_lda_phi_big, dim3(m->grid_s), dim3(kBlock), sizeof(double) * kWavesPerBlock * m->K, ctx->stream, m->dev(), Elntheta, expElnbeta, m->phi.p);
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    if (m->KP > 32) {       // 33..64 topics (off the iteration path): the 64-topic build, topics >= K skipped
        hipLaunchKernelGGL((k_lda_phi<64, false>), dim3(m->grid_s), dim3(kBlock), 0, ctx->stream, m->dev(), Elntheta, expElnbeta, m->phi.p);
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    MMM_KP_SWITCH(m, {
        if (m->wide) hipLaunchKernelGGL((k_lda_phi<KPV, false>), dim3(m->grid_s), dim3(kBlock), 0, ctx->stream, m->dev(), Elntheta, expElnbeta, m->phi.p);
        else {
            auto k = k_lda_phi<KPV, true>; int rc;
            if ((rc = set_lds(ctx, k, m->lds_tab))) return rc;
            hipLaunchKernelGGL(k, dim3(m->grid_s), dim3(kBlock), m->lds_tab, ctx->stream, m->dev(), Elntheta, expElnbeta, m->phi.p);
        }
    })
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

int launch_loglik(mmm_lda* m, const double* gamma, const double* beta, double* theta, int compute_ll)
{
    mmm_ctx* ctx = m->ctx;
    const size_t lds = compute_ll ? m->lds_tab : 0;
    if (m->K > 64) {
        hipLaunchKernelGGL(k_lda_loglik_big, dim3(m->grid_s), dim3(kBlock), sizeof(double) * kWavesPerBlock * m->K, ctx->stream, m->dev(), gamma, beta, theta, m->llpart.p, compute_ll);
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    if (m->KP > 32) {
        hipLaunchKernelGGL((k_lda_loglik<64, false>), dim3(m->grid_s), dim3(kBlock), 0, ctx->stream, m->dev(), gamma, beta, theta, m->llpart.p, compute_ll);
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    MMM_KP_SWITCH(m, {
        if (m->wide) hipLaunchKernelGGL((k_lda_loglik<KPV, false>), dim3(m->grid_s), dim3(kBlock), 0, ctx->stream, m->dev(), gamma, beta, theta, m->llpart.p, compute_ll);
        else {
            auto k = k_lda_loglik<KPV, true>; int rc;
            if ((rc = set_lds(ctx, k, lds))) return rc;
            hipLaunchKernelGGL(k, dim3(m->grid_s), dim3(kBlock), lds, ctx->stream, m->dev(), gamma, beta, theta, m->llpart.p, compute_ll);
        }
    })
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

// bring the host mirror of the control block up to date (needed after fused passes that may have stopped early)
int sync_ctl(mmm_lda* m)
{
    if (!m->inflight) return MMM_OK;
    mmm_ctx* ctx = m->ctx;
    LdaCtl h;
    MMM_HIP(ctx, hipMemcpyAsync(&h, m->ctl.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    { int rc = mmm_p2p_check(ctx); if (rc) return rc; }
    if (h.wait_timeout) return mmm_fail(ctx, MMM_ERR_HIP, "LDA: a block of the merged reduce + M-step launch gave up waiting for its neighbours");
    const bool stopped = h.stop != 0;
    if (stopped) m->stop_seen = true;
    m->t = h.t; m->n_hist = h.n_hist;
    m->inflight = false;
    m->ll_pending = m->lag_ll && !stopped;   // after a stop the ll of the kept iteration is already recorded
    return MMM_OK;
}

// phi of the current state: after fused passes it is softmax_k(Elntheta_t + Elnbeta_{t-1}) (LDA.jl:69-76)
int materialise_phi(mmm_lda* m)
{
    int rc = sync_ctl(m);
    if (rc) return rc;
    if (m->phi_valid) return MMM_OK;
    const int c = m->cur(), p = (m->t + 2) % 3;
    const double* table = m->phi_from_prev ? (m->phi_table_beta ? m->beta[p].p : m->expElnbeta[p].p) : m->expElnbeta[c].p;
    if ((rc = launch_phi(m, m->Elntheta[c].p, table))) return rc;
    m->phi_valid = true;
    return MMM_OK;
}

int ensure_hist(mmm_lda* m, int extra)
{
    if (m->n_hist + extra + 2 <= m->cap_hist) return MMM_OK;
    int rc = sync_ctl(m);
    if (rc) return rc;
    const int cap = std::max(2 * m->cap_hist, m->n_hist + extra + 66);
    DevBuf<double> nb;
    MMM_HIP(m->ctx, nb.alloc(cap));
    if (m->n_hist) MMM_HIP(m->ctx, hipMemcpyAsync(nb.p, m->ll_hist.p, sizeof(double) * m->n_hist, hipMemcpyDeviceToDevice, m->ctx->stream));
    MMM_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    m->ll_hist.swap(nb);
    m->cap_hist = cap;
    return MMM_OK;
}

// record the ll of the current state (LDA.jl:174-188, 209) when the lagged evaluation has not covered it yet
int flush_ll(mmm_lda* m, double* also_dev)
{
    int rc = sync_ctl(m);
    if (rc) return rc;
    if (!m->ll_pending && !also_dev) return MMM_OK;
    mmm_ctx* ctx = m->ctx;
    const int c = m->cur();
    if ((rc = ensure_hist(m, 1))) return rc;
    if ((rc = launch_loglik(m, m->gamma[c].p, m->beta[c].p, m->theta.p, 1))) return rc;
    m->theta_valid = true;
    double* num = m->scratch.p + (size_t)m->V * m->K;
    hipLaunchKernelGGL(k_sum_columns, dim3(1), dim3(64), 0, ctx->stream, m->llpart.p, m->grid_s, 1, num);
    MMM_LAUNCH_CHECK(ctx);
    if ((rc = mmm_allreduce_sum(ctx, num, 1))) return rc;
    const bool push = m->ll_pending;
    hipLaunchKernelGGL(k_ll_push, dim3(1), dim3(1), 0, ctx->stream, m->ctl.p, num, m->Nglobal, push ? m->ll_hist.p : nullptr, also_dev);
    MMM_LAUNCH_CHECK(ctx);
    if (push) { m->n_hist++; m->ll_pending = false; }
    return MMM_OK;
}

// lambda[c] = eta + all-reduced sums; Elnbeta[c], exp table (stage path, in place on the current slot)
int run_topic_update(mmm_lda* m, bool from_sums)
{
    mmm_ctx* ctx = m->ctx;
    const int c = m->cur();
    if (from_sums) { int rc = mmm_allreduce_sum(ctx, m->scratch.p, (size_t)m->V * m->K); if (rc) return rc; }
    if (m->ilda) {
        hipLaunchKernelGGL(k_ilda_mstep, dim3(m->K), dim3(64 * m->ids.I), 0, ctx->stream, m->ids, from_sums ? 0 : 1, m->scratch.p, m->ilam[c].p, m->iEln[c].p,
                           m->ibeta[c].p, m->Elnbeta[c].p, m->expElnbeta[c].p, m->beta[c].p, (const int*)nullptr, 0, ReduceArgs{}, 0);
        MMM_LAUNCH_CHECK(ctx);
        return MMM_OK;
    }
    hipLaunchKernelGGL(k_lda_topic, dim3(m->K), dim3(256), 0, ctx->stream, m->V, m->eta, from_sums ? m->scratch.p : nullptr,
                       m->lambda[c].p, m->Elnbeta[c].p, m->expElnbeta[c].p, m->beta[c].p, 0);
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

int fused_passes(mmm_lda* m, int n_iter, double tol, int conv_base)
{
    mmm_ctx* ctx = m->ctx;
    int rc;
    if ((rc = ensure_hist(m, n_iter))) return rc;
    if (!m->gnext_valid) {
        // update_γ! for the first pass (LDA.jl:82-90) from the resident phi
        if ((rc = materialise_phi(m))) return rc;
        hipLaunchKernelGGL(m->K > 64 ? k_lda_gamma_from_phi_big : k_lda_gamma_from_phi, dim3(m->grid_s), dim3(kBlock), 0, ctx->stream, m->dev(), m->phi.p, m->gamma[(m->t + 1) % 3].p, (double*)nullptr);
        MMM_LAUNCH_CHECK(ctx);
        m->gnext_valid = true;
    }
    const int VK = m->V * m->K;
    for (int it = 0; it < n_iter; ++it) {
        const int t = m->t + 1;
        const int do_ll = (m->ll_pending || it > 0) ? 1 : 0;
        // Where the ll of pass t-1 is evaluated: in extra blocks of the reduce launch (the reduction occupies 60 CUs for ~6 us,
        // the ll sweep fits beside it and the E-step kernel sheds 43 % of its chunk-loop instructions and half its table
        // reads).
        ReduceArgs r{m->partial.p, m->llpart.p, m->grid_e, VK, m->stats[t & 1].p, m->ctl.p, t, m->Nglobal, tol, m->ll_hist.p, do_ll, conv_base, 1};
        r.p2p = 0; r.p2p_seq = 0;
        const bool fold = !mmm_off(m->tune, MMM_OFF_P2P_FOLDED);
        const int Vp = (m->V + 15) & ~15;
        if (fold && !m->ilda && !m->wide && mmm_p2p_begin(ctx, (size_t)Vp * m->K + 1, &r.px, &r.p2p_seq)) r.p2p = 1;      // (k_ilda_mstep does not receive)
        const bool ll_in_k2 = !m->wide;
        const bool via_cells = ll_in_k2 && !r.p2p && mmm_comm_active(ctx);      // RCCL transport
        // V <= 256, plain LDA, one GPU or mailboxes: reduction, ll sweep and M-step in ONE launch (k_lda_reduce_ll_mstep), statistics
        // rows padded to a multiple of 16; MMM_OFF_LDA_MERGED keeps the split kernels (A/B, tests)
        bool merged = !mmm_off(m->tune, MMM_OFF_LDA_MERGED) && ll_in_k2 && (r.p2p || !mmm_comm_active(ctx)) && !m->wide && m->V <= 256 &&
                      (!m->ilda || (m->ids.SJ <= 16 && !mmm_comm_active(ctx)));
        const size_t lds_red = sizeof(double) * ((size_t)m->KP * m->V + 64 * (size_t)m->KP + MMM_LOGTAB_N);      // beta table | theta rows | log table
        int cap = 0;       // residency of the launch whose blocks wait for each other
        if (merged) {
            const int ai = m->ilda ? 2 : r.p2p;
            if (m->cap_mm[ai] < 0) {
                MMM_KP_SWITCH(m, {
                    auto k = m->ilda ? k_lda_reduce_ll_mstep<KPV, false, true> : (r.p2p ? k_lda_reduce_ll_mstep<KPV, true, false> : k_lda_reduce_ll_mstep<KPV, false, false>);
                    if (!m->attr_mm[ai]) { if ((rc = set_lds(ctx, k, lds_red))) return rc; m->attr_mm[ai] = true; }
                    if ((rc = residency_cap(m, k, lds_red, &m->cap_mm[ai]))) return rc;
                })
            }
            cap = m->cap_mm[ai];
            // every reduce block waits for the blocks of its topic, wave 1 of block 0 for the ll blocks: all of them must fit
            if (cap < (Vp * m->K + 15) / 16 + 1) merged = false;
        }
        // Single-step build: two logical reduce blocks (16 entries each) per physical block.  The launch is bounded by its ll blocks, whose
        // sweep is issue-bound per SIMD: at BASELINE config 2, 60 + 192 resident blocks leave 13 busy waves per ll block (one SIMD with four),
        // 30 + 209 leave 12 (three everywhere) -- merged launch 7.7 -> 6.8 us (measured at 9,216 documents before this was built)
        const int epb = (merged && m->single_step && !m->dense && !m->ilda && ((Vp * m->K) / 16) % 2 == 0) ? 2 : 1;
        if (merged) r.VK = Vp * m->K;
        r.llpart2 = m->llpart2.p; r.ll_in_k2 = ll_in_k2 ? 1 : 0;
        r.ll_cells = via_cells ? m->cells.p + 2 * 512 : nullptr; r.ll_seq = via_cells ? ++m->kseq : 0;
        const int docs_per_ll_block = 16 * (MMM_WAVE / (m->KP <= 15 ? 16 : (m->KP <= 31 ? 32 : 64)));
        // as many ll blocks as can be resident beside the reduce blocks (cut to the launch's residency below), at least one busy wave each
        const int waves_ll = (m->D + docs_per_ll_block / 16 - 1) / (docs_per_ll_block / 16);
        const int blocks_ll = (m->D + docs_per_ll_block - 1) / docs_per_ll_block;      // split launches (no residency bound): full blocks
        r.n_ll = (ll_in_k2 && do_ll) ? std::max(1, std::min((merged || via_cells) ? waves_ll : blocks_ll, 512)) : 0;
        if (!merged && via_cells && r.n_ll > 0) {      // RCCL transport: wave 1 of reduce block 0 waits for the ll blocks' cells
            if (m->cap_m < 0) {
                MMM_KP_SWITCH(m, {
                    auto k = k_lda_reduce_ll<KPV>;
                    if (!m->attr_m) { if ((rc = set_lds(ctx, k, lds_red))) return rc; m->attr_m = true; }
                    if ((rc = residency_cap(m, k, lds_red, &m->cap_m))) return rc;
                })
            }
            cap = m->cap_m;
        }
        if (r.n_ll > 0 && (merged || via_cells)) {
            const int nred_ = ((r.VK + 15) / 16) / (merged ? epb : 1);
            if (cap - nred_ < 1) return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "LDA: the reduce launch cannot hold its %d reduce blocks and one ll block at once (%d resident)", nred_, cap);
            r.n_ll = std::min(r.n_ll, cap - nred_);       // the ll blocks stride over the documents: fewer blocks, same sums per block id
        }
        LdaDev edev = m->dev();
        if (mmm_off(m->tune, MMM_OFF_LDA_PADDED_ROWS)) { edev.ell = nullptr; edev.dense = nullptr; edev.dense16 = nullptr; }
        EstepArgs a{edev, m->ctl.p, m->ring(m->gamma), m->ring(m->Elntheta), m->ring(m->expElnbeta), m->ring(m->beta),
                    m->partial.p, m->llpart.p, ll_in_k2 ? 0 : do_ll, t, merged ? Vp : m->V};
        if (m->aexp_for == t) a.aexp = m->aexp_next.p;       // the previous pass's merged launch has run this pass's prologue
        {   // the E-step kernel is idempotent (it reads pass t's inputs and overwrites pass t's outputs), so a profiled span may
            // hold it several times: (span with 2 launches) - (span with 1) is the kernel's duration free of the event overhead
            ProfSpan span(ctx);
            const int reps = span.on ? ctx->prof_repeat : 1;
            for (int q = 0; q < reps && !rc; ++q) rc = launch_estep(m, a);
        }
        if (rc) return rc;
        ProfSpan tail_span(ctx, 1);      // mmm_ctx_profile_select(1): everything of the pass after the E-step kernel
        const int nred = (r.VK + 15) / 16;
        if (merged) {
            const size_t lds = lds_red;
            MergeArgs ms{m->V, m->eta, m->ring(m->lambda), m->ring(m->Elnbeta), m->ring(m->expElnbeta), m->ring(m->beta), m->cells.p, ++m->kseq, nred, 0};
            ms.epb = epb; ms.nredp = nred / epb;
            ms.ll_join = (!mmm_off(m->tune, MMM_OFF_LDA_LL_JOIN) && r.n_ll > 0 && (int64_t)m->D > (int64_t)r.n_ll * docs_per_ll_block && r.n_ll + ms.nredp - 1 <= 512) ? 1 : 0;
            ms.n_ll = r.n_ll;
            // The next pass's prologue beside this pass's reduction, in the ll blocks: single-step build (every wave of the E-step kernel walks
            // its chain once, the prologue is 2 us of it), every document in exactly one ll block's single step, plain LDA
            ms.pro = (m->single_step && !m->dense && !m->ilda && m->KP <= 12 && m->L == 16 && m->aexp_next.p && r.n_ll > 0 && !ms.ll_join &&
                      (int64_t)r.n_ll * docs_per_ll_block >= (int64_t)m->D && !mmm_off(m->tune, MMM_OFF_LDA_EARLY_PROLOGUE)) ? 1 : 0;
            ms.pro_gamma = m->gamma[(t + 1) % 3].p; ms.pro_Eln = m->Elntheta[(t + 1) % 3].p; ms.pro_a = m->aexp_next.p;
            const int c3 = t % 3;
            IldaMerge im{};
            if (m->ilda) im = IldaMerge{m->ids, m->ilam[c3].p, m->iEln[c3].p, m->ibeta[c3].p, m->fcells.p};
            MMM_KP_SWITCH(m, {
                auto k = m->ilda ? k_lda_reduce_ll_mstep<KPV, false, true> : (r.p2p ? k_lda_reduce_ll_mstep<KPV, true, false> : k_lda_reduce_ll_mstep<KPV, false, false>);
                const int ai = m->ilda ? 2 : r.p2p;
                if (!m->attr_mm[ai]) { if ((rc = set_lds(ctx, k, lds))) return rc; m->attr_mm[ai] = true; }
                hipLaunchKernelGGL(k, dim3(ms.nredp + r.n_ll), dim3(16, 64), lds, ctx->stream, r, m->dev(), m->gamma[(t + 2) % 3].p, m->beta[(t + 2) % 3].p, ms, im);
            })
            MMM_LAUNCH_CHECK(ctx);
            m->aexp_for = ms.pro ? t + 1 : -1;
            if (ms.pro) m->pro_used = true;
            if (do_ll) m->n_hist++;
            m->t = t;
            m->ll_pending = true;
            continue;
        }
        if (m->wide && m->KP > 32) {
            const int nw = std::min(m->stats_waves, m->KP > 128 ? 1 : (m->KP > 64 ? 2 : 4));
            const size_t lds = sizeof(double) * ((size_t)nw * m->KP * MMM_WAVE + m->KP + (size_t)nw * m->KP);
            if (!m->attr_bigs) { if ((rc = set_lds(ctx, k_lda_stats_big, lds))) return rc; m->attr_bigs = true; }
            hipLaunchKernelGGL(k_lda_stats_big, dim3(m->V + 1), dim3(nw * MMM_WAVE), lds, ctx->stream, m->V, m->K, m->KP, m->term_ptr.p, m->tpost.p, m->aexp.p,
                               m->expElnbeta[(t + 2) % 3].p, r);
        } else if (m->wide) {
            MMM_KP_SWITCH(m, { hipLaunchKernelGGL(k_lda_stats_terms<KPV>, dim3(m->V + 1), dim3(m->stats_waves * MMM_WAVE), 0, ctx->stream, m->V, m->K,
                                                  m->term_ptr.p, m->tpost.p, m->aexp.p, m->expElnbeta[(t + 2) % 3].p, r); })
        } else if (r.n_ll > 0) {
            const size_t lds = lds_red;
            MMM_KP_SWITCH(m, {
                auto k = k_lda_reduce_ll<KPV>;
                if (!m->attr_m) { if ((rc = set_lds(ctx, k, lds))) return rc; m->attr_m = true; }
                hipLaunchKernelGGL(k, dim3(nred + r.n_ll), dim3(16, 64), lds, ctx->stream, r, m->dev(), m->gamma[(t + 2) % 3].p, m->beta[(t + 2) % 3].p,
                                   m->llpart2.p, nred);
            })
        } else hipLaunchKernelGGL(k_lda_reduce, dim3(nred), dim3(16, 64), 0, ctx->stream, r);
        MMM_LAUNCH_CHECK(ctx);
        if (!r.p2p && (rc = mmm_allreduce_sum(ctx, m->stats[t & 1].p, (size_t)VK + 1))) return rc;
        if (m->ilda) {
            const int c = t % 3;
            hipLaunchKernelGGL(k_ilda_mstep, dim3(m->K + 1), dim3(64 * m->ids.I), 0, ctx->stream, m->ids, 0, m->stats[t & 1].p, m->ilam[c].p, m->iEln[c].p, m->ibeta[c].p,
                               m->Elnbeta[c].p, m->expElnbeta[c].p, m->beta[c].p, (const int*)&m->ctl.p->stop, 0, r, 1);
        } else if (m->wide)
            hipLaunchKernelGGL(k_lda_mstep_wide, dim3(m->K + 1), dim3(512), 0, ctx->stream, r, m->V, m->eta, m->ring(m->lambda), m->ring(m->Elnbeta),
                               m->ring(m->expElnbeta), m->ring(m->beta));
        else
            hipLaunchKernelGGL(r.p2p ? k_lda_mstep<true> : k_lda_mstep<false>, dim3(m->K + 1), dim3(128), 0, ctx->stream, r, m->V, m->eta, m->ring(m->lambda),
                               m->ring(m->Elnbeta), m->ring(m->expElnbeta), m->ring(m->beta));
        MMM_LAUNCH_CHECK(ctx);
        // host mirror, assuming no early stop (sync_ctl corrects it)
        if (do_ll) m->n_hist++;
        m->t = t;
        m->ll_pending = true;
    }
    if (n_iter > 0) {
        m->inflight = true; m->lag_ll = true; m->phi_table_beta = false;
        m->phi_valid = false; m->phi_from_prev = true; m->gnext_valid = true; m->theta_valid = false;
    }
    return MMM_OK;
}

// n_iter frozen-topic passes: update_γ!, (unsmoothed_)update_ϕ!, update_θ!, ll (LDA.jl:241-247 / :274-279).  The topic state
// must be replicated in the three ring slots (mmm_lda_infer does that once).
int frozen_passes(mmm_lda* m, int n_iter, int unsmoothed, double tol, int conv_base)
{
    mmm_ctx* ctx = m->ctx;
    int rc;
    if ((rc = ensure_hist(m, n_iter))) return rc;
    if (!m->gnext_valid) {
        if ((rc = materialise_phi(m))) return rc;
        hipLaunchKernelGGL(m->K > 64 ? k_lda_gamma_from_phi_big : k_lda_gamma_from_phi, dim3(m->grid_s), dim3(kBlock), 0, ctx->stream, m->dev(), m->phi.p, m->gamma[(m->t + 1) % 3].p, (double*)nullptr);
        MMM_LAUNCH_CHECK(ctx);
        m->gnext_valid = true;
    }
    const int VK = m->V * m->K;
    const bool comm = mmm_comm_active(ctx);
    for (int it = 0; it < n_iter; ++it) {
        const int t = m->t + 1;
        // theta_t = gamma_t / sum gamma_t feeds the ll of this pass: the "previous gamma" slot of the kernel is gamma_t itself
        Ring g = m->ring(m->gamma);
        g.s[(t + 2) % 3] = g.s[t % 3];
        EstepArgs a{m->dev(), m->ctl.p, g, m->ring(m->Elntheta), unsmoothed ? m->ring(m->beta) : m->ring(m->expElnbeta), m->ring(m->beta),
                    m->partial.p, m->llpart.p, 1, t, m->V};
        { ProfSpan span(ctx); rc = launch_estep(m, a); }
        if (rc) return rc;
        ReduceArgs r{m->partial.p, m->llpart.p, m->grid_e, VK, m->stats[t & 1].p, m->ctl.p, t, m->Nglobal, tol, m->ll_hist.p, 1, conv_base, 1};
        if (!comm) hipLaunchKernelGGL(k_lda_infer_tail, dim3(1), dim3(64), 0, ctx->stream, r, 3);
        else {
            hipLaunchKernelGGL(k_lda_infer_tail, dim3(1), dim3(64), 0, ctx->stream, r, 1);
            MMM_LAUNCH_CHECK(ctx);
            if ((rc = mmm_allreduce_sum(ctx, m->stats[t & 1].p + VK, 1))) return rc;
            hipLaunchKernelGGL(k_lda_infer_tail, dim3(1), dim3(64), 0, ctx->stream, r, 2);
        }
        MMM_LAUNCH_CHECK(ctx);
        m->n_hist++;
        m->t = t;
    }
    if (n_iter > 0) {
        m->inflight = true; m->lag_ll = false; m->ll_pending = false; m->phi_table_beta = unsmoothed != 0;
        m->phi_valid = false; m->phi_from_prev = true; m->gnext_valid = true; m->theta_valid = false;
    }
    return MMM_OK;
}

int prepare_call(mmm_lda* m)
{
    if (int rc = mmm_ctx_usable(m->ctx, "LDA call")) return rc;
    MMM_HIP(m->ctx, hipSetDevice(m->ctx->device));
    m->aexp_for = -1;      // whatever this call does to the state, its first pass forms its own prologue
    return sync_ctl(m);
}

} // namespace

// Chunks of passes, pipelined: the control block is snapshotted in-stream (pinned memory + event) after every chunk, and the host
// examines chunk i's snapshot only after chunk i+1 has been enqueued -- no bubble on the GPU between chunks; the cost is at most
// one chunk of no-op launches after the device-side criterion has fired.  *enq = passes enqueued.
template <class Enqueue>
static int run_chunks_pipelined(mmm_lda* m, int maxiter, int* enq_out, Enqueue enqueue)
{
    mmm_ctx* ctx = m->ctx;
    static_assert(sizeof(LdaCtl) <= 64, "control block larger than a pinned slot");
    if (!ctx->pin_ctl) MMM_HIP(ctx, hipHostMalloc(&ctx->pin_ctl, 128, hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) if (!ctx->pin_ev[i]) MMM_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
    int enq = 0, slot = 0, rc;
    bool have_prev = false;
    while (enq < maxiter) {
        const int chunk = std::min(maxiter - enq, enq == 0 ? 12 : 8);
        if ((rc = enqueue(chunk))) return rc;
        enq += chunk;
        LdaCtl* snap = (LdaCtl*)((char*)ctx->pin_ctl + 64 * slot);
        MMM_HIP(ctx, hipMemcpyAsync(snap, m->ctl.p, sizeof(LdaCtl), hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipEventRecord(ctx->pin_ev[slot], ctx->stream));
        if (have_prev) {
            MMM_HIP(ctx, hipEventSynchronize(ctx->pin_ev[slot ^ 1]));
            const LdaCtl* prev = (const LdaCtl*)((const char*)ctx->pin_ctl + 64 * (slot ^ 1));
            if (prev->stop || prev->wait_timeout) break;
        }
        have_prev = true; slot ^= 1;
    }
    *enq_out = enq;
    return MMM_OK;
}

#ifdef MMM_DIAG_STAMPS
extern "C" int mmm_diag_lda_stamps(unsigned long long out[16])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lda_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
extern "C" int mmm_diag_ll_times(unsigned long long out[1024])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ll_times), sizeof(unsigned long long) * 1024) == hipSuccess ? 0 : -2;
}
extern "C" int mmm_diag_red_stamps(unsigned long long out[32])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_red_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
}
#endif

extern "C" {

constexpr int kMinWavesSingle = 4;      // single-step build: fewest waves per block the library picks by itself (shard sizes: profiles/r05_lda_small_shards.txt)

static int lda_create_impl(mmm_ctx* ctx, int D, int V, int K, double alpha, double eta, const int64_t* doc_ptr, const int32_t* term,
                           const int32_t* count, const double* lambda0, int I, const int* J, const double* eta_i, const int32_t* features,
                           mmm_lda** out)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, out && doc_ptr && lambda0, "mmm_lda_create: NULL argument");
    const bool ilda = I > 0;
    int SJ = 0;
    if (ilda) {
        MMM_CHECK(ctx, J && eta_i && features, "mmm_ilda_create: NULL argument");
        if (I > kIldaMaxI) return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_ilda_create: I=%d features (max %d)", I, kIldaMaxI);
        for (int i = 0; i < I; ++i) { MMM_CHECK(ctx, J[i] >= 1, "mmm_ilda_create: J[%d] < 1", i); SJ += J[i]; }
        if (SJ > kIldaMaxSJ) return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_ilda_create: sum(J)=%d (max %d)", SJ, kIldaMaxSJ);
        for (int i = 0; i < I; ++i) for (int v = 0; v < V; ++v)
            MMM_CHECK(ctx, features[(size_t)i * V + v] >= 0 && features[(size_t)i * V + v] < J[i], "mmm_ilda_create: feature value out of range (i=%d v=%d)", i, v);
    }
    MMM_CHECK(ctx, D >= 0 && V >= 1 && K >= 1, "mmm_lda_create: bad sizes D=%d V=%d K=%d", D, V, K);
    *out = nullptr;
    const int KP = pick_kp(K);
    if (KP < 0) return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_lda_create: K=%d not supported (max 256: an LDS column block of 512 K bytes per wave)", K);
    const int64_t nnz = doc_ptr[D];
    MMM_CHECK(ctx, doc_ptr[0] == 0 && nnz >= 0 && (nnz == 0 || (term && count)), "mmm_lda_create: bad CSR");
    std::vector<int2> tc((size_t)nnz);
    for (int d = 0; d < D; ++d) MMM_CHECK(ctx, doc_ptr[d + 1] >= doc_ptr[d], "mmm_lda_create: doc_ptr not monotone at %d", d);
    for (int64_t e = 0; e < nnz; ++e) {
        MMM_CHECK(ctx, term[e] >= 0 && term[e] < V && count[e] >= 0, "mmm_lda_create: entry %lld out of range (term %d, count %d)", (long long)e, term[e], count[e]);
        tc[(size_t)e] = make_int2(term[e], count[e]);
    }
    const int L = (K <= 15) ? 16 : (K <= 31 ? 32 : 64);
    const int G = MMM_WAVE / L;
    const int ncu = mmm_geo_cus(ctx);      // the CU count the geometry (and so the association of the cross-document sums) is derived from
    // waves per block of the fused kernel: as many as fit 160 KiB of LDS next to the two tables, at most 8
    const size_t tabB = (size_t)KP * V * sizeof(double);
    // Small corpora (every document resident at once): 6-wave blocks, two per CU, one step per wave with the <= 168-VGPR
    // single-step build (3 waves per SIMD).  Larger corpora: 8-wave blocks, one per CU, grid-stride steps (2 waves per SIMD).
    // dense-row E-step (k_lda_estep_dense): a dense corpus (at least half of the D x V entries present, no term listed twice in a
    // document) of at least 192 documents per CU, topics and vocabulary within the register budget of a lane (SL KP <= 64
    // doubles).  MMM_LDA_DENSE=1 takes it for any corpus that has the shape (tests), 0 never.
    // Rows of counts (drows): a dense corpus over <= 128 terms is also kept as rows of 16 SL int32 counts -- 4 bytes per term slot against
    // 8 per nonzero -- which the single-step E-step build (96-term vocabularies) and the ll blocks read instead of the padded (term,count)
    // rows: BASELINE config 2 21.2 -> 19.9 us per iteration on the same box.  MMM_LDA_DROWS=0 keeps the (term,count) rows (A/B).
    bool dense = false, drows = false;
    const int SL = dense_slots(V);
    {
        const int build = ctx->tune.lda_build;
        const int dmode = build == MMM_BUILD_DENSE ? 1 : (build == MMM_BUILD_AUTO ? -1 : 0);
        const bool drows_env = !mmm_off(ctx->tune, MMM_OFF_LDA_COUNT_ROWS);
        const bool rshape = SL > 0 && L == 16 && D > 0 && build != MMM_BUILD_WIDE;          // rows of counts make sense
        const bool shape = rshape && KP >= 4 && KP * SL <= 64;                               // ... and the dense-row E-step build exists
        const bool dense_enough = 2 * nnz >= (int64_t)D * V;
        bool dup = false;
        if (rshape && ((shape && dmode != 0) || (drows_env && dense_enough))) {
            std::vector<int> seen((size_t)V, -1);
            for (int d = 0; d < D && !dup; ++d)
                for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) { if (seen[(size_t)term[e]] == d) { dup = true; break; } seen[(size_t)term[e]] = d; }
        }
        // measured on MI355X (K = 10, V = 96; E-step launch, dense rows vs the CSR sweep): 10k documents 12.4 vs 10.9 us (the single-step build),
        // 15k 12.8 vs 14.1, 20k 16.7 vs 19.3, 40k 20.2 vs 27.7, 640k 190 vs 380 -- every corpus beyond the single-step build's reach
        // (profiles/experiments/r03_sweeps/dense_crossover.sh; round 2's build only paid from 49k documents: its epilogue cost 19 us)
        const bool big = D > 12 * G * ncu;
        const bool off32 = (int64_t)D * K * 8 < ((int64_t)1 << 32) && (int64_t)D * 16 * (SL + 1) * 4 < ((int64_t)1 << 32);   // the build's 32-bit byte offsets
        dense = shape && !dup && off32 && dmode != 0 && (dmode > 0 || (big && dense_enough));
        drows = drows_env && rshape && !dup && dense_enough;
    }
    const bool small = !dense && (V <= 96) && KP <= 12 && ((D + 12 * G - 1) / (12 * G) <= ncu) && ctx->tune.grid_blocks == 0;
    // single-step build: just enough waves per block to cover the corpus with one block per CU (fewer co-resident waves
    // per SIMD = shorter step); mmm_tuning_opts.waves_per_block pins the count (more blocks than CUs are fine: two blocks share a CU)
    const int swaves = ctx->tune.waves_per_block > 0 ? std::min(12, ctx->tune.waves_per_block) : std::max(kMinWavesSingle, std::min(12, (D + G * ncu - 1) / (G * ncu)));
    int waves = small ? swaves : 8;
    auto lds_for = [&](int w) { return tabB * (2 + w) + (size_t)2 * w * G * KP * sizeof(double); };
    while (waves > 1 && lds_for(waves) > (small ? 150 : 80) * 1024) --waves;
    // tables + one slab beyond LDS: the wide path (k_lda_estep_wide).  It also takes K > 24: the LDS kernel's 32-topic build
    // spills (10k x 96-term documents, K = 32: 264 us per iteration against 189).  lda_build = MMM_BUILD_WIDE forces it, MMM_BUILD_SPARSE /
    // _DENSE avoid it where the LDS kernels can run (tests, A/B).
    const bool wide = lds_for(waves) > 160 * 1024 || KP > 32 || (ctx->tune.lda_build == MMM_BUILD_WIDE) || (ctx->tune.lda_build == MMM_BUILD_AUTO && KP >= 32);

    MMM_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mmm_lda> guard(new mmm_lda());      // every early return below (MMM_HIP, ...) destroys the model and its buffers
    mmm_lda* m = guard.get();
    m->tune = ctx->tune;
    m->ctx = ctx; m->D = D; m->V = V; m->K = K; m->KP = KP; m->L = L; m->nnz = nnz; m->alpha = alpha; m->eta = eta;
    m->waves_e = waves; m->lds_e = wide ? 0 : lds_for(waves); m->lds_tab = tabB; m->wide = wide;
    m->dense = dense && !wide; m->SL = SL; m->drows = (dense || drows) && !wide;
    if (!wide && ctx->tune.waves_per_block > 0) { const int w = ctx->tune.waves_per_block; if (w >= 1 && w <= (small ? kMaxWavesE : 8) && lds_for(w) <= 160 * 1024) { m->waves_e = w; m->lds_e = lds_for(w); } }
    const size_t VK = (size_t)V * K, KD = (size_t)K * D;
    const int docs_per_block = m->waves_e * G;
    const int blocks_per_cu = wide ? 8 : std::max(1, std::min<int>((small ? 12 : 8) / m->waves_e, (int)((160 * 1024) / m->lds_e)));
    m->single_step = !wide && small && (int64_t)m->waves_e * G * ncu * blocks_per_cu >= D;      // the grid covers every document at once
    m->grid_e = std::max(1, std::min((D + docs_per_block - 1) / docs_per_block, ncu * blocks_per_cu));
    if (wide) m->grid_e = std::max(1, std::min((D + kWavesPerBlock - 1) / kWavesPerBlock, ncu * blocks_per_cu));     // wave per document
    if (ctx->tune.grid_blocks > 0) m->grid_e = ctx->tune.grid_blocks;
    if ((int64_t)m->grid_e * docs_per_block < D) m->single_step = false;
    if (m->dense)      // [16 SL][KP] table | [waves][K][V] slabs | [waves][G][KP] a_k | [waves][64][KP] gamma sums
        m->lds_d = sizeof(double) * ((size_t)16 * SL * KP + (size_t)m->waves_e * 16 * SL * KP + (size_t)m->waves_e * G * KP + (size_t)m->waves_e * MMM_WAVE * KP);
    if (m->dense && m->lds_d > 160 * 1024) { m->dense = false; m->drows = drows && !wide; }      // no dense-row build: rows only if the corpus is dense enough
    m->grid_s = std::max(1, std::min((D + kWavesPerBlock - 1) / kWavesPerBlock, ncu * 4));
    const int grid_max = std::max(m->grid_e, m->grid_s);
#define A(buf, n) do { hipError_t e_ = m->buf.alloc(n); if (e_ != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(" #buf "): %s", hipGetErrorString(e_)); return rc; } } while (0)
    A(doc_ptr, (size_t)D + 1); A(tc, (size_t)nnz);
    for (int i = 0; i < 3; ++i) { A(lambda[i], VK); A(Elnbeta[i], VK); A(expElnbeta[i], VK); A(beta[i], VK); A(gamma[i], KD); A(Elntheta[i], KD); }
    A(theta, KD); A(phi, (size_t)K * nnz);
    const size_t VKp = (size_t)((V + 15) & ~15) * K;       // rows padded to 16 (k_lda_reduce_ll_mstep)
    A(partial, wide ? 1 : (size_t)m->grid_e * VKp); A(stats[0], VKp + 16); A(stats[1], VKp + 16); A(scratch, VK + 16); A(llpart, (size_t)grid_max); A(llpart2, 1024); A(elbopart, (size_t)m->grid_s * 5 + 8);
    A(ctl, 1); A(cells, 2 * 1024);
    if (m->single_step && !ilda) A(aexp_next, KD);
    if (ilda) {
        A(fcells, (size_t)2 * 512 * 16);
        A(features, (size_t)I * V);
        for (int i = 0; i < 3; ++i) { A(ilam[i], (size_t)SJ * K); A(iEln[i], (size_t)SJ * K); A(ibeta[i], (size_t)SJ * K); }
    }
#undef A
    hipStream_t st = ctx->stream;
    MMM_HIP(ctx, hipMemcpyAsync(m->doc_ptr.p, doc_ptr, sizeof(int64_t) * (D + 1), hipMemcpyHostToDevice, st));
    if (nnz) MMM_HIP(ctx, hipMemcpyAsync(m->tc.p, tc.data(), sizeof(int2) * nnz, hipMemcpyHostToDevice, st));
    std::vector<int64_t> tptr;
    std::vector<int2> tpost;
    if (wide) {     // postings by term, documents ascending within a term (counting sort): the summation order of k_lda_stats_terms
        tptr.assign((size_t)V + 1, 0);
        for (int64_t e = 0; e < nnz; ++e) tptr[(size_t)term[e] + 1]++;
        for (int v = 0; v < V; ++v) tptr[(size_t)v + 1] += tptr[(size_t)v];
        tpost.resize((size_t)nnz);
        std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
        for (int d = 0; d < D; ++d)
            for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) tpost[(size_t)fill[(size_t)term[e]]++] = make_int2(d, count[e]);
        hipError_t e1 = m->term_ptr.alloc((size_t)V + 1), e2 = m->tpost.alloc((size_t)nnz), e3 = m->aexp.alloc((size_t)KP * D), e4 = m->tabT.alloc((size_t)2 * V * KP);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(postings): out of memory"); return rc; }
        MMM_HIP(ctx, hipMemcpyAsync(m->term_ptr.p, tptr.data(), sizeof(int64_t) * ((size_t)V + 1), hipMemcpyHostToDevice, st));
        if (nnz) MMM_HIP(ctx, hipMemcpyAsync(m->tpost.p, tpost.data(), sizeof(int2) * (size_t)nnz, hipMemcpyHostToDevice, st));
        // waves per term block: segments of >= 128 postings on average, at most 8
        const int64_t avg = nnz / std::max(1, V);
        m->stats_waves = 1;
        while (m->stats_waves < 8 && avg / (m->stats_waves * 2) >= 128) m->stats_waves *= 2;
    }
    const bool rows16_env = !mmm_off(ctx->tune, MMM_OFF_LDA_ROWS16);
    int maxcount = 0;
    for (int64_t e = 0; e < nnz; ++e) maxcount = std::max(maxcount, count[e]);
    if (m->drows && rows16_env && maxcount < 65536) {
        m->SLs = (SL + 1) & ~1;          // lane-major rows (LdaDev::dense): an even number of 16-bit slots per lane
        const int Vp = 16 * m->SLs;
        std::vector<unsigned short> rows((size_t)D * Vp + 8, 0);       // (+ 16 bytes: the ll blocks read 16 bytes from a lane's first slot)
        for (int d = 0; d < D; ++d)
            for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) rows[(size_t)d * Vp + (term[e] & 15) * m->SLs + (term[e] >> 4)] = (unsigned short)count[e];
        hipError_t e_ = m->cnt16.alloc(rows.size());
        if (e_ != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(cnt16): %s", hipGetErrorString(e_)); return rc; }
        MMM_HIP(ctx, hipMemcpyAsync(m->cnt16.p, rows.data(), sizeof(unsigned short) * rows.size(), hipMemcpyHostToDevice, st));
        MMM_HIP(ctx, hipStreamSynchronize(st));
    } else if (m->drows) {
        m->SLs = SL;
        const int Vp = 16 * SL;
        std::vector<int> rows((size_t)D * Vp, 0);
        for (int d = 0; d < D; ++d)
            for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) rows[(size_t)d * Vp + (term[e] & 15) * SL + (term[e] >> 4)] = count[e];
        hipError_t e_ = m->cnt_dense.alloc(rows.size());
        if (e_ != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(cnt_dense): %s", hipGetErrorString(e_)); return rc; }
        MMM_HIP(ctx, hipMemcpyAsync(m->cnt_dense.p, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice, st));
        MMM_HIP(ctx, hipStreamSynchronize(st));
    }
    std::vector<int2> ell;
    {   // padded rows for the ll blocks (V <= 128 slots, no duplicate terms: then a document always fits its row)
        int64_t maxW = 0;
        for (int d = 0; d < D; ++d) maxW = std::max<int64_t>(maxW, doc_ptr[d + 1] - doc_ptr[d]);
        // (the ll blocks take their lanes per document from KP, the rows of counts serve 16-lane groups only: K = 13..15 needs the padded rows too)
        if (V <= 128 && maxW <= V && D > 0 && !wide && (!m->drows || m->KP > 15)) {
            ell.assign((size_t)D * V, make_int2(-1, 0));
            for (int d = 0; d < D; ++d)
                for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) ell[(size_t)d * V + (e - doc_ptr[d])] = tc[(size_t)e];
            hipError_t e_ = m->tc_ell.alloc(ell.size());
            if (e_ != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(tc_ell): %s", hipGetErrorString(e_)); return rc; }
            MMM_HIP(ctx, hipMemcpyAsync(m->tc_ell.p, ell.data(), sizeof(int2) * ell.size(), hipMemcpyHostToDevice, st));
        }
    }
    if (!ilda) MMM_HIP(ctx, hipMemcpyAsync(m->lambda[0].p, lambda0, sizeof(double) * VK, hipMemcpyHostToDevice, st));
    else {
        m->ilda = true;
        IldaDesc& ds = m->ids;
        ds.I = I; ds.V = V; ds.K = K; ds.SJ = SJ; ds.joff[0] = 0;
        for (int i = 0; i < I; ++i) { ds.J[i] = J[i]; ds.joff[i + 1] = ds.joff[i] + J[i]; ds.eta[i] = eta_i[i]; }
        ds.features = m->features.p;
        MMM_HIP(ctx, hipMemcpyAsync(m->features.p, features, sizeof(int) * (size_t)I * V, hipMemcpyHostToDevice, st));
        MMM_HIP(ctx, hipMemcpyAsync(m->ilam[0].p, lambda0, sizeof(double) * (size_t)SJ * K, hipMemcpyHostToDevice, st));
        for (int i = 0; i < 3; ++i) MMM_HIP(ctx, hipMemsetAsync(m->lambda[i].p, 0, sizeof(double) * VK, st));      // unused for ILDA
    }
    MMM_HIP(ctx, hipMemsetAsync(m->ctl.p, 0, sizeof(LdaCtl), st));
    MMM_HIP(ctx, hipMemsetAsync(m->cells.p, 0, sizeof(unsigned long long) * 2 * 1024, st));
    if (ilda) MMM_HIP(ctx, hipMemsetAsync(m->fcells.p, 0, sizeof(unsigned long long) * 2 * 512 * 16, st));
    if (!wide) MMM_HIP(ctx, hipMemsetAsync(m->partial.p, 0, sizeof(double) * (size_t)m->grid_e * VKp, st));      // pad entries are never written
    if (KD) MMM_HIP(ctx, hipMemsetAsync(m->theta.p, 0, sizeof(double) * KD, st));
    MMM_HIP(ctx, hipStreamSynchronize(st));   // tc (host vector) must outlive the copy
    // constructor state (LDA.jl:36-49): Elnbeta from lambda0; gamma = 1 -> Elntheta; phi = 1/K
    if (!ilda) hipLaunchKernelGGL(k_lda_topic, dim3(K), dim3(256), 0, st, V, eta, (const double*)nullptr, m->lambda[0].p, m->Elnbeta[0].p, m->expElnbeta[0].p, m->beta[0].p, 0);
    else hipLaunchKernelGGL(k_ilda_mstep, dim3(K), dim3(64 * m->ids.I), 0, st, m->ids, 1, (const double*)nullptr, m->ilam[0].p, m->iEln[0].p, m->ibeta[0].p,
                            m->Elnbeta[0].p, m->expElnbeta[0].p, m->beta[0].p, (const int*)nullptr, 0, ReduceArgs{}, 0);      // ILDA.jl:36-40 (+ tables)
    if (KD) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((KD + 255) / 256)), dim3(256), 0, st, m->gamma[0].p, KD, 1.0);
        hipLaunchKernelGGL(m->K > 64 ? k_lda_Elntheta_big : k_lda_Elntheta, dim3(m->grid_s), dim3(kBlock), 0, st, m->dev(), m->gamma[0].p, m->Elntheta[0].p);
    }
    if (nnz) hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)K * nnz + 255) / 256)), dim3(256), 0, st, m->phi.p, (size_t)K * nnz, 1.0 / K);
    MMM_HIP(ctx, hipMemsetAsync(m->scratch.p, 0, sizeof(double) * (VK + 16), st));
    double* ncount = m->scratch.p + VK + 1;
    if (nnz) hipLaunchKernelGGL(k_doc_counts, dim3(64), dim3(256), 0, st, m->dev(), ncount);
    MMM_LAUNCH_CHECK(ctx);
    // global N and D (sum over ranks)
    double hd[2] = {0.0, (double)D};
    MMM_HIP(ctx, hipMemcpyAsync(&hd[0], ncount, sizeof(double), hipMemcpyDeviceToHost, st));
    MMM_HIP(ctx, hipStreamSynchronize(st));
    if (mmm_comm_active(ctx)) {
        MMM_HIP(ctx, hipMemcpyAsync(ncount, hd, sizeof hd, hipMemcpyHostToDevice, st));
        int rc = mmm_allreduce_sum(ctx, ncount, 2);
        if (rc) { return rc; }
        MMM_HIP(ctx, hipMemcpyAsync(hd, ncount, sizeof hd, hipMemcpyDeviceToHost, st));
        MMM_HIP(ctx, hipStreamSynchronize(st));
    }
    m->Nglobal = hd[0]; m->Dglobal = hd[1];
    m->phi_valid = true; m->phi_from_prev = false; m->gnext_valid = false; m->ll_pending = false; m->theta_valid = false;
    *out = guard.release();
    mmm_ctx_model_created(ctx);
    return MMM_OK;
}

int mmm_lda_create(mmm_ctx* ctx, int D, int V, int K, double alpha, double eta, const int64_t* doc_ptr, const int32_t* term,
                   const int32_t* count, const double* lambda0, mmm_lda** out)
{
    return lda_create_impl(ctx, D, V, K, alpha, eta, doc_ptr, term, count, lambda0, 0, nullptr, nullptr, nullptr, out);
}

int mmm_ilda_create(mmm_ctx* ctx, int D, int V, int K, double alpha, int I, const int* J, const double* eta, const int32_t* features,
                    const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const double* lambda0, mmm_lda** out)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, I >= 1, "mmm_ilda_create: I < 1");
    return lda_create_impl(ctx, D, V, K, alpha, eta ? eta[0] : 0.0, doc_ptr, term, count, lambda0, I, J, eta, features, out);
}

int mmm_lda_destroy(mmm_lda* m)
{
    if (!m) return MMM_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    mmm_ctx* ctx = m->ctx;
    delete m;
    mmm_ctx_model_destroyed(ctx);
    return MMM_OK;
}

static int lda_field(mmm_lda* m, int field, double** p, size_t* n)
{
    const size_t VK = (size_t)m->V * m->K, KD = (size_t)m->K * m->D;
    const int c = m->cur();
    switch (field) {
        case MMM_LDA_LAMBDA: *p = m->lambda[c].p; *n = VK; break;
        case MMM_LDA_ELNBETA: *p = m->Elnbeta[c].p; *n = VK; break;
        case MMM_LDA_BETA: *p = m->beta[c].p; *n = VK; break;
        case MMM_LDA_GAMMA: *p = m->gamma[c].p; *n = KD; break;
        case MMM_LDA_ELNTHETA: *p = m->Elntheta[c].p; *n = KD; break;
        case MMM_LDA_THETA: *p = m->theta.p; *n = KD; break;
        case MMM_LDA_PHI: *p = m->phi.p; *n = (size_t)m->K * m->nnz; break;
        case MMM_ILDA_LAMBDA: case MMM_ILDA_ELNBETA: case MMM_ILDA_BETA:
            if (!m->ilda) return mmm_fail(m->ctx, MMM_ERR_ARG, "field %d exists for ILDA handles only", field);
            *p = field == MMM_ILDA_LAMBDA ? m->ilam[c].p : (field == MMM_ILDA_ELNBETA ? m->iEln[c].p : m->ibeta[c].p);
            *n = (size_t)m->ids.SJ * m->K; break;
        default: return mmm_fail(m->ctx, MMM_ERR_ARG, "unknown LDA field %d", field);
    }
    return MMM_OK;
}

int mmm_lda_get(mmm_lda* m, int field, double* host, size_t n)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc) return rc;
    double* p; size_t cnt;
    if ((rc = lda_field(m, field, &p, &cnt))) return rc;
    MMM_CHECK(ctx, host && n == cnt, "mmm_lda_get(field %d): expected %zu doubles, got %zu", field, cnt, n);
    if (field == MMM_LDA_PHI && (rc = materialise_phi(m))) return rc;
    if (field == MMM_LDA_THETA && !m->theta_valid && m->t > 0) {       // fit! leaves theta = gamma/sum (LDA.jl:207)
        if ((rc = launch_loglik(m, m->gamma[m->cur()].p, nullptr, m->theta.p, 0))) return rc;
        m->theta_valid = true;
    }
    if (n) MMM_HIP(ctx, hipMemcpyAsync(host, p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_lda_set(mmm_lda* m, int field, const double* host, size_t n)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc) return rc;
    double* p; size_t cnt;
    if ((rc = lda_field(m, field, &p, &cnt))) return rc;
    MMM_CHECK(ctx, host && n == cnt, "mmm_lda_set(field %d): expected %zu doubles, got %zu", field, cnt, n);
    if ((rc = materialise_phi(m))) return rc;     // make the implicit phi explicit before state is overwritten
    if ((rc = flush_ll(m, nullptr))) return rc;
    m->gnext_valid = false; m->phi_from_prev = false;
    if (field == MMM_LDA_THETA) m->theta_valid = true;
    if (field == MMM_LDA_GAMMA) m->theta_valid = false;
    if (n) MMM_HIP(ctx, hipMemcpyAsync(p, host, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (field == MMM_LDA_ELNBETA && n) {
        hipLaunchKernelGGL(k_exp_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, m->Elnbeta[m->cur()].p, m->expElnbeta[m->cur()].p);
        MMM_LAUNCH_CHECK(ctx);
    }
    if (field == MMM_ILDA_ELNBETA || field == MMM_ILDA_BETA) {      // effective tables follow the uploaded factors
        const int c = m->cur();
        hipLaunchKernelGGL(k_ilda_mstep, dim3(m->K), dim3(64 * m->ids.I), 0, ctx->stream, m->ids, 2, (const double*)nullptr, m->ilam[c].p, m->iEln[c].p, m->ibeta[c].p,
                           m->Elnbeta[c].p, m->expElnbeta[c].p, m->beta[c].p, (const int*)nullptr, field == MMM_ILDA_BETA ? 1 : 0, ReduceArgs{}, 0);
        MMM_LAUNCH_CHECK(ctx);
    }
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_lda_set_hyper(mmm_lda* m, double alpha, const double* eta, int n_eta)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc) return rc;
    MMM_CHECK(ctx, eta && n_eta == (m->ilda ? m->ids.I : 1), "mmm_lda_set_hyper: expected %d eta value(s), got %d", m->ilda ? m->ids.I : 1, n_eta);
    bool same = alpha == m->alpha && eta[0] == m->eta;
    if (m->ilda) for (int i = 0; i < n_eta; ++i) same = same && eta[i] == m->ids.eta[i];
    if (same) return MMM_OK;
    // gamma_{t+1}, which the last fused pass has formed with the old alpha, is dropped: the next pass starts from phi (LDA.jl:82-90)
    if ((rc = materialise_phi(m)) || (rc = flush_ll(m, nullptr))) return rc;
    m->gnext_valid = false; m->phi_from_prev = false;
    m->alpha = alpha; m->eta = eta[0];
    if (m->ilda) for (int i = 0; i < n_eta; ++i) m->ids.eta[i] = eta[i];
    return MMM_OK;
}

int mmm_lda_update_gamma(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prepare_call(m);
    if (rc || (rc = materialise_phi(m)) || (rc = flush_ll(m, nullptr))) return rc;
    const int c = m->cur();
    hipLaunchKernelGGL(m->K > 64 ? k_lda_gamma_from_phi_big : k_lda_gamma_from_phi, dim3(m->grid_s), dim3(kBlock), 0, m->ctx->stream, m->dev(), m->phi.p, m->gamma[c].p, m->Elntheta[c].p);
    MMM_LAUNCH_CHECK(m->ctx);
    m->gnext_valid = false; m->phi_from_prev = false; m->theta_valid = false;
    return MMM_OK;
}

int mmm_lda_update_phi(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prepare_call(m);
    if (rc || (rc = flush_ll(m, nullptr))) return rc;
    const int c = m->cur();
    if ((rc = launch_phi(m, m->Elntheta[c].p, m->expElnbeta[c].p))) return rc;
    m->phi_valid = true; m->phi_from_prev = false; m->gnext_valid = false;
    return MMM_OK;
}

int mmm_lda_update_lambda(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc || (rc = materialise_phi(m)) || (rc = flush_ll(m, nullptr))) return rc;
    MMM_HIP(ctx, hipMemsetAsync(m->scratch.p, 0, sizeof(double) * (size_t)m->V * m->K, ctx->stream));
    if (m->nnz) hipLaunchKernelGGL(k_lda_lambda_from_phi, dim3((unsigned)((m->nnz + 255) / 256)), dim3(256), 0, ctx->stream, m->dev(), m->nnz, m->phi.p, m->scratch.p);
    MMM_LAUNCH_CHECK(ctx);
    m->gnext_valid = false; m->phi_from_prev = false;
    return run_topic_update(m, true);
}

int mmm_lda_update_Elntheta(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prepare_call(m);
    if (rc || (rc = materialise_phi(m)) || (rc = flush_ll(m, nullptr))) return rc;
    const int c = m->cur();
    if (m->D) hipLaunchKernelGGL(m->K > 64 ? k_lda_Elntheta_big : k_lda_Elntheta, dim3(m->grid_s), dim3(kBlock), 0, m->ctx->stream, m->dev(), m->gamma[c].p, m->Elntheta[c].p);
    MMM_LAUNCH_CHECK(m->ctx);
    m->gnext_valid = false; m->phi_from_prev = false;
    return MMM_OK;
}

int mmm_lda_update_Elnbeta(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prepare_call(m);
    if (rc || (rc = materialise_phi(m)) || (rc = flush_ll(m, nullptr))) return rc;
    m->gnext_valid = false; m->phi_from_prev = false;
    return run_topic_update(m, false);      // ILDA handles: the factors' Elnβ[i] and the effective table (ILDA.jl:96-101)
}

int mmm_lda_update_beta(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prepare_call(m);
    if (rc) return rc;
    const int c = m->cur();
    if (m->ilda)
        hipLaunchKernelGGL(k_ilda_mstep, dim3(m->K), dim3(64 * m->ids.I), 0, m->ctx->stream, m->ids, 1, (const double*)nullptr, m->ilam[c].p, m->iEln[c].p, m->ibeta[c].p,
                           m->Elnbeta[c].p, m->expElnbeta[c].p, m->beta[c].p, (const int*)nullptr, 1, ReduceArgs{}, 0);
    else
    hipLaunchKernelGGL(k_lda_topic, dim3(m->K), dim3(256), 0, m->ctx->stream, m->V, m->eta, (const double*)nullptr, m->lambda[c].p,
                       (double*)nullptr, (double*)nullptr, m->beta[c].p, 1);
    MMM_LAUNCH_CHECK(m->ctx);
    return MMM_OK;
}

int mmm_lda_update_theta(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    int rc = prepare_call(m);
    if (rc) return rc;
    if ((rc = launch_loglik(m, m->gamma[m->cur()].p, nullptr, m->theta.p, 0))) return rc;
    m->theta_valid = true;
    return MMM_OK;
}

int mmm_lda_loglik(mmm_lda* m, double* ll)
{
    if (!m || !ll) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc) return rc;
    // the reference evaluates with the stored theta and beta (LDA.jl:194-196); this entry point recomputes theta from
    // gamma first, which is what fit! has just done (LDA.jl:207) -- beta must be current (update_β! or a fused pass).
    double* dst = m->scratch.p + (size_t)m->V * m->K + 4;
    if ((rc = flush_ll(m, dst))) return rc;
    MMM_HIP(ctx, hipMemcpyAsync(ll, dst, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_lda_iterate(mmm_lda* m, int n_iter)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    if (int rc = mmm_ctx_usable(ctx, "mmm_lda_iterate")) return rc;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    MMM_CHECK(ctx, n_iter >= 0, "mmm_lda_iterate: n_iter < 0");
    m->aexp_for = -1;        // (as prepare_call: the first pass of a call forms its own prologue)
    if (m->stop_seen) {      // a previous fit! left the device stop flag set
        hipLaunchKernelGGL(k_ctl_clear_stop, dim3(1), dim3(1), 0, ctx->stream, m->ctl.p);
        MMM_LAUNCH_CHECK(ctx);
        m->stop_seen = false;
    }
    return fused_passes(m, n_iter, -1.0, 0);
}

int mmm_lda_geometry(const mmm_lda* m, int out[8])
{
    if (!m || !out) return MMM_ERR_ARG;
    out[0] = m->L; out[1] = m->grid_e; out[2] = m->waves_e; out[3] = m->single_step ? 1 : 0; out[4] = m->wide ? 1 : 0;
    out[5] = m->dense ? 1 : 0; out[6] = m->dense ? m->SL : 0; out[7] = m->KP;
    return MMM_OK;
}

int mmm_lda_row_bytes(const mmm_lda* m)
{
    if (!m || m->wide) return 0;
    if (m->drows) return (m->cnt16.p ? 2 : 4) * 16 * m->SLs;
    if (m->tc_ell.p && !m->dense) return 8 * m->V;
    return 0;
}

int mmm_lda_prologue_moved(const mmm_lda* m) { return (m && m->pro_used) ? 1 : 0; }

int mmm_lda_ll_history(mmm_lda* m, double* ll, int max_n, int* n)
{
    if (!m || !n) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc || (rc = flush_ll(m, nullptr))) return rc;
    const int cnt = std::min(max_n, m->n_hist);
    if (cnt > 0 && ll) MMM_HIP(ctx, hipMemcpyAsync(ll, m->ll_hist.p + (m->n_hist - cnt), sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // (the flush ends in the ranks' exchange: a peer that never came -- e.g. a caller that reads the history on one rank only -- must be an
    // error here, not a silently wrong last row)
    if ((rc = mmm_p2p_check(ctx))) return rc;
    *n = cnt;
    return MMM_OK;
}

int mmm_lda_events(mmm_lda* m, int64_t out[4])
{
    if (!m || !out) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc || (rc = flush_ll(m, nullptr))) return rc;
    std::vector<double> ll((size_t)std::max(m->n_hist, 0));
    if (!ll.empty()) MMM_HIP(ctx, hipMemcpyAsync(ll.data(), m->ll_hist.p, sizeof(double) * ll.size(), hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = mmm_p2p_check(ctx))) return rc;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (double v : ll) if (!std::isfinite(v)) ++out[2];
    return MMM_OK;
}

int mmm_lda_elbo(mmm_lda* m, double* elbo, double terms[7])
{
    if (!m || !elbo) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    int rc = prepare_call(m);
    if (rc || (rc = materialise_phi(m))) return rc;
    const int c = m->cur();
    double* acc = m->elbopart.p + (size_t)m->grid_s * 5;      // [0..4] doc sums, [5..6] topic sums
    hipLaunchKernelGGL(m->K > 64 ? k_lda_elbo_docs_big : k_lda_elbo_docs, dim3(m->grid_s), dim3(kBlock), 0, ctx->stream, m->dev(), m->phi.p, m->gamma[c].p, m->Elntheta[c].p, m->Elnbeta[c].p, m->elbopart.p);
    hipLaunchKernelGGL(k_sum_columns, dim3(5), dim3(64), 0, ctx->stream, m->elbopart.p, m->grid_s, 5, acc);
    if (m->ilda) hipLaunchKernelGGL(k_ilda_elbo_topics, dim3(1), dim3(256), 0, ctx->stream, m->ids, m->ilam[c].p, m->iEln[c].p, acc + 5);
    else hipLaunchKernelGGL(k_lda_elbo_topics, dim3(1), dim3(256), 0, ctx->stream, m->V, m->K, m->lambda[c].p, m->Elnbeta[c].p, acc + 5);
    MMM_LAUNCH_CHECK(ctx);
    if ((rc = mmm_allreduce_sum(ctx, acc, 5))) return rc;
    double h[7];
    MMM_HIP(ctx, hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = mmm_p2p_check(ctx))) return rc;
    const double K = m->K, V = m->V, al = m->alpha, et = m->eta;
    double t[7];
    t[0] = K * (lgamma(V * et) - V * lgamma(et)) + (et - 1.0) * h[5];            // LDA.jl:114-118
    if (m->ilda) {                                                               // ILDA.jl:132-141
        t[0] = h[5];
        for (int i = 0; i < m->ids.I; ++i) t[0] += K * (lgamma(m->ids.J[i] * m->ids.eta[i]) - m->ids.J[i] * lgamma(m->ids.eta[i]));
    }
    t[1] = m->Dglobal * (lgamma(K * al) - K * lgamma(al)) + (al - 1.0) * h[0];   // LDA.jl:120-124
    t[2] = h[1]; t[3] = h[2]; t[4] = h[6]; t[5] = h[4]; t[6] = h[3];
    if (terms) memcpy(terms, t, sizeof t);
    *elbo = t[0] + t[1] + t[2] + t[3] - t[4] - t[5] - t[6];
    return MMM_OK;
}

int mmm_lda_fit(mmm_lda* m, int maxiter, double tol, double* ll_hist, int* n_iter, int* converged, double* elbo)
{
    if (!m || !n_iter || !converged) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_CHECK(ctx, maxiter >= 1, "mmm_lda_fit: maxiter < 1");
    int rc = prepare_call(m);
    if (rc || (rc = flush_ll(m, nullptr))) return rc;
    hipLaunchKernelGGL(k_ctl_clear_stop, dim3(1), dim3(1), 0, ctx->stream, m->ctl.p);
    MMM_LAUNCH_CHECK(ctx);
    m->stop_seen = false;
    *converged = 0;
    const int base = m->n_hist, t0 = m->t;
    // The stopping rule (LDA.jl:215 + common.jl:53-56) is evaluated on the device in the M-step tail of pass i+1 for
    // pass i (lagged ll); later launches are no-ops once it fires.  The host only looks at the flag between chunks.
    int enq = 0;
    if ((rc = run_chunks_pipelined(m, maxiter, &enq, [&](int chunk) { return fused_passes(m, chunk, tol, base); }))) return rc;
    if ((rc = sync_ctl(m))) return rc;
    const bool stopped = (m->t - t0) < enq;       // the device discarded passes after the criterion fired
    if (stopped) *converged = 1;
    else {
        // maxiter passes ran; the ll of the last one is still pending and its convergence test is done here
        if ((rc = flush_ll(m, nullptr))) return rc;
    }
    const int n = m->n_hist - base;
    std::vector<double> ll((size_t)std::max(n, 1));
    if (n > 0) MMM_HIP(ctx, hipMemcpyAsync(ll.data(), m->ll_hist.p + base, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!stopped && n > 10 && fabs(ll[n - 2] - ll[n - 1]) / fabs(ll[n - 1]) < tol) *converged = 1;
    *n_iter = n;
    if (ll_hist) memcpy(ll_hist, ll.data(), sizeof(double) * n);
    if (elbo) return mmm_lda_elbo(m, elbo, nullptr);
    return MMM_OK;
}

int mmm_lda_infer(mmm_lda* m, int unsmoothed, int maxiter, double tol, double* ll_hist, int* n_iter, int* converged)
{
    if (!m || !n_iter || !converged) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_CHECK(ctx, maxiter >= 1, "mmm_lda_infer: maxiter < 1");
    int rc = prepare_call(m);
    if (rc || (rc = flush_ll(m, nullptr))) return rc;
    if (!m->gnext_valid && (rc = materialise_phi(m))) return rc;
    hipLaunchKernelGGL(k_ctl_clear_stop, dim3(1), dim3(1), 0, ctx->stream, m->ctl.p);
    MMM_LAUNCH_CHECK(ctx);
    m->stop_seen = false;
    // the topics do not change: every ring slot holds them, whichever slot a pass calls "current"
    const int c = m->cur();
    const size_t VKb = sizeof(double) * m->V * m->K;
    for (int s = 0; s < 3; ++s) {
        if (s == c) continue;
        MMM_HIP(ctx, hipMemcpyAsync(m->lambda[s].p, m->lambda[c].p, VKb, hipMemcpyDeviceToDevice, ctx->stream));
        MMM_HIP(ctx, hipMemcpyAsync(m->Elnbeta[s].p, m->Elnbeta[c].p, VKb, hipMemcpyDeviceToDevice, ctx->stream));
        MMM_HIP(ctx, hipMemcpyAsync(m->expElnbeta[s].p, m->expElnbeta[c].p, VKb, hipMemcpyDeviceToDevice, ctx->stream));
        MMM_HIP(ctx, hipMemcpyAsync(m->beta[s].p, m->beta[c].p, VKb, hipMemcpyDeviceToDevice, ctx->stream));
        if (m->ilda) {
            const size_t SJb = sizeof(double) * m->ids.SJ * m->K;
            MMM_HIP(ctx, hipMemcpyAsync(m->ilam[s].p, m->ilam[c].p, SJb, hipMemcpyDeviceToDevice, ctx->stream));
            MMM_HIP(ctx, hipMemcpyAsync(m->iEln[s].p, m->iEln[c].p, SJb, hipMemcpyDeviceToDevice, ctx->stream));
            MMM_HIP(ctx, hipMemcpyAsync(m->ibeta[s].p, m->ibeta[c].p, SJb, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    *converged = 0;
    const int base = m->n_hist;
    int enq = 0;
    if ((rc = run_chunks_pipelined(m, maxiter, &enq, [&](int chunk) { return frozen_passes(m, chunk, unsmoothed, tol, base); }))) return rc;
    if ((rc = sync_ctl(m))) return rc;
    if (m->stop_seen) *converged = 1;
    const int n = m->n_hist - base;
    *n_iter = n;
    if (ll_hist && n > 0) MMM_HIP(ctx, hipMemcpyAsync(ll_hist, m->ll_hist.p + base, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

} // extern "C"
