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
#endif

__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one chunk of L terms of a document group: phi_kw n_w into the accumulators and the wave's slab, and (LL) the
// log-likelihood numerator of the previous iteration.  __restrict__ tells the compiler that the slab atomics do not
// alias the table reads, so the reads of the following chunk can be issued ahead of them.
template <int KP, bool LL>
__device__ __forceinline__ void lda_chunk(const int2 tcv, const bool act, const int V, const double (&av)[KP], double (&acc)[KP],
                                          const double* __restrict__ sB, const double* __restrict__ sBeta,
                                          const double* __restrict__ myT, double* __restrict__ slab, double& ll_acc)
{
    const double n = (double)tcv.y;
    if (LL) {
        const double* bc = sBeta + tcv.x;
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(myT[k], bc[k * V], p0); p1 = fma(myT[k + 1], bc[(k + 1) * V], p1); }
        if (KP & 1) p0 = fma(myT[KP - 1], bc[(KP - 1) * V], p0);
        ll_acc = fma(n, dev_log_pos(p0 + p1), ll_acc);                    // inactive lanes: n = 0
        __builtin_amdgcn_sched_barrier(0);                                  // keep the two halves' live ranges apart
    }
    const double* bcol = sB + tcv.x;
    if (KP >= 20) {
        // many topics: the products are formed twice (a second LDS read of the column) instead of being kept -- 64 registers
        // less, which is the difference between this build fitting its 256 and spilling
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < KP; k += 2) { s0 = fma(av[k], bcol[k * V], s0); s1 = fma(av[k + 1], bcol[(k + 1) * V], s1); }
        const double r = act ? n * dev_rcp(s0 + s1) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        double* scol = slab + tcv.x;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const double x = av[k] * bcol[k * V] * r;
            acc[k] += x;
            if (act) unsafeAtomicAdd(&scol[k * V], x);
        }
        return;
    }
    double b[KP], s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < KP; ++k) b[k] = av[k] * bcol[k * V];
#pragma unroll
    for (int k = 0; k + 1 < KP; k += 2) { s0 += b[k]; s1 += b[k + 1]; }
    if (KP & 1) s0 += b[KP - 1];
    const double r = act ? n * dev_rcp(s0 + s1) : 0.0;
#pragma unroll
    for (int k = 0; k < KP; ++k) { b[k] *= r; acc[k] += b[k]; }        // phi_kw * n_w (padded topics: exact zeros)
    if (act) {
        double* scol = slab + tcv.x;
#pragma unroll
        for (int k = 0; k < KP; ++k) unsafeAtomicAdd(&scol[k * V], b[k]);
    }
}

// SINGLE: the grid covers every document with one step per wave (no step loop: 46 VGPRs less -> 3 waves per SIMD)
template <int KP, int L, bool LL, int VT, bool SINGLE>
__global__ __launch_bounds__(SINGLE ? kMaxWavesE * MMM_WAVE : 512, SINGLE ? 3 : 2) void k_lda_estep(EstepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int G = MMM_WAVE / L;                   // documents per wave step
    constexpr int PRE = (96 + L - 1) / L;             // chunks prefetched into registers (covers a 96-term document)
    MMM_STAMP(0);
    const int t = a.t;
    const int stop = a.ctl->stop;                     // consumed after the first prologue (its latency is hidden)
    const double* __restrict__ gam = a.gamma.s[t % 3];
    const double* __restrict__ gprev = a.gamma.s[(t + 2) % 3];
    double* __restrict__ gnext = a.gamma.s[(t + 1) % 3];
    double* __restrict__ Eln = a.Elntheta.s[t % 3];
    const double* __restrict__ eB = a.expElnbeta.s[(t + 2) % 3];
    const double* __restrict__ bprev = a.beta.s[(t + 2) % 3];

    const int K = a.c.K, D = a.c.D;
    const int V = VT ? VT : a.c.V;                    // VT != 0: row stride known at compile time (immediate LDS offsets)
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / L, l = lane % L;
    double* sB = smem;                                   // [KP][V] exp(Elnbeta_{t-1})
    double* sBeta = sB + (size_t)KP * V;                 // [KP][V] beta_{t-1}
    double* sSlab = sBeta + (size_t)KP * V;              // [NW][KP][V]
    double* sA = sSlab + (size_t)NW * KP * V;            // [NW][G][KP]
    double* sT = sA + (size_t)NW * G * KP;               // [NW][G][KP]
    double* slab = sSlab + (size_t)wid * KP * V;
    double* myA = sA + ((size_t)wid * G + g) * KP;
    double* myT = sT + ((size_t)wid * G + g) * KP;
    const int stride = gridDim.x * NW * G;
    int base = (blockIdx.x * NW + wid) * G;
    double ll_acc = 0.0;

    // ---- document loads of the first step are issued before the tables are staged (latency overlap) -------------
    int d = base + g;
    bool valid = d < D;
    // ext: this pass's prologue (digamma, exp: 2 us of this kernel's 10 at BASELINE config 2, all of it on every wave's dependent chain)
    // has run beside the previous pass's reduction, off the critical path; the same functions on the same lanes, hence the same bits
    const bool ext = SINGLE && !LL && a.aexp != nullptr;
    double gk = ext ? ((valid && l < K) ? a.aexp[(size_t)d * K + l] : 0.0) : ((valid && l < K) ? gam[(size_t)d * K + l] : (l < K ? 1.0 : 0.0));
    double gp = (LL && valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
    // Single-step build over padded rows (c.ell: [D][V] (term,count), (-1,0) past the document's end): the document's pairs are
    // addressed by d alone, so their loads leave with the gamma row instead of a memory round trip later (doc_ptr -> tc), every
    // document has the same V / L chunks (static register indices, no per-step shuffles)
    const bool drows = V <= PRE * L && (a.c.dense != nullptr || a.c.dense16 != nullptr);      // rows of counts: term = slot, 4 or 2 bytes per slot
    const bool rows = drows || (V <= PRE * L && a.c.ell != nullptr);      // (the grid-stride build requests the next step's row a step ahead)
    int64_t start = (valid && !rows) ? a.c.doc_ptr[d] : 0;
    int W = (valid && !rows) ? (int)(a.c.doc_ptr[d + 1] - start) : 0;
    // SINGLE: the table stays in registers (<= 5 entries per thread: KP*V <= 12 * 96, >= 4 waves) until just before the barrier, so
    // that the prologue arithmetic below runs while these loads are in flight instead of after them
    constexpr int TB = KP <= 10 ? 4 : 5;
    double tb[TB];
    if (SINGLE) {
#pragma unroll
        for (int q = 0; q < TB; ++q) { const int i = tid + q * (int)blockDim.x; tb[q] = (i < K * V) ? eB[i] : 0.0; }
    }
    for (int i = tid; i < NW * KP * V; i += blockDim.x) sSlab[i] = 0.0;
    for (int i = tid; i < KP * V; i += blockDim.x) {
        if (!SINGLE) sB[i] = (i < K * V) ? eB[i] : 0.0;
        if (LL) sBeta[i] = (i < K * V) ? bprev[i] : 0.0;
    }
    MMM_STAMP(1);

    // Grid-stride build: a two-deep software pipeline over the wave's steps.  The CSR offsets of step i+2 and the (term,count)
    // pairs + gamma row of step i+1 are requested while step i computes, so that no step starts with the two dependent memory
    // round trips doc_ptr -> tc (at 160k-640k documents they were ~70 % of a step: the SIMDs ran at 30 % VALU utilisation).
    int d1 = 0; bool valid1 = false; int64_t start1 = 0; int W1 = 0;
    if (!SINGLE) {
        d1 = base + stride + g; valid1 = (base + stride < D) && d1 < D;
        start1 = (valid1 && !rows) ? a.c.doc_ptr[d1] : 0;
        W1 = (valid1 && !rows) ? (int)(a.c.doc_ptr[d1 + 1] - start1) : 0;
    }
    int2 tcp[PRE];                           // (term,count) of the first PRE chunks of the current step
    bool first = true;
    for (;;) {
        // ---- groups start at rotated chunks so that the G documents of a wave instruction touch different term ranges of the slab
        const int NCHR = VT ? (VT + L - 1) / L : (V + L - 1) / L;      // chunks of a padded row (<= PRE)
        const int nch = rows ? NCHR : (W + L - 1) / L;
        const int rot = nch > 0 ? g % nch : 0;
        const int2* __restrict__ tcd = a.c.tc + start;
        int nchmax = nch;
        if (!rows) {
            if (G >= 2) nchmax = max(nchmax, __shfl_xor(nchmax, 32, MMM_WAVE));
            if (G >= 4) nchmax = max(nchmax, __shfl_xor(nchmax, 16, MMM_WAVE));
            nchmax = __builtin_amdgcn_readfirstlane(nchmax);
        }
        if (drows && (SINGLE || first)) {
            const int* __restrict__ row = a.c.dense + (size_t)(valid ? d : 0) * a.c.Vp;
            const unsigned short* __restrict__ row16 = a.c.dense16 + (size_t)(valid ? d : 0) * a.c.Vp;
            const bool h16 = a.c.dense16 != nullptr;
            const int slp = a.c.Vp >> 4;
            if (L == 16 && h16) {      // one 16-byte load instead of one 2-byte load per chunk (six loads whose last waited for the first five)
                const unsigned* __restrict__ r32 = (const unsigned*)(row16 + (size_t)l * slp);
                const unsigned w0 = r32[0], w1 = r32[1], w2 = r32[2], w3 = r32[3];
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    int c = j + rot; if (c >= NCHR) c -= NCHR;
                    const int w = c * L + l;
                    const bool in = valid && j < NCHR && w < V;
                    const int n = in ? row16_count(w0, w1, w2, w3, c) : 0;
                    tcp[j] = make_int2(n > 0 ? w : -1, n);
                }
            } else
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot; if (c >= NCHR) c -= NCHR;
                const int w = c * L + l;
                const bool in = valid && j < NCHR && w < V;
                const int n = in ? (h16 ? (int)row16[row_slot(w, slp)] : row[row_slot(w, slp)]) : 0;
                tcp[j] = make_int2(n > 0 ? w : -1, n);
            }
        } else if (rows && !drows && (SINGLE || first)) {
            const int2* __restrict__ row = a.c.ell + (size_t)(valid ? d : 0) * V;
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot; if (c >= NCHR) c -= NCHR;
                const int w = c * L + l;
                tcp[j] = (valid && j < NCHR && w < V) ? row[w] : make_int2(-1, 0);
            }
        } else if (!rows && (SINGLE || first)) {               // first step: loads issued before the prologue math (later steps: requested a step ahead)
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot; if (c >= nch) c -= nch;
                const int w = c * L + l;
                tcp[j] = ((j < nch) && (w < W)) ? tcd[w] : make_int2(-1, 0);
            }
        }
        // ---- Elntheta (LDA.jl:78-80), a_k = exp(Elntheta_k), theta_{t-1} (LDA.jl:92-94) ------------------------------
        double el = 0.0;
        if (ext) { if (l < KP) myA[l] = gk; }
        else {
            const double S = group_sum<L>(gk);
            const double ps = dev_digamma_pos(l < K ? gk : S);        // lane K of the group holds psi(S)
            const double psS = __shfl(ps, g * L + K, MMM_WAVE);
            el = ps - psS;
            if (l < KP) myA[l] = (l < K) ? ar_exp(el) : 0.0;
        }
        if (LL) {
            const double Sp = group_sum<L>(gp);
            if (l < KP) myT[l] = (l < K) ? gp / Sp : 0.0;
        }
        if (first) {
            if (stop) return;            // a previous pass met the stopping rule: this launch must not touch the state
            if (SINGLE) {
#pragma unroll
                for (int q = 0; q < TB; ++q) { const int i = tid + q * (int)blockDim.x; if (i < KP * V) sB[i] = tb[q]; }
            }
            __syncthreads();
            first = false;
            MMM_STAMP(2);
        } else lds_wave_sync();
        if (!ext && valid && l < K) Eln[(size_t)d * K + l] = el;
        MMM_STAMP(3);
        // ---- requests of the next two steps (grid-stride build) ---------------------------------------------------------
        int2 tcn[PRE];
        double gkn = 0.0, gpn = 0.0;
        int d2 = 0; bool valid2 = false; int64_t start2 = 0; int W2 = 0;
        const bool more = !SINGLE && base + stride < D;
        if (more) {
            const int nch1 = rows ? NCHR : (W1 + L - 1) / L;
            const int rot1 = nch1 > 0 ? g % nch1 : 0;
            const int2* __restrict__ tcd1 = a.c.tc + start1;
            if (drows) {
                const int* __restrict__ row = a.c.dense + (size_t)(valid1 ? d1 : 0) * a.c.Vp;
                const unsigned short* __restrict__ row16 = a.c.dense16 + (size_t)(valid1 ? d1 : 0) * a.c.Vp;
                const bool h16 = a.c.dense16 != nullptr;
                const int slp = a.c.Vp >> 4;
                if (L == 16 && h16) {
                    const unsigned* __restrict__ r32 = (const unsigned*)(row16 + (size_t)l * slp);
                    const unsigned w0 = r32[0], w1 = r32[1], w2 = r32[2], w3 = r32[3];
#pragma unroll
                    for (int j = 0; j < PRE; ++j) {
                        int c = j + rot1; if (c >= NCHR) c -= NCHR;
                        const int w = c * L + l;
                        const bool in = valid1 && j < NCHR && w < V;
                        const int n = in ? row16_count(w0, w1, w2, w3, c) : 0;
                        tcn[j] = make_int2(n > 0 ? w : -1, n);
                    }
                } else
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    int c = j + rot1; if (c >= NCHR) c -= NCHR;
                    const int w = c * L + l;
                    const bool in = valid1 && j < NCHR && w < V;
                    const int n = in ? (h16 ? (int)row16[row_slot(w, slp)] : row[row_slot(w, slp)]) : 0;
                    tcn[j] = make_int2(n > 0 ? w : -1, n);
                }
            } else if (rows) {
                const int2* __restrict__ row = a.c.ell + (size_t)(valid1 ? d1 : 0) * V;
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    int c = j + rot1; if (c >= NCHR) c -= NCHR;
                    const int w = c * L + l;
                    tcn[j] = (valid1 && j < NCHR && w < V) ? row[w] : make_int2(-1, 0);
                }
            } else {
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                int c = j + rot1; if (c >= nch1) c -= nch1;
                const int w = c * L + l;
                tcn[j] = ((j < nch1) && (w < W1)) ? tcd1[w] : make_int2(-1, 0);
            }
            }
            gkn = (valid1 && l < K) ? gam[(size_t)d1 * K + l] : (l < K ? 1.0 : 0.0);
            gpn = (LL && valid1 && l < K) ? gprev[(size_t)d1 * K + l] : (l < K ? 1.0 : 0.0);
            d2 = base + 2 * stride + g; valid2 = (base + 2 * stride < D) && d2 < D;
            start2 = (valid2 && !rows) ? a.c.doc_ptr[d2] : 0;
            W2 = (valid2 && !rows) ? (int)(a.c.doc_ptr[d2 + 1] - start2) : 0;
        }
        {
            double av[KP], acc[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) { av[k] = myA[k]; acc[k] = 0.0; }
            if (rows) {
#pragma unroll
                for (int j = 0; j < PRE; ++j) {
                    if (j < NCHR) {
                        int2 tcv = tcp[j];
                        const bool act = tcv.x >= 0;
                        tcv.x = act ? tcv.x : 0;
                        lda_chunk<KP, LL>(tcv, act, V, av, acc, sB, sBeta, myT, slab, ll_acc);
                    }
                }
            } else
#pragma unroll 2
            for (int j = 0; j < nchmax; ++j) {
                int2 tcv = tcp[0];
#pragma unroll
                for (int q = 1; q < PRE; ++q) tcv = (j == q) ? tcp[q] : tcv;     // register select (static indices only)
                if (j >= PRE) {
                    int c = j + rot; if (c >= nch) c -= nch;
                    const int w = c * L + l;
                    tcv = ((j < nch) && (w < W)) ? tcd[w] : make_int2(-1, 0);
                }
                const bool act = tcv.x >= 0;
                tcv.x = act ? tcv.x : 0;
                lda_chunk<KP, LL>(tcv, act, V, av, acc, sB, sBeta, myT, slab, ll_acc);
            }
            MMM_STAMP(4);
            // ---- gamma_{t+1} = alpha + sum_w phi_kw n_w (LDA.jl:83-87 of the next pass) ---------------------------
            double mine = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) { const double tot = group_sum<L>(acc[k]); if (l == k) mine = tot; }
            if (valid && l < K) gnext[(size_t)d * K + l] = a.c.alpha + mine;
        }
        MMM_STAMP(5);
        if (SINGLE) break;
        base += stride;
        if (base >= D) break;
        // ---- the next step's operands were requested above ------------------------------------------------------------
        d = d1; valid = valid1; gk = gkn; gp = gpn; start = start1; W = W1;
#pragma unroll
        for (int j = 0; j < PRE; ++j) tcp[j] = tcn[j];
        d1 = d2; valid1 = valid2; start1 = start2; W1 = W2;
        lds_wave_sync();
    }
    MMM_STAMP(6);
    // ---- block epilogue: slabs -> one partial; ll partial ------------------------------------------------------
    if (LL) ll_acc = wave_sum(ll_acc);
    __syncthreads();
    if (LL && lane == 0) sA[wid] = ll_acc;      // sA is free now
    double* out = a.partial + (size_t)blockIdx.x * K * a.pstride;
    for (int i = tid; i < K * V; i += blockDim.x) {
        double v8[kMaxWavesE];
#pragma unroll
        for (int w = 0; w < kMaxWavesE; ++w) v8[w] = (w < NW) ? sSlab[(size_t)w * KP * V + i] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kMaxWavesE; ++w) s += v8[w];
        out[a.pstride == V ? i : (i / V) * a.pstride + i % V] = s;
    }
    if (LL) {
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += sA[w];
            a.llpart[blockIdx.x] = s;
        }
    }
    MMM_STAMP(7);
}


// ---- dense-row E-step (large corpora over a small vocabulary) ---------------------------------------------------------------------
// The corpora this model is used on are dense: mutation catalogues over the 96 SNV channels list nearly every channel in every sample
// (data/brca-eu_snv_counts.tsv: 53,559 of 53,760 entries).  For such a corpus the CSR sweep of k_lda_estep spends its time in LDS: per
// nonzero an 80-byte table column read and an 80-byte read-modify-write of the wave's slab (PMC: LDS pipe, not VALU, bounds the chunk
// loop).  Here a document is a row of Vp = 16 SL counts (zeros where a term is absent); lane l of a 16-lane document group owns the terms
// l, 16 + l, ..., the same ones in every document it meets, so the statistics sum_d phi_kv n_dv of its terms stay in REGISTERS for the
// whole launch (SL * KP doubles per lane) and reach the slab once, at the end.  Per term slot: a conflict-free 16-byte-per-lane read
// of the term-major table and 2 KP + 16 f64 instructions; no atomics in the loop.  HBM per document: 4 Vp bytes of counts instead of
// 8 bytes per nonzero.  Same formulas as k_lda_estep (LDA.jl:69-108); the sums are associated per lane, then lanes, waves, blocks.
template <class T> __device__ __forceinline__ T* at_byte(T* base, unsigned off) { return (T*)((char*)base + off); }   // uniform base + 32-bit lane offset: one VGPR per address
typedef unsigned short mmm_us2 __attribute__((ext_vector_type(2)));

template <int KP, int SL, bool C16>
__global__ __launch_bounds__(512, 2) void k_lda_estep_dense(EstepArgs a, const int* __restrict__ cnt, const unsigned short* __restrict__ cnt16)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int L = 16, G = MMM_WAVE / L, Vp = L * SL;
    MMM_STAMP(0);
    const int t = a.t;
    const int stop = a.ctl->stop;
    const double* __restrict__ gam = a.gamma.s[t % 3];
    double* __restrict__ gnext = a.gamma.s[(t + 1) % 3];
    double* __restrict__ Eln = a.Elntheta.s[t % 3];
    const double* __restrict__ eB = a.expElnbeta.s[(t + 2) % 3];
    const int K = a.c.K, D = a.c.D, V = a.c.V;
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane / L, l = lane % L;
    double* sT = smem;                                   // [Vp][KP] exp(Elnbeta_{t-1}), term-major; rows v >= V hold 1 (their counts are 0)
    double* sSlab = sT + (size_t)Vp * KP;                // [NW][Vp][KP], term-major like the table (written once, in the epilogue)
    double* sA = sSlab + (size_t)NW * Vp * KP;           // [NW][G][KP]
    double* sR = sA + (size_t)NW * G * KP;               // [NW][64][KP] gamma sums, lane-major
    double* slab = sSlab + (size_t)wid * Vp * KP;
    double* myA = sA + ((size_t)wid * G + g) * KP;
    double* myR = sR + (size_t)wid * MMM_WAVE * KP;
    const int stride = gridDim.x * NW * G;
    int base = (blockIdx.x * NW + wid) * G;
    // exp / log tables of the prologue (mmm_arith.h: ar_exp_tab, ar_digamma_pos_tab -- a third of this kernel's vector work is the K + 1
    // digammas and K exps per document; no division in exp, 20 instructions fewer in the log)
    __shared__ __attribute__((aligned(16))) double sTabs[MMM_EXPTAB_N + MMM_LOGTAB_N];
    for (int i = tid; i < MMM_EXPTAB_N + MMM_LOGTAB_N; i += blockDim.x) sTabs[i] = i < MMM_EXPTAB_N ? g_mmm_exptab[i] : g_mmm_logtab[i - MMM_EXPTAB_N];

    // The next step's gamma row and counts are requested a step ahead and must stay in flight across the step: nothing between a request and
    // its use may wait for memory (vmcnt counts in order, so ONE scratch reload in the loop waits for every load before it -- the build
    // that spilled 12 registers exposed the HBM round trip in every step: 640k documents 280 us at 53 % of its own issue time).  So: loads
    // are unconditional (a clamped document index; masks are applied when the values are used), 16-bit counts land in register halves
    // (SL / 2 registers), addresses are a uniform base + one 32-bit offset per lane (D K 8 and D Vp 4 bytes < 4 GB, checked at create).
    constexpr int NC = C16 ? (SL + 1) / 2 : SL;
    int d = base + g;
    bool valid = d < D;
    const int lk = l < K ? l : K - 1;
    unsigned dl = valid ? (unsigned)d : 0u;
    double gk = *at_byte(gam, (dl * (unsigned)K + lk) * 8u);
    gk = (valid && l < K) ? gk : (l < K ? 1.0 : 0.0);
    constexpr int SLs = C16 ? 2 * NC : SL;                  // slots a lane owns in a stored row
    unsigned c[NC], cn[NC];
    auto request = [&](unsigned* o, unsigned dd) {          // the lane's part of the row: NC consecutive 32-bit words, one load
        const unsigned* row = C16 ? at_byte((const unsigned*)cnt16, (dd * (unsigned)(16 * SLs) + l * SLs) * 2u)
                                  : at_byte((const unsigned*)cnt, (dd * (unsigned)(16 * SLs) + l * SLs) * 4u);
#pragma unroll
        for (int j = 0; j < NC; ++j) o[j] = row[j];
    };
    auto take = [&](const unsigned* raw, bool ok) {
#pragma unroll
        for (int j = 0; j < NC; ++j) c[j] = ok ? raw[j] : 0u;
    };
    request(cn, dl);
    take(cn, valid);
    for (int i = tid; i < Vp * KP; i += blockDim.x) {
        const int v = i / KP, k = i % KP;
        sT[i] = (k < K) ? (v < V ? eB[(size_t)k * V + v] : 1.0) : 0.0;
    }
    __syncthreads();          // the function tables are read by the first step's prologue
    double st[SL][KP];
#pragma unroll
    for (int q = 0; q < SL; ++q)
#pragma unroll
        for (int k = 0; k < KP; ++k) st[q][k] = 0.0;
    bool first = true;
    for (;;) {
        // ---- the next step's gamma row and counts are requested before this step's term phase
        const int dn = d + stride;
        const bool more = base + stride < D, validn = more && dn < D;
        const unsigned dnl = validn ? (unsigned)dn : dl;
        // ---- Elntheta (LDA.jl:78-80), a_k = exp(Elntheta_k)
        const double S = group_sum<L>(gk);
        const double ps = ar_digamma_pos_tab(l < K ? gk : S, sTabs + MMM_EXPTAB_N);        // lane K of the group holds psi(S)
        const double psS = __shfl(ps, g * L + K, MMM_WAVE);
        const double el = ps - psS;
        const double ak = (l < K) ? ar_exp_tab(el, sTabs) : 0.0;
        if (l < KP) myA[l] = ak;
        // (requested here, after the prologue: its polynomial constants overflow the scalar registers and one is reloaded from scratch in there)
        const double gkn = *at_byte(gam, (dnl * (unsigned)K + lk) * 8u);
        request(cn, dnl);
        if (first) {
            if (stop) return;            // a previous pass met the stopping rule: this launch must not touch the state
            __syncthreads();
            first = false;
            MMM_STAMP(1);
        } else lds_wave_sync();
        if (valid && l < K) *at_byte(Eln, (dl * (unsigned)K + l) * 8u) = el;
        double acc[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] = 0.0;
        // ---- phi_kv n_v (LDA.jl:92-106) for the lane's SL terms.  (Tried at K = 10, V = 96, 640k documents: a_k re-read from LDS in every slot,
        // no spilled register instead of 12: 330 vs 307 us; the next slot's table row requested a slot ahead, 62 spilled: 509 us.)
        double av[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) av[k] = myA[k];
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            const double* tb = sT + (size_t)(q * L + l) * KP;
            // three fused multiply-adds per (term slot, topic): the normaliser s = sum_k a_k B_kv, the lane's gamma sums WITHOUT their factor
            // a_k (it is the document's, applied once after the lanes' sums have met) and the statistics WITHOUT their factor B_kv (it is
            // the term's, the same for every document, applied once when the registers reach the slab)
            double b[KP], s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) b[k] = tb[k];
#pragma unroll
            for (int k = 0; k + 1 < KP; k += 2) { s0 = fma(av[k], b[k], s0); s1 = fma(av[k + 1], b[k + 1], s1); }
            if (KP & 1) s0 = fma(av[KP - 1], b[KP - 1], s0);
            // a slot without mass must not see 0 x rcp(0) = NaN: with tiny priors the normaliser of a never-observed term underflows to 0.
            // (v_max with the smallest normal: the bits of every other quotient are unchanged; a select on the count costs 150 spilled registers here)
            const unsigned cq = C16 ? ((q & 1) ? c[q / 2] >> 16 : c[q / 2] & 0xffffu) : c[q];
            const double r = (double)cq * dev_rcp(dev_max_raw(s0 + s1, 2.2250738585072014e-308));
#pragma unroll
            for (int k = 0; k < KP; ++k) { acc[k] = fma(b[k], r, acc[k]); st[q][k] = fma(av[k], r, st[q][k]); }
            // one slot at a time, its statistics updated here (left alone the compiler sinks the SL KP updates to the end of the step and keeps
            // every slot's products alive until then: 190 spilled registers)
#pragma unroll
            for (int k = 0; k < KP; ++k) asm volatile("" : "+v"(st[q][k]));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        MMM_STAMP(2);
        // ---- the requested values are taken over HERE, before the step's last store: the compiler prices a wait for loads as if the
        // (lane-conditional) stores after them had not been issued, i.e. as vmcnt(0) -- placed after the gamma store below it waited for
        // that store's round trip in every step
        const bool valid_now = valid;
        const unsigned dl_now = dl;
        // (the empty statements are the first use of the loaded registers and cannot move above the term phase's own)
        double gk_next = gkn;
        asm volatile("" : "+v"(gk_next) :: "memory");
#pragma unroll
        for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(cn[j]) :: "memory");
        gk_next = (validn && l < K) ? gk_next : (l < K ? 1.0 : 0.0);
        take(cn, validn);
#pragma unroll
        for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(c[j]));
        __builtin_amdgcn_sched_barrier(0);
        // ---- gamma_{t+1} = alpha + sum_v phi_kv n_v: the lanes' sums meet in LDS, lane k of the group adds its column
#pragma unroll
        for (int k = 0; k < KP; ++k) myR[(size_t)lane * KP + k] = acc[k];
        lds_wave_sync();
        if (l < K) {
            const double* col = myR + (size_t)(g * L) * KP + l;
            double r0 = col[0], r1 = col[KP], r2 = col[2 * KP], r3 = col[3 * KP];
#pragma unroll
            for (int j = 4; j < L; j += 4) { r0 += col[j * KP]; r1 += col[(j + 1) * KP]; r2 += col[(j + 2) * KP]; r3 += col[(j + 3) * KP]; }
            if (valid_now) *at_byte(gnext, (dl_now * (unsigned)K + l) * 8u) = fma(ak, (r0 + r1) + (r2 + r3), a.c.alpha);
        }
        base += stride;
        if (base >= D) break;
        d = dn; valid = validn; dl = dnl; gk = gk_next;
        lds_wave_sync();
    }
    MMM_STAMP(3);
    // ---- the wave's statistics: the four document groups' registers are added across the rows of the wave (rows_sum4: (g0 + g2) + (g1 + g3),
    // no LDS), multiplied by the term's table entry once, and the first group's lanes store them -- the slab is term-major with padded
    // bounds like the table, written once (no zero fill, no read-modify-write; 16-byte pairs at compile-time offsets).  (One group at a
    // time through LDS with run-time bounds, every entry its own round trip: 19 of the 27 us of a 15k-document launch,
    // tools/diag_dense_stamps.py; batched per term slot: 6.4.)
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        double* sl = slab + (size_t)(q * L + l) * KP;
        const double* tb = sT + (size_t)(q * L + l) * KP;
        double t[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) t[k] = rows_sum4(st[q][k]) * tb[k];
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) sl[k] = t[k];
        }
    }
    MMM_STAMP(4);
    __syncthreads();
    MMM_STAMP(5);
    double* out = a.partial + (size_t)blockIdx.x * K * a.pstride;
    for (int i = tid; i < K * V; i += blockDim.x) {
        const int kk = i / V, v = i - kk * V;
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += sSlab[((size_t)w * Vp + v) * KP + kk];
        out[a.pstride == V ? i : kk * a.pstride + v] = s;
    }
    MMM_STAMP(6);
    MMM_STAMP(7);
}

// ---- slab reduction + log-likelihood / stopping rule ------------------------------------------------------------
struct ReduceArgs {
    const double* partial; const double* llpart; int nslab; int VK;
    double* stats;         // out: [VK] summed lambda statistics of pass t, [VK] ll numerator of pass t-1
    LdaCtl* ctl;
    int t;                 // this pass (host count)
    double Nglobal, tol;
    double* ll_hist;
    int do_ll, conv_base, run_tail;
    // several GPUs with the mailboxes up: k_lda_reduce sends its entries to the peers as it produces them and k_lda_mstep sums
    // the contributions in rank order as it consumes them -- the all-reduce costs no launch of its own
    int p2p; unsigned int p2p_seq; P2PArgs px;
    // the log-likelihood of pass t-1 evaluated by extra blocks of the reduce launch (k_lda_reduce_ll) instead of inside the
    // E-step kernel: per-block numerators in llpart2[n_ll], summed (and exchanged) by the pass-tail block
    const double* llpart2; int n_ll, ll_in_k2;
    // RCCL transport: the numerator has to sit in stats[VK] before ncclAllReduce, so wave 1 of reduce block 0 collects the ll blocks'
    // numerators inside the reduce launch, through seq-tagged cells (as k_lda_reduce_ll_mstep does); the tail then only finishes
    unsigned long long* ll_cells; unsigned int ll_seq;
};

// ll_{t-1}, the convergence test of common.jl:53-56 after > 10 values (LDA.jl:215) and t += 1 (one thread)
__device__ void lda_pass_tail(const ReduceArgs& r)
{
    int stop = 0;
    if (r.do_ll) {
        const int n = r.ctl->n_hist;
        const double ll = r.stats[r.VK] / r.Nglobal;
        r.ll_hist[n] = ll;
        r.ctl->n_hist = n + 1;
        if (n + 1 - r.conv_base > 10) {
            const double prev = r.ll_hist[n - 1];
            if (fabs(prev - ll) / fabs(ll) < r.tol) { stop = 1; r.ctl->stop = 1; r.ctl->stop_iter = r.t - 1; }
        }
    }
    if (!stop) r.ctl->t = r.t;     // on convergence at t-1 the state of pass t is discarded
    r.ctl->ticket = 0;
}

// the pass-tail block (one wave) of the M-step launches: finishes the ll numerator of pass t-1 -- sum of the k_lda_reduce_ll
// partials and/or the peers' share -- and runs lda_pass_tail
template <bool P2P>
__device__ __forceinline__ void lda_tail_block(const ReduceArgs& r, int lane)
{
    double v = 0.0;
    const bool from_parts = r.ll_in_k2 && !r.ll_cells;
    if (from_parts && r.do_ll) {
        for (int i = lane; i < r.n_ll; i += 64) v += r.llpart2[i];
        v = wave_sum(v);
    }
    if (lane != 0) return;
    if (from_parts) {
        if (r.do_ll) {
            if (P2P && r.p2p) { p2p_send(r.px, r.p2p_seq, r.VK, v); v = p2p_recv_sum(r.px, r.p2p_seq, r.VK, v); }
            r.stats[r.VK] = v;
        }
    } else if (P2P && r.p2p) r.stats[r.VK] = p2p_recv_sum(r.px, r.p2p_seq, r.VK, r.stats[r.VK]);
    lda_pass_tail(r);
}

// log-likelihood numerator of pass t-1 (LDA.jl:174-188 with theta_{t-1} = gamma_{t-1} / sum, beta_{t-1}) for the documents of
// "ll block" lb of nlb, by a block of 16 waves laid out like k_lda_reduce's (16 x 64 threads): L lanes per document (as in the
// E-step), 64/L documents per wave step, beta staged in LDS -- the ll half of the E-step's chunk loop, moved out of it.
template <int KP, int L>
__device__ void lda_ll_block(const LdaDev& c, const double* __restrict__ gprev, const double* __restrict__ bprev, double* llpart2, int lb, int nlb,
                             double* smem, unsigned long long* cell = nullptr, unsigned int seq = 0)
{
    __shared__ double s_w[16];
    const int tid = threadIdx.y * 16 + threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int G = MMM_WAVE / L;
    const int g = lane / L, l = lane % L;
    const int K = c.K, V = c.V, D = c.D;
    double* sBeta = smem;
    double* myT = smem + (size_t)KP * V + ((size_t)wid * G + g) * KP;
    double* sLog = smem + (size_t)KP * V + (size_t)64 * KP;         // [256] the log table (dev_log_tab)
    // the first step's document loads go out before the table is staged (as in the E-step kernel).  (Splitting a document
    // group's chunks over 2 or 4 waves -- more, lighter blocks on the CUs the reduction leaves idle -- was slower: 29.7 / 33.6
    // vs 26.6 us per iteration; the launch is bound by block dispatch and table staging, not by the sweep's arithmetic.)
    // wave w of block lb is wave slot w * nlb + lb: the documents fill wave 0 of every block, then wave 1, ... -- a corpus of fewer
    // than 64 nlb documents leaves every block the same number of busy waves (the sweep is issue-bound per CU: 157 blocks of 16 busy
    // waves were 0.6 us slower at BASELINE config 2 than 192 blocks of 13)
    // (corpora of more than one step per wave keep a block's waves on neighbouring documents)
    const int wslot = ((int64_t)nlb * 16 * G >= (int64_t)D) ? wid * nlb + lb : lb * 16 + wid, nslots = nlb * 16;
    int base = wslot * G;
    int d = base + g;
    bool valid = d < D;
    double gp = (valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
    constexpr int PRE = 128 / L;          // padded rows: every chunk of the document is requested up front, no doc_ptr needed
    const bool dense = L == 16 && (c.dense != nullptr || c.dense16 != nullptr);       // rows of counts: term = slot index, 4 or 2 bytes per slot, table columns read in lane order
    const bool h16 = c.dense16 != nullptr;
    const bool ell = dense || c.ell != nullptr;
    int2 pre[PRE];
    // 16-bit lane-major rows: the lane's <= 8 slots are the first words of ONE 16-byte load (the rows are allocated with 16 bytes to spare),
    // and in the loop below the next step's gamma row and counts are requested while this step computes (the rules of k_lda_estep_dense)
    const bool fast = L == 16 && dense && h16 && (int64_t)D * K * 8 < ((int64_t)1 << 32) && (int64_t)D * c.Vp * 2 < ((int64_t)1 << 32);
    unsigned wq[4] = {0u, 0u, 0u, 0u};
    const int lk = l < K ? l : K - 1;
    unsigned dl = valid ? (unsigned)d : 0u;
    if (fast) {
        const unsigned* row = at_byte((const unsigned*)c.dense16, (dl * 16u + (unsigned)l) * (unsigned)(c.Vp >> 4) * 2u);
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = row[j];
    } else if (dense) {
        const int* __restrict__ row = c.dense + (size_t)(valid ? d : 0) * c.Vp;
        const unsigned short* __restrict__ row16 = c.dense16 + (size_t)(valid ? d : 0) * c.Vp;
#pragma unroll
        for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? make_int2(j * L + l, h16 ? (int)row16[row_slot(j * L + l, c.Vp >> 4)] : row[row_slot(j * L + l, c.Vp >> 4)]) : make_int2(-1, 0);
    } else if (ell) {
        const int2* __restrict__ row = c.ell + (size_t)(valid ? d : 0) * V;
#pragma unroll
        for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? row[j * L + l] : make_int2(-1, 0);
    }
    int64_t start = (!ell && valid) ? c.doc_ptr[d] : 0;
    int W = (!ell && valid) ? (int)(c.doc_ptr[d + 1] - start) : 0;
    // term-major copy [v][KP] of beta_{t-1}: a lane reads the KP entries of its term as 16-byte pairs at immediate offsets (lane stride
    // 8 KP bytes: the 16 lanes of a document group cover the banks once), instead of KP reads with an address computed for each
    for (int i = tid; i < KP * V; i += 1024) { const int v = i / KP, k = i - v * KP; sBeta[i] = (k < K) ? bprev[(size_t)k * V + v] : 0.0; }
    if (tid < MMM_LOGTAB_N) sLog[tid] = g_mmm_logtab[tid];
    MMM_RSTAMP(lb == 0 && tid == 0, 20);       // own loads (gamma row, document row, table entries) have arrived
    __syncthreads();
    MMM_RSTAMP(lb == 0 && tid == 0, 21);       // tables staged by all waves
    double acc = 0.0;
    if (fast) {
        const int nch = (V + L - 1) / L;
        const int stride = nslots * G;
        const unsigned slp2 = (unsigned)(c.Vp >> 4) * 2u;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = valid ? wq[j] : 0u;
        for (; base < D; base += stride) {
            const int dn = d + stride;
            const bool validn = base + stride < D && dn < D;
            const unsigned dnl = validn ? (unsigned)dn : dl;
            const double Sp = group_sum<L>(gp);
            lds_wave_sync();
            if (l < KP) myT[l] = (l < K) ? gp / Sp : 0.0;
            lds_wave_sync();
            double tv[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) tv[k] = myT[k];
            // the next step's values (unconditional loads, clamped indices; taken over after the chunks)
            double gpn = *at_byte(gprev, (dnl * (unsigned)K + (unsigned)lk) * 8u);
            {
                const unsigned* row = at_byte((const unsigned*)c.dense16, (dnl * 16u + (unsigned)l) * slp2);
#pragma unroll
                for (int j = 0; j < 4; ++j) wq[j] = row[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nch) {
                    const unsigned cq = (j & 1) ? w[j / 2] >> 16 : w[j / 2] & 0xffffu;
                    const bool act = valid && j * L + l < V;
                    const double* bc = sBeta + (size_t)(act ? j * L + l : 0) * KP;
                    double p0 = 0.0, p1 = 0.0;
#pragma unroll
                    for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(tv[k], bc[k], p0); p1 = fma(tv[k + 1], bc[k + 1], p1); }
                    if (KP & 1) p0 = fma(tv[KP - 1], bc[KP - 1], p0);
                    const double p = act ? p0 + p1 : 1.0;
                    acc = fma((double)cq, dev_log_tab(p, sLog), acc);
                }
            }
            asm volatile("" : "+v"(acc) :: "memory");
            asm volatile("" : "+v"(gpn) :: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(wq[j]) :: "memory");
            d = dn; valid = validn; dl = dnl;
            gp = (valid && l < K) ? gpn : (l < K ? 1.0 : 0.0);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = valid ? wq[j] : 0u;
        }
    } else if (ell) {
        const int nch = (V + L - 1) / L;
        for (; base < D; base += nslots * G) {
            if (base != wslot * G) {
                d = base + g; valid = d < D;
                gp = (valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
                if (dense) {
                    const int* __restrict__ row = c.dense + (size_t)(valid ? d : 0) * c.Vp;
                    const unsigned short* __restrict__ row16 = c.dense16 + (size_t)(valid ? d : 0) * c.Vp;
#pragma unroll
                    for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? make_int2(j * L + l, h16 ? (int)row16[row_slot(j * L + l, c.Vp >> 4)] : row[row_slot(j * L + l, c.Vp >> 4)]) : make_int2(-1, 0);
                } else {
                    const int2* __restrict__ row = c.ell + (size_t)(valid ? d : 0) * V;
#pragma unroll
                    for (int j = 0; j < PRE; ++j) pre[j] = (valid && j * L + l < V) ? row[j * L + l] : make_int2(-1, 0);
                }
            }
            const double Sp = group_sum<L>(gp);
            lds_wave_sync();
            if (l < KP) myT[l] = (l < K) ? gp / Sp : 0.0;
            lds_wave_sync();
            double tv[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) tv[k] = myT[k];
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                if (j >= nch) break;
                const int2 t = pre[j];
                const bool act = t.x >= 0;
                const double* bc = sBeta + (size_t)(act ? t.x : 0) * KP;
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(tv[k], bc[k], p0); p1 = fma(tv[k + 1], bc[k + 1], p1); }
                if (KP & 1) p0 = fma(tv[KP - 1], bc[KP - 1], p0);
                const double p = act ? p0 + p1 : 1.0;
                acc = fma((double)t.y, dev_log_tab(p, sLog), acc);
            }
        }
    } else
    for (; base < D; base += nslots * G) {
        if (base != wslot * G) {
            d = base + g; valid = d < D;
            gp = (valid && l < K) ? gprev[(size_t)d * K + l] : (l < K ? 1.0 : 0.0);
            start = valid ? c.doc_ptr[d] : 0;
            W = valid ? (int)(c.doc_ptr[d + 1] - start) : 0;
        }
        const int2* __restrict__ tcd = c.tc + start;
        const double Sp = group_sum<L>(gp);
        lds_wave_sync();
        if (l < KP) myT[l] = (l < K) ? gp / Sp : 0.0;
        lds_wave_sync();
        double tv[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) tv[k] = myT[k];
        int nchmax = (W + L - 1) / L;
        if (G >= 2) nchmax = max(nchmax, __shfl_xor(nchmax, 32, MMM_WAVE));
        if (G >= 4) nchmax = max(nchmax, __shfl_xor(nchmax, 16, MMM_WAVE));
        nchmax = __builtin_amdgcn_readfirstlane(nchmax);
        for (int j = 0; j < nchmax; ++j) {
            const int w = j * L + l;
            const bool act = w < W;
            const int2 t = act ? tcd[w] : make_int2(0, 0);
            const double* bc = sBeta + (size_t)t.x * KP;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int k = 0; k + 1 < KP; k += 2) { p0 = fma(tv[k], bc[k], p0); p1 = fma(tv[k + 1], bc[k + 1], p1); }
            if (KP & 1) p0 = fma(tv[KP - 1], bc[KP - 1], p0);
            const double p = act ? p0 + p1 : 1.0;
            acc = fma((double)t.y, dev_log_tab(p, sLog), acc);
        }
    }
    MMM_RSTAMP(lb == 0 && tid == 0, 22);       // sweep done
    acc = wave_sum(acc);
    if (lane == 0) s_w[wid] = acc;
    __syncthreads();
    MMM_RSTAMP(lb == 0 && tid == 0, 23);       // all waves done
    if (tid == 0) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += s_w[w];
        if (cell) cell_store(cell, v, seq); else llpart2[lb] = v;
    }
}

// grid = ceil(V*K/16) blocks of (16 entries, 64 slab lanes): fixed-order (deterministic) sum of the per-block partials
__device__ void lda_reduce_block(const ReduceArgs& r)
{
    __shared__ double sm[64][17];
    const int stop = r.ctl->stop;        // only the stores depend on it: the partial loads below are issued alongside this load
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int e = blockIdx.x * 16 + tx;
    double acc = 0.0;
    if (e < r.VK) for (int sl = ty; sl < r.nslab; sl += 64) acc += r.partial[(size_t)sl * r.VK + e];
    if (stop) {      // a no-op pass still keeps the mailbox rendezvous of its sequence number (p2p.hip header): element 0, value unused
        if (r.p2p && blockIdx.x == 0 && tx == 0 && ty == 0) p2p_send(r.px, r.p2p_seq, 0, 0.0);
        return;
    }
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty < 8) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[ty * 8 + j][tx];
        sm[ty * 8][tx] = v;
    }
    __syncthreads();
    if (ty == 0 && e < r.VK) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[j * 8][tx];
        r.stats[e] = v;
        if (r.p2p) p2p_send(r.px, r.p2p_seq, e, v);
    }
    if (blockIdx.x == 0 && ty >= 4 && ty < 8 && r.ll_cells && r.do_ll) {      // wave 1 of block 0: ll numerator of pass t-1 from the ll blocks' cells
        const int lane = (ty * 16 + tx) & 63;
        double v = 0.0;
        v = cells_wait_sum(r.ll_cells, r.n_ll, lane, r.ll_seq, r.ctl);
        v = wave_sum(v);
        if (lane == 0) r.stats[r.VK] = v;
    }
    if (blockIdx.x == 0 && ty == 1 && !r.ll_in_k2 && !r.ll_cells) {       // wave 1 of block 0: ll numerator of pass t-1 (from the E-step's partials)
        double v = 0.0;
        for (int i = tx + 16 * 0; i < r.nslab; i += 16) v += r.llpart[i];
        v = group_sum<16>(v);
        if (tx == 0) { r.stats[r.VK] = v; if (r.p2p) p2p_send(r.px, r.p2p_seq, r.VK, v); }
    }
}

__global__ __launch_bounds__(1024) void k_lda_reduce(ReduceArgs r) { lda_reduce_block(r); }

// the same launch with the ll of pass t-1 riding along: blocks [0, nred) are k_lda_reduce's, blocks [nred, gridDim) evaluate the
// log-likelihood numerators while the reduction -- 60 blocks -- leaves most of the chip idle
template <int KP>
__global__ __launch_bounds__(1024) void k_lda_reduce_ll(ReduceArgs r, LdaDev c, const double* gprev, const double* bprev, double* llpart2, int nred)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x < nred) { lda_reduce_block(r); return; }
    if (r.ctl->stop) return;
    constexpr int L = KP <= 15 ? 16 : (KP <= 31 ? 32 : 64);       // K <= KP: the E-step's lane-group width (K = 16 -> KP = 16 -> 32 lanes)
    const int lb = (int)blockIdx.x - nred;
    lda_ll_block<KP, L>(c, gprev, bprev, llpart2, lb, (int)gridDim.x - nred, smem, r.ll_cells ? r.ll_cells + 2 * lb : nullptr, r.ll_seq);
}

// ---- V <= 256, plain LDA, no RCCL in the path: the reduction, the ll sweep AND the M-step in one launch.  A topic's column sum needs the V/16
// reduce blocks of that topic; they hand each other their partial column sums through 16-byte cells in device memory --
// {low half | seq} {high half | seq}, complete when both words carry this launch's sequence number, so no fence and no
// flag (the mailbox format of p2p.hip) -- and then run the M-step of their own 16 entries, in parallel, while the ll blocks
// are still sweeping.  Wave 1 of the first reduce block collects the ll blocks' numerators the same way and runs the pass tail.
// The reduce blocks have the lowest block ids (dispatched first; putting the ll blocks first was 0.4 us slower) and wait only for
// each other and for the ll blocks, which wait for nothing; every wait has an iteration cap (ctl->wait_timeout, reported by the
// next host synchronisation).
struct IldaDesc {
    int I, V, K, SJ;
    int J[kIldaMaxI], joff[kIldaMaxI + 1];     // joff = prefix sums of J
    double eta[kIldaMaxI];
    const int* features;                       // [i*V + v], 0-based feature values
};

// ILDA in the merged launch: the factor arrays of the pass's ring slot and the cells the blocks of a topic use to hand each other
// their partial folds (16 per block: sum(J) <= 16)
struct IldaMerge {
    IldaDesc ds;
    double* ilam; double* iEln; double* ibeta;
    unsigned long long* fcells;
};

struct MergeArgs {
    int V; double eta;
    Ring lambda, Elnbeta, expElnbeta, beta;
    unsigned long long* cells;      // [nred] column-sum cells, then [512] ll cells
    unsigned int seq;               // never reused (a discarded pass must not leave valid-looking cells behind)
    int nred;
    int ll_join;                    // large corpora (the ll blocks loop over their documents): the reduce blocks 1.. take a share of the ll sweep
                                    // once their 16 entries are done -- the launch holds only as many blocks as are resident at once (61 of the
                                    // 256 at K = 10, V = 96 are reduce blocks, busy for ~6 us of a ~200 us sweep at 640k documents)
    int n_ll;                       // ll blocks [nred, nred + n_ll)
    // pro: the ll blocks also form Elntheta_{t+1} = psi(gamma_{t+1}) - psi(sum) and a = exp(Elntheta_{t+1}) (LDA.jl:78-80) of their documents
    // for the NEXT pass's single-step E-step kernel, after their numerator has left (one step covers the corpus: n_ll x 64 >= D)
    int pro;
    const double* pro_gamma; double* pro_Eln; double* pro_a;
};

// P2P: several GPUs with the mailboxes up -- a reduce block sends its 16 sums to the peers and adds theirs (rank order) before the
// column-sum exchange, the tail wave does the same with the ll numerator: the all-reduce rides inside this launch.
// ILDA (sum(J) <= 16, one GPU): the blocks of a topic exchange their partial FOLDS of the statistics onto the feature values
// (one cell per (feature, value)) instead of one column sum, every block forms the topic's lambda[i][j] from them in block
// order, and writes the effective tables of its own 16 entries; the topic's first block also writes the factor arrays.
template <int KP, bool P2P, bool ILDA>
__global__ __launch_bounds__(1024) void k_lda_reduce_ll_mstep(ReduceArgs r, LdaDev c, const double* gprev, const double* bprev, MergeArgs ms, IldaMerge im)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double sm[64][17];
    const int stop = r.ctl->stop;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 16 + tx;
    MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 0);                         // reduce block 1, wave 0
    MMM_RSTAMP(blockIdx.x == 0 && tid == 64, 8);                        // tail wave
    MMM_RSTAMP((int)blockIdx.x == ms.nred && tid == 0, 16);              // first ll block
    if ((int)blockIdx.x >= ms.nred) {        // ---- ll block: numerator of pass t-1 into its cell
        if (stop) return;
        constexpr int L = KP <= 15 ? 16 : (KP <= 31 ? 32 : 64);
        const int lb = (int)blockIdx.x - ms.nred, n_ll = ms.n_ll;
        // ms.pro: the block also runs the NEXT pass's prologue for its documents (the four of each wave, 16 lanes per document as
        // k_lda_estep<., 16, ...> has them): gamma_{t+1} is requested before the sweep and used after the block's numerator has left --
        // the pass tail (wave 1 of block 0, the end of this launch's critical path) does not wait a cycle longer for it
        double gnx = 0.0;
        int pd = 0;
        if constexpr (KP <= 12) {
            if (ms.pro) {
                const int lane = tid & 63, l = lane & 15;
                pd = ((tid >> 6) * n_ll + lb) * 4 + (lane >> 4);          // lda_ll_block's wave slots
                gnx = (pd < c.D && l < c.K) ? ms.pro_gamma[(size_t)pd * c.K + l] : (l < c.K ? 1.0 : 0.0);
            }
        }
        lda_ll_block<KP, L>(c, gprev, bprev, nullptr, lb, n_ll + (ms.ll_join ? ms.nred - 1 : 0), smem, ms.cells + 2 * (ms.nred + lb), ms.seq);
        MMM_RSTAMP((int)blockIdx.x == ms.nred && tid == 0, 17);
        if constexpr (KP <= 12) {
            if (ms.pro) {          // Elntheta_{t+1}, exp(Elntheta_{t+1}) (LDA.jl:78-80): the operations of the E-step kernel's prologue
                const int lane = tid & 63, g = lane >> 4, l = lane & 15, K = c.K;
                const double S = group_sum<16>(gnx);
                const double ps = dev_digamma_pos(l < K ? gnx : S);        // lane K of the group holds psi(S)
                const double psS = __shfl(ps, g * 16 + K, MMM_WAVE);
                const double el = ps - psS;
                if (pd < c.D && l < K) { ms.pro_Eln[(size_t)pd * K + l] = el; ms.pro_a[(size_t)pd * K + l] = ar_exp(el); }
            }
        }
        return;
    }
    // ---- reduce block: 16 entries of the statistics (as lda_reduce_block)
    const int rb = (int)blockIdx.x;             // reduce block
    const int e = rb * 16 + tx;
    double acc = 0.0;
    for (int sl = ty; sl < r.nslab; sl += 64) acc += r.partial[(size_t)sl * r.VK + e];
    if (stop) {      // a no-op pass still keeps the mailbox rendezvous of its sequence number (p2p.hip header): element 0, value unused
        if (P2P && rb == 0 && tid == 0) { p2p_send(r.px, r.p2p_seq, 0, 0.0); (void)p2p_recv_sum(r.px, r.p2p_seq, 0, 0.0); }
        return;
    }
    MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 1);                         // partial loads done
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty < 8) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[ty * 8 + j][tx];
        sm[ty * 8][tx] = v;
    }
    __syncthreads();
    MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 2);                         // tree done
    if (ty == 0) {                           // lanes 0..15 of wave 0
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += sm[j * 8][tx];
        if (P2P) { p2p_send(r.px, r.p2p_seq, e, v); v = p2p_recv_sum(r.px, r.p2p_seq, e, v); }
        r.stats[e] = v;
        // ---- M-step of these 16 entries (LDA.jl:96-112): column sum = the topic's block sums in block order.  Rows are padded
        //      to a multiple of 16 (Vp): a block never straddles two topics; pad entries carry zeros and are not written back
        const int V = ms.V, Vp = (V + 15) & ~15, nb = Vp / 16, k = e / Vp, vv = e - k * Vp, slot = r.t % 3;
        const bool real = vv < V;
        if (ILDA) {
            const IldaDesc& ds = im.ds;
            const int K = ds.K, SJ = ds.SJ;
            // (feature, value) of lane q = tx; partial fold of this block's 16 entries onto it (ILDA.jl:107-126)
            int qi = 0, qj = 0;
            double mine = 0.0;
            for (int i = 0; i < ds.I; ++i) {
                const int fi = real ? ds.features[(size_t)i * V + vv] : -1;
                for (int j = 0; j < ds.J[i]; ++j) {
                    const double pq = group_sum<16>(fi == j ? v : 0.0);
                    if (tx == ds.joff[i] + j) { mine = pq; qi = i; qj = j; }
                }
            }
            if (tx < SJ) cell_store(im.fcells + 2 * ((size_t)rb * 16 + tx), mine, ms.seq);
            double lam = 0.0;
            if (tx < SJ) {
                lam = ds.eta[qi];
                for (int b = 0; b < nb; ++b) lam += cell_wait(im.fcells + 2 * ((size_t)(k * nb + b) * 16 + tx), ms.seq, r.ctl);
            }
            double cs = 0.0;
            for (int i = 0; i < ds.I; ++i) {
                const double ci = group_sum<16>((tx < SJ && qi == i) ? lam : 0.0);
                if (tx < SJ && qi == i) cs = ci;
            }
            double el = 0.0, bq = 1.0;
            if (tx < SJ) {
                el = dev_digamma_pos(lam) - dev_digamma_pos(cs); bq = lam / cs;
                if (vv == tx) {         // the topic's first block (its lane 0 sits on entry 0 of the row) keeps the model arrays (ILDA.jl:6-9 layout)
                    const size_t o = (size_t)K * ds.joff[qi] + (size_t)ds.J[qi] * k + qj;
                    im.ilam[o] = lam; im.iEln[o] = el; im.ibeta[o] = bq;
                }
            }
            lds_wave_sync();
            sm[1][tx] = el; sm[2][tx] = bq;          // (the reduction tree above is done with sm)
            lds_wave_sync();
            if (real) {
                double ee = 0.0, bb = 1.0;
                for (int i = 0; i < ds.I; ++i) { const int q = ds.joff[i] + ds.features[(size_t)i * V + vv]; ee += sm[1][q]; bb *= sm[2][q]; }
                const size_t o = (size_t)k * V + vv;
                ms.Elnbeta.s[slot][o] = ee; ms.expElnbeta.s[slot][o] = exp(ee); ms.beta.s[slot][o] = bb;
            }
        } else {
        const double lam = real ? ms.eta + v : 0.0;
        const double part = group_sum<16>(lam);
        if (tx == 0) cell_store(ms.cells + 2 * rb, part, ms.seq);
        const double got = (tx < nb) ? cell_wait(ms.cells + 2 * (k * nb + tx), ms.seq, r.ctl) : 0.0;
        double cs = 0.0;
        for (int j = 0; j < nb; ++j) cs += __shfl(got, j, 16);
        MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 3);                     // column sum in hand
        const double el = dev_digamma_pos(real ? lam : 1.0) - dev_digamma_pos(cs);
        const size_t o = (size_t)k * V + vv;
        if (real) { ms.lambda.s[slot][o] = lam; ms.Elnbeta.s[slot][o] = el; ms.expElnbeta.s[slot][o] = exp(el); ms.beta.s[slot][o] = lam / cs; }
        MMM_RSTAMP(blockIdx.x == 1 && tid == 0, 4);                     // M-step stores done
        }
    }
    if (ms.ll_join && rb > 0) {              // (uniform per block; block 0 keeps the pass tail)
        constexpr int L = KP <= 15 ? 16 : (KP <= 31 ? 32 : 64);
        const int n_ll = ms.n_ll, lb = n_ll + rb - 1;
        lda_ll_block<KP, L>(c, gprev, bprev, nullptr, lb, n_ll + ms.nred - 1, smem, ms.cells + 2 * (ms.nred + lb), ms.seq);
        return;
    }
    if (rb == 0 && ty >= 4 && ty < 8) {      // wave 1 of block 0: ll numerator of pass t-1, stopping rule, pass counter
        const int lane = tid & 63, n_ll = ms.n_ll + (ms.ll_join ? ms.nred - 1 : 0);
        // what the tail needs from memory is fetched before the wait, not after it (lda_pass_tail's dependent loads)
        const int n = r.ctl->n_hist;
        const double prev = (r.do_ll && n > 0) ? r.ll_hist[n - 1] : 0.0;
        double v = 0.0;
        if (r.do_ll) {
            MMM_RSTAMP(lane == 0, 9);
            v = cells_wait_sum(ms.cells + 2 * ms.nred, n_ll, lane, ms.seq, r.ctl);
            MMM_RSTAMP(lane == 0, 10);
            v = wave_sum(v);
            if (P2P && lane == 0) { p2p_send(r.px, r.p2p_seq, r.VK, v); v = p2p_recv_sum(r.px, r.p2p_seq, r.VK, v); }
        }
        if (lane == 0) {
            int halt = 0;
            if (r.do_ll) {
                r.stats[r.VK] = v;
                const double ll = v / r.Nglobal;
                r.ll_hist[n] = ll;
                r.ctl->n_hist = n + 1;
                if (n + 1 - r.conv_base > 10 && fabs(prev - ll) / fabs(ll) < r.tol) { halt = 1; r.ctl->stop = 1; r.ctl->stop_iter = r.t - 1; }      // common.jl:53-56
            }
            if (!halt) r.ctl->t = r.t;
            r.ctl->ticket = 0;
        }
        MMM_RSTAMP(lane == 0, 11);
    }
}

// M-step of pass t from the (all-reduced) statistics, one wave per topic (no inter-block dependency: Elnbeta_k needs
// only the column sum of topic k): lambda = eta + sums, Elnbeta, exp table, beta (LDA.jl:96-112); block 0 then finalises
// ll_{t-1}, the stopping rule and the pass counter.
// Blocks of two waves: with the mailbox exchange folded in, both waves receive (V <= 128 entries in ONE polling round);
// wave 0 alone then runs the topic's M-step.
// P2P = false: the build without the mailbox code (its polling arrays live in scratch memory; a single-GPU launch carries none).
template <bool P2P>
__global__ __launch_bounds__(128) void k_lda_mstep(ReduceArgs r, int V, double eta, Ring lambda, Ring Elnbeta, Ring expElnbeta, Ring beta)
{
    const int stop = r.ctl->stop;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, c = r.t % 3;
    if (k == (int)gridDim.x - 1) {      // the extra block: pass tail, concurrent with the topic blocks (its loads are a dependent chain)
        if (!stop && tid < 64) lda_tail_block<P2P>(r, lane);
        return;
    }
    double* sums = r.stats + (size_t)k * V;
    if (P2P && r.p2p) {             // all-reduce folded in: own statistics + the peers', summed in rank order, written back for the passes below
        if (!stop) for (int v = tid; v < V; v += 128) sums[v] = p2p_recv_sum(r.px, r.p2p_seq, k * V + v, sums[v]);
        else if (k == 0 && tid == 0) (void)p2p_recv_sum(r.px, r.p2p_seq, 0, 0.0);      // no-op pass: the rendezvous of lda_reduce_block's dummy send
        __syncthreads();
    }
    // both waves form the column sum (same loads, same order: same bits, and no barrier); each then takes every other 64 entries
    double part = 0.0;
    for (int v = lane; v < V; v += 64) part += eta + sums[v];
    if (stop) return;
    const double cs = wave_sum(part);
    const double psi = dev_digamma_pos(cs);
    for (int v = tid; v < V; v += 128) {
        const double l = eta + sums[v];
        const double el = dev_digamma_pos(l) - psi;
        const size_t e = (size_t)k * V + v;
        lambda.s[c][e] = l; Elnbeta.s[c][e] = el; expElnbeta.s[c][e] = exp(el); beta.s[c][e] = l / cs;
    }
}

// k_lda_mstep for wide vocabularies: 512 threads per topic instead of one wave (V in the thousands), no folded exchange
__global__ __launch_bounds__(512) void k_lda_mstep_wide(ReduceArgs r, int V, double eta, Ring lambda, Ring Elnbeta, Ring expElnbeta, Ring beta)
{
    __shared__ double sh[16];
    const int stop = r.ctl->stop;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, c = r.t % 3;
    if (k == (int)gridDim.x - 1) {
        if (!stop && tid < 64) lda_tail_block<false>(r, lane);
        return;
    }
    if (stop) return;
    const double* sums = r.stats + (size_t)k * V;
    double part = 0.0;
    for (int v = tid; v < V; v += 512) part += eta + sums[v];
    part = wave_sum(part);
    if (lane == 0) sh[wid] = part;
    __syncthreads();
    double cs = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) cs += sh[w];
    const double psi = dev_digamma_pos(cs);
    for (int v = tid; v < V; v += 512) {
        const double l = eta + sums[v];
        const double el = dev_digamma_pos(l) - psi;
        const size_t e = (size_t)k * V + v;
        lambda.s[c][e] = l; Elnbeta.s[c][e] = el; expElnbeta.s[c][e] = exp(el); beta.s[c][e] = l / cs;
    }
}

// ---- ILDA (src/ILDA.jl): LDA whose topic-term distribution factorises over I features of the term, beta_kv = prod_i
// beta[i][f_vi, k].  The E-step, ll and ELBO document kernels run unchanged on EFFECTIVE V x K tables (Elnbeta_eff[v,k] =
// sum_i Elnbeta[i][f_vi, k], exp of it, beta_eff = prod_i beta[i][f_vi, k]); only the topic M-step differs: the V x K
// statistics are folded onto the feature values.  Model layout: lambda[i] is J_i x K column-major at K * sum_{q<i} J_q.

// mode 0: lambda = eta + folded sums (update_λ!, ILDA.jl:107-126); 1: from the stored lambda (update_Elnβ!/update_β!,
// :97-104,128-130); 2: effective tables only, from the stored Elnbeta / beta (after an upload).  One wave per topic.
__global__ __launch_bounds__(64 * kIldaMaxI) void k_ilda_mstep(IldaDesc ds, int mode, const double* sums, double* ilam, double* iEln, double* ibeta,
                                                   double* Eeff, double* expEeff, double* beff, const int* stop, int write_beta_only,
                                                   ReduceArgs tail, int with_tail)
{
    __shared__ double sE[kIldaMaxSJ], sB[kIldaMaxSJ];
    if (stop && *stop) return;
    // one wave per feature (launch: 64 * I threads): the features' folds and digamma chains run side by side
    const int k = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwv = blockDim.x >> 6, V = ds.V, K = ds.K;
    if (with_tail && k == K) {           // extra block of the fused pass: ll_{t-1}, stopping rule, pass counter (as k_lda_mstep)
        if (wid == 0) lda_tail_block<false>(tail, lane);      // (the ILDA exchange is never folded)
        return;
    }
    // V <= 256 (the 96 SNV contexts): the topic's statistics are fetched once, four per lane, and every masked sum below runs
    // out of registers (same lane assignment and order as the general loop, so the same bits)
    const bool in_regs = mode == 0 && V <= 256;
    double sv[4] = {0.0, 0.0, 0.0, 0.0};
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int v = lane + 64 * q; if (v < V) sv[q] = sums[(size_t)k * V + v]; }
    }
    for (int i = wid; i < ds.I; i += nwv) {
        const int Ji = ds.J[i];
        const size_t base = (size_t)K * ds.joff[i] + (size_t)Ji * k;
        const int* f = ds.features + (size_t)i * V;
        int fv[4] = {-1, -1, -1, -1};
        if (in_regs) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int v = lane + 64 * q; if (v < V) fv[q] = f[v]; }
        }
        double part = 0.0;
        for (int j0 = 0; j0 < Ji; j0 += 64) {
            const int j = j0 + lane;
            double l = 0.0;
            if (mode == 0) {
                // fold the topic's V statistics onto this feature's values: all 64 lanes walk the terms, one butterfly sum per
                // value (a lane-per-value loop over V global loads is a 60 us dependent chain)
                for (int jj = j0; jj < min(Ji, j0 + 64); ++jj) {
                    double t = 0.0;
                    if (in_regs) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) t += (fv[q] == jj) ? sv[q] : 0.0;
                    } else
                        for (int v = lane; v < V; v += 64) t += (f[v] == jj) ? sums[(size_t)k * V + v] : 0.0;
                    t = wave_sum(t);
                    if (lane == jj - j0) l = ds.eta[i] + t;
                }
                if (j < Ji) ilam[base + j] = l;
            } else if (j < Ji) l = ilam[base + j];
            part += l;
        }
        const double cs = wave_sum(part);
        const double psi = dev_digamma_pos(cs);
        for (int j0 = 0; j0 < Ji; j0 += 64) {
            const int j = j0 + lane;
            if (j < Ji) {
                double el, b;
                if (mode == 2) { el = iEln[base + j]; b = ibeta[base + j]; }
                else {
                    const double l = ilam[base + j];
                    el = dev_digamma_pos(l) - psi; b = l / cs;
                    if (!write_beta_only) iEln[base + j] = el;
                    ibeta[base + j] = b;
                }
                sE[ds.joff[i] + j] = el; sB[ds.joff[i] + j] = b;
            }
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        double e = 0.0, b = 1.0;
        for (int i = 0; i < ds.I; ++i) { const int j = ds.features[(size_t)i * V + v]; e += sE[ds.joff[i] + j]; b *= sB[ds.joff[i] + j]; }
        const size_t o = (size_t)k * V + v;
        if (!write_beta_only) { Eeff[o] = e; expEeff[o] = exp(e); }
        beff[o] = b;
    }
}

// Frozen-topic passes (transform / fit_heldout, LDA.jl:233-295): the E-step kernel runs with fixed tables and evaluates
// the ll of the SAME pass (theta_t and beta are both known); this kernel sums the per-block numerators (phase & 1), and
// (phase & 2) records ll_t, applies the stopping rule (LDA.jl:252 / :285) and advances the pass counter.
__global__ __launch_bounds__(64) void k_lda_infer_tail(ReduceArgs r, int phase)
{
    if (r.ctl->stop) return;
    const int lane = threadIdx.x;
    if (phase & 1) {
        double v = 0.0;
        for (int i = lane; i < r.nslab; i += 64) v += r.llpart[i];
        v = wave_sum(v);
        if (lane == 0) r.stats[r.VK] = v;
    }
    if ((phase & 2) && lane == 0) {
        const int n = r.ctl->n_hist;
        const double ll = r.stats[r.VK] / r.Nglobal;
        r.ll_hist[n] = ll;
        r.ctl->n_hist = n + 1;
        if (n + 1 - r.conv_base > 10) {
            const double prev = r.ll_hist[n - 1];
            if (fabs(prev - ll) / fabs(ll) < r.tol) { r.ctl->stop = 1; r.ctl->stop_iter = r.t; }
        }
        r.ctl->t = r.t;           // the state of the stopping pass is kept (its ll is not lagged)
    }
}

__global__ void k_lda_tail_only(ReduceArgs r) { if (!r.ctl->stop) lda_pass_tail(r); }

// ---- on-demand / stage kernels (reference-granularity entry points; not on the fused path) -----------------------
// phi = softmax_k(Elntheta + Elnbeta[v]) written to HBM (update_ϕ!, LDA.jl:69-76); one wave per document
// (TAB_LDS = false: vocabularies whose table does not fit LDS read it through L2; topics k >= K are then skipped, not padded)
template <int KP, bool TAB_LDS>
__global__ __launch_bounds__(kBlock) void k_lda_phi(LdaDev c, const double* Elntheta, const double* expElnbeta, double* phi)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int K = c.K, V = c.V;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const double* tab = TAB_LDS ? smem : expElnbeta;
    if (TAB_LDS) {
        for (int i = tid; i < KP * V; i += kBlock) smem[i] = (i < K * V) ? expElnbeta[i] : 0.0;
        __syncthreads();
    }
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        const double ak = (lane < K) ? exp(Elntheta[(size_t)d * K + lane]) : 0.0;
        double av[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) av[k] = wave_bcast(ak, k);
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int v = c.tc[start + w].x;
            double e[KP], s = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) { e[k] = (TAB_LDS || k < K) ? av[k] * tab[(size_t)k * V + v] : 0.0; s += e[k]; }
            double* ph = phi + (size_t)(start + w) * K;
#pragma unroll
            for (int k = 0; k < KP; ++k) if (k < K) ph[k] = e[k] / s;
        }
    }
}

// deterministic block sum (256 threads), result valid in every thread
__device__ __forceinline__ double block_sum_256(double v, double* sh)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// per topic k (one block per topic): lambda = eta + sums (if sums), Elnbeta / exp table (if Elnbeta), beta (if write_beta)
__global__ __launch_bounds__(256) void k_lda_topic(int V, double eta, const double* sums, double* lambda, double* Elnbeta,
                                                   double* expElnbeta, double* beta, int write_beta)
{
    __shared__ double sh[4];
    const int k = blockIdx.x;
    double part = 0.0;
    for (int v = threadIdx.x; v < V; v += 256) {
        double l = sums ? eta + sums[(size_t)k * V + v] : lambda[(size_t)k * V + v];
        if (sums) lambda[(size_t)k * V + v] = l;
        part += l;
    }
    const double cs = block_sum_256(part, sh);
    const double pcs = dev_digamma(cs);
    for (int v = threadIdx.x; v < V; v += 256) {
        const double l = lambda[(size_t)k * V + v];
        if (Elnbeta) {
            const double el = dev_digamma(l) - pcs;
            Elnbeta[(size_t)k * V + v] = el;
            expElnbeta[(size_t)k * V + v] = exp(el);
        }
        if (write_beta) beta[(size_t)k * V + v] = l / cs;
    }
}

__global__ void k_exp_table(int n, const double* in, double* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = exp(in[i]);
}

// theta = gamma / sum gamma (LDA.jl:92-94) and the log-likelihood numerator (LDA.jl:174-188), wave per document
template <int KP, bool TAB_LDS>
__global__ __launch_bounds__(kBlock) void k_lda_loglik(LdaDev c, const double* gamma, const double* beta, double* theta,
                                                       double* llpart, int compute_ll)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesPerBlock];
    const int K = c.K, V = c.V;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const double* tab = TAB_LDS ? smem : beta;
    if (TAB_LDS && compute_ll) {
        for (int i = tid; i < KP * V; i += kBlock) smem[i] = (i < K * V) ? beta[i] : 0.0;
        __syncthreads();
    }
    double wave_ll = 0.0;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        const double g = (lane < K) ? gamma[(size_t)d * K + lane] : 0.0;
        const double S = wave_sum(g);
        const double th = g / S;
        if (lane < K && theta) theta[(size_t)d * K + lane] = th;
        if (!compute_ll) continue;
        double tv[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) tv[k] = wave_bcast(th, k);
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        double acc = 0.0;
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int2 t = c.tc[start + w];
            double p = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) if (TAB_LDS || k < K) p = fma(tv[k], tab[(size_t)k * V + t.x], p);
            acc += (double)t.y * log(p);
        }
        wave_ll += wave_sum(acc);
    }
    if (compute_ll) {
        if (lane == 0) shw[wid] = wave_ll;
        __syncthreads();
        if (tid == 0) llpart[blockIdx.x] = shw[0] + shw[1] + shw[2] + shw[3];
    }
}

// ---- wide vocabularies: K*V tables that do not fit LDS next to a slab (e.g. 1536 pentanucleotide contexts, or any V in the
// thousands).  Same pass structure and rings as the fused path, different data flow: the per-block statistics partials
// (grid x K*V doubles) are out of the question here, and so is streaming phi (K*nnz doubles) out and back in a different
// order.  Instead every phi_kw is evaluated TWICE, in two sweeps that each read 8 B per nonzero:
//   * document-major (k_lda_estep_wide, one wave per document, table rows gathered through L2 from term-major copies): Elntheta_t,
//     a_d = exp(Elntheta_t) -> `aexp` (D x KP), gamma_{t+1} = alpha + sum_w phi_t n, and (do_ll) the ll numerator of pass t-1;
//   * term-major (k_lda_stats_terms, one block per term over a posting list (doc, count) built at create): with the term's
//     table column in scalar registers and a_d read from the L2-resident D x KP array (one contiguous row per posting), stats[k][v] = sum_postings
//     n a_dk eB_kv / (sum_k' a_dk' eB_k'v) in posting order -- a fixed summation order, no atomics.
// term-major copies of the two tables the document sweep gathers from ([v][KP], zero-padded): a lane then reads its term's
// K values as one contiguous run (2-3 sectors) instead of K sectors V doubles apart -- 8x fewer L2 requests when documents
// are sparse in the vocabulary, the same number when they are dense
__global__ void k_lda_tables_by_term(int V, int K, int KP, const double* __restrict__ eB, const double* __restrict__ beta,
                                     double* __restrict__ eBT, double* __restrict__ betaT)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)V * KP) return;
    const int v = (int)(i / KP), k = (int)(i % KP);
    eBT[i] = (k < K) ? eB[(size_t)k * V + v] : 0.0;
    if (beta) betaT[i] = (k < K) ? beta[(size_t)k * V + v] : 0.0;
}

template <int KP>
__global__ __launch_bounds__(kBlock) void k_lda_estep_wide(EstepArgs a, double* __restrict__ aexp, const double* __restrict__ eBT,
                                                           const double* __restrict__ betaT)
{
    __shared__ double shw[kWavesPerBlock];
    if (a.ctl->stop) return;
    const int t = a.t;
    const double* __restrict__ gam = a.gamma.s[t % 3];
    const double* __restrict__ gprev = a.gamma.s[(t + 2) % 3];
    double* __restrict__ gnext = a.gamma.s[(t + 1) % 3];
    double* __restrict__ Eln = a.Elntheta.s[t % 3];
    const int K = a.c.K, D = a.c.D;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double wave_ll = 0.0;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < D; d += gridDim.x * kWavesPerBlock) {
        const double gk = (lane < K) ? gam[(size_t)d * K + lane] : 0.0;
        const double S = wave_sum(gk);
        const double ps = dev_digamma_pos(lane < K ? gk : S);          // lanes >= K hold psi(S)
        const double el = ps - wave_bcast(ps, K);
        const double ak = (lane < K) ? exp(el) : 0.0;
        if (lane < K) Eln[(size_t)d * K + lane] = el;
        if (lane < KP) aexp[(size_t)d * KP + lane] = ak;      // D x KP rows, zero-padded
        double th = 0.0;
        if (a.do_ll) {
            const double gp = (lane < K) ? gprev[(size_t)d * K + lane] : 0.0;
            th = gp / wave_sum(gp);
        }
        double av[KP], tv[KP], acc[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) { av[k] = wave_readlane(ak, k); tv[k] = wave_readlane(th, k); acc[k] = 0.0; }      // scalar registers
        const int64_t start = a.c.doc_ptr[d];
        const int W = (int)(a.c.doc_ptr[d + 1] - start);
        double ll = 0.0;
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int2 tc = a.c.tc[start + w];
            const double n = (double)tc.y;
            const double2* __restrict__ col = (const double2*)(eBT + (size_t)tc.x * KP);       // KP is even: 16-byte aligned
            double e[KP], s = 0.0;
#pragma unroll
            for (int k = 0; k < KP; k += 2) {
                const double2 b = col[k / 2];
                e[k] = av[k] * b.x; e[k + 1] = av[k + 1] * b.y;          // padded topics: 0 * 0
                s += e[k]; s += e[k + 1];
            }
            const double rn = n / s;
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[k] = fma(e[k], rn, acc[k]);
            if (a.do_ll) {
                const double2* __restrict__ bc = (const double2*)(betaT + (size_t)tc.x * KP);
                double p = 0.0;
#pragma unroll
                for (int k = 0; k < KP; k += 2) { const double2 b = bc[k / 2]; p = fma(tv[k], b.x, p); p = fma(tv[k + 1], b.y, p); }
                ll = fma(n, log(p), ll);
            }
        }
        double mine = 0.0;
#pragma unroll
        for (int k = 0; k < KP; ++k) { const double tot = wave_sum(acc[k]); if (lane == k) mine = tot; }
        if (lane < K) gnext[(size_t)d * K + lane] = a.c.alpha + mine;
        if (a.do_ll) wave_ll += wave_sum(ll);
    }
    if (a.do_ll) {
        if (lane == 0) shw[wid] = wave_ll;
        __syncthreads();
        if (threadIdx.x == 0) a.llpart[blockIdx.x] = shw[0] + shw[1] + shw[2] + shw[3];
    }
}

// ---- more than 32 topics (round 3): the two sweeps of the wide path with the topic loops ROLLED -- runtime K <= 256, KP = K rounded up to
// even -- so that no build per K is needed and nothing goes to scratch: a_k / theta_k of the wave's document sit in LDS (broadcast reads),
// and the sums over a document's nonzeros (gamma_{t+1,k}) resp. over a term's postings (the lambda statistics) are kept as one LDS column
// per lane and topic and added up across the lanes at the end.  Where a document's topics are spread over the lanes (the Elntheta
// prologue, the final sums) lane l holds topics l, l + 64, l + 128, l + 192 (kLdaSlots).  The reference has no limit on K (LDA.jl:24-54);
// here it is the LDS column block: 512 K bytes per wave, one wave per block from K = 129.
// LDS per wave: [KP][64] column sums | [KP] a_k | [KP] theta_k.
constexpr int kLdaSlots = 4;

__global__ __launch_bounds__(kBlock) void k_lda_estep_big(EstepArgs a, double* __restrict__ aexp, const double* __restrict__ eBT,
                                                          const double* __restrict__ betaT, int KP)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesPerBlock];
    if (a.ctl->stop) return;
    const int t = a.t;
    const double* __restrict__ gam = a.gamma.s[t % 3];
    const double* __restrict__ gprev = a.gamma.s[(t + 2) % 3];
    double* __restrict__ gnext = a.gamma.s[(t + 1) % 3];
    double* __restrict__ Eln = a.Elntheta.s[t % 3];
    const int K = a.c.K, D = a.c.D;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, NW = blockDim.x >> 6;
    double* wacc = smem + (size_t)wid * ((size_t)KP * MMM_WAVE + 2 * KP);
    double* wav = wacc + (size_t)KP * MMM_WAVE;
    double* wtv = wav + KP;
    double wave_ll = 0.0;
    for (int d = blockIdx.x * NW + wid; d < D; d += gridDim.x * NW) {
        double gk[kLdaSlots], gp[kLdaSlots], gsum = 0.0, psum = 0.0;
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) {
            const int k = lane + 64 * s;
            gk[s] = (k < K) ? gam[(size_t)d * K + k] : 0.0;
            gp[s] = (a.do_ll && k < K) ? gprev[(size_t)d * K + k] : 0.0;
            gsum += gk[s]; psum += gp[s];
        }
        const double psS = dev_digamma_pos(wave_sum(gsum));             // Elntheta (LDA.jl:78-80)
        const double Sp = a.do_ll ? wave_sum(psum) : 1.0;
        lds_wave_sync();
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) {
            const int k = lane + 64 * s;
            if (k < KP) {
                const double el = (k < K) ? dev_digamma_pos(gk[s]) - psS : 0.0;
                const double ak = (k < K) ? exp(el) : 0.0;
                if (k < K) Eln[(size_t)d * K + k] = el;
                aexp[(size_t)d * KP + k] = ak;      // D x KP rows, zero-padded
                wav[k] = ak; wtv[k] = gp[s] / Sp;
            }
        }
        for (int k = 0; k < KP; ++k) wacc[(size_t)k * MMM_WAVE + lane] = 0.0;
        lds_wave_sync();
        const int64_t start = a.c.doc_ptr[d];
        const int W = (int)(a.c.doc_ptr[d + 1] - start);
        double ll = 0.0;
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int2 tc = a.c.tc[start + w];
            const double n = (double)tc.y;
            const double2* __restrict__ col = (const double2*)(eBT + (size_t)tc.x * KP);       // KP is even: 16-byte aligned
            double s = 0.0;
            for (int k = 0; k < KP; k += 2) {
                const double2 b = col[k / 2];
                s += wav[k] * b.x; s += wav[k + 1] * b.y;          // padded topics: 0 * 0
            }
            const double rn = n / s;
            for (int k = 0; k < KP; k += 2) {
                const double2 b = col[k / 2];
                double* c0 = wacc + (size_t)k * MMM_WAVE + lane;
                c0[0] = fma(wav[k] * b.x, rn, c0[0]);
                c0[MMM_WAVE] = fma(wav[k + 1] * b.y, rn, c0[MMM_WAVE]);
            }
            if (a.do_ll) {
                const double2* __restrict__ bc = (const double2*)(betaT + (size_t)tc.x * KP);
                double p = 0.0;
                for (int k = 0; k < KP; k += 2) { const double2 b = bc[k / 2]; p = fma(wtv[k], b.x, p); p = fma(wtv[k + 1], b.y, p); }
                ll = fma(n, log(p), ll);
            }
        }
        lds_wave_sync();
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) {       // lane l adds the 64 column sums of its topics, starting at column l (rotated: the lanes stay on different LDS banks)
            const int k = lane + 64 * s;
            if (k < K) {
                const double* row = wacc + (size_t)k * MMM_WAVE;
                double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
                for (int j = 0; j < MMM_WAVE; j += 4) {
                    r0 += row[(j + lane) & 63]; r1 += row[(j + 1 + lane) & 63]; r2 += row[(j + 2 + lane) & 63]; r3 += row[(j + 3 + lane) & 63];
                }
                gnext[(size_t)d * K + k] = a.c.alpha + ((r0 + r1) + (r2 + r3));
            }
        }
        if (a.do_ll) wave_ll += wave_sum(ll);
    }
    if (a.do_ll) {
        if (lane == 0) shw[wid] = wave_ll;
        __syncthreads();
        if (threadIdx.x == 0) { double s = 0.0; for (int w = 0; w < NW; ++w) s += shw[w]; a.llpart[blockIdx.x] = s; }
    }
}

// the term-major sweep (k_lda_stats_terms) with rolled topic loops: block v < V = term v, at most 4 waves (one from K = 129), each over a
// contiguous segment of the term's postings; LDS: [waves][KP][64] column sums | [KP] the term's table column | [waves][KP] segment sums.
// Block V: the ll partials.
__global__ __launch_bounds__(256) void k_lda_stats_big(int V, int K, int KP, const int64_t* __restrict__ term_ptr, const int2* __restrict__ tpost,
                                                       const double* __restrict__ aexp, const double* __restrict__ eB, ReduceArgs r)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (r.ctl->stop) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int v = blockIdx.x;
    if (v == V) {
        if (wid == 0) {
            double s = 0.0;
            for (int i = lane; i < r.nslab; i += MMM_WAVE) s += r.llpart[i];
            s = wave_sum(s);
            if (lane == 0) r.stats[r.VK] = s;
        }
        return;
    }
    double* wacc = smem + (size_t)wid * KP * MMM_WAVE;
    double* seb = smem + (size_t)nw * KP * MMM_WAVE;
    double* sh = seb + KP;
    for (int k = threadIdx.x; k < KP; k += blockDim.x) seb[k] = (k < K) ? eB[(size_t)k * V + v] : 0.0;
    for (int k = 0; k < KP; ++k) wacc[(size_t)k * MMM_WAVE + lane] = 0.0;
    __syncthreads();
    const int64_t p0 = term_ptr[v], p1 = term_ptr[v + 1];
    const int64_t seg = (p1 - p0 + nw - 1) / nw;
    const int64_t q0 = p0 + wid * seg, q1 = (q0 + seg < p1) ? q0 + seg : p1;
    for (int64_t j = q0 + lane; j < q1; j += MMM_WAVE) {
        const int2 dn = tpost[j];
        const double2* __restrict__ ad = (const double2*)(aexp + (size_t)dn.x * KP);      // one contiguous run per posting
        double s = 0.0;
        for (int k = 0; k < KP; k += 2) { const double2 x = ad[k / 2]; s += x.x * seb[k]; s += x.y * seb[k + 1]; }
        const double rn = (double)dn.y / s;
        for (int k = 0; k < KP; k += 2) {
            const double2 x = ad[k / 2];
            double* c0 = wacc + (size_t)k * MMM_WAVE + lane;
            c0[0] = fma(x.x * seb[k], rn, c0[0]);
            c0[MMM_WAVE] = fma(x.y * seb[k + 1], rn, c0[MMM_WAVE]);
        }
    }
    lds_wave_sync();
    for (int k = lane; k < KP; k += MMM_WAVE) {
        const double* row = wacc + (size_t)k * MMM_WAVE;
        double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
        for (int j = 0; j < MMM_WAVE; j += 4) {
            r0 += row[(j + lane) & 63]; r1 += row[(j + 1 + lane) & 63]; r2 += row[(j + 2 + lane) & 63]; r3 += row[(j + 3 + lane) & 63];
        }
        sh[(size_t)wid * KP + k] = (r0 + r1) + (r2 + r3);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        double tot = 0.0;
        for (int w = 0; w < nw; ++w) tot += sh[(size_t)w * KP + k];
        r.stats[(size_t)k * V + v] = tot;
    }
}

// ---- more than 64 topics: the per-document kernels that give every topic a lane, with lane l holding topics l + 64 s -------------------
// gamma[:,d] = alpha + phi[d] * n_d (LDA.jl:83-87) from a resident phi, then Elntheta (if asked)
__global__ __launch_bounds__(kBlock) void k_lda_gamma_from_phi_big(LdaDev c, const double* phi, double* gamma, double* Elntheta)
{
    const int K = c.K;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        double mine[kLdaSlots] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
            for (int w = lane; w < W; w += MMM_WAVE) acc += phi[(size_t)(start + w) * K + k] * (double)c.tc[start + w].y;
            acc = wave_sum(acc);
#pragma unroll
            for (int s = 0; s < kLdaSlots; ++s) if (k == lane + 64 * s) mine[s] = acc;
        }
        double g[kLdaSlots], gs = 0.0;
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) {
            const int k = lane + 64 * s;
            g[s] = (k < K) ? c.alpha + mine[s] : 0.0;
            if (k < K) gamma[(size_t)d * K + k] = g[s];
            gs += g[s];
        }
        if (Elntheta) {
            const double psS = dev_digamma(wave_sum(gs));
#pragma unroll
            for (int s = 0; s < kLdaSlots; ++s) { const int k = lane + 64 * s; if (k < K) Elntheta[(size_t)d * K + k] = dev_digamma(g[s]) - psS; }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_lda_Elntheta_big(LdaDev c, const double* gamma, double* Elntheta)
{
    const int K = c.K;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        double g[kLdaSlots], gs = 0.0;
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) { const int k = lane + 64 * s; g[s] = (k < K) ? gamma[(size_t)d * K + k] : 0.0; gs += g[s]; }
        const double psS = dev_digamma(wave_sum(gs));
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) { const int k = lane + 64 * s; if (k < K) Elntheta[(size_t)d * K + k] = dev_digamma(g[s]) - psS; }
    }
}

// phi (LDA.jl:69-76) with a_k in LDS and rolled topic loops; dynamic LDS: [waves][K]
__global__ __launch_bounds__(kBlock) void k_lda_phi_big(LdaDev c, const double* Elntheta, const double* expElnbeta, double* phi)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int K = c.K, V = c.V;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double* wav = smem + (size_t)wid * K;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        lds_wave_sync();
        for (int k = lane; k < K; k += MMM_WAVE) wav[k] = exp(Elntheta[(size_t)d * K + k]);
        lds_wave_sync();
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int v = c.tc[start + w].x;
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += wav[k] * expElnbeta[(size_t)k * V + v];
            double* ph = phi + (size_t)(start + w) * K;
            for (int k = 0; k < K; ++k) ph[k] = wav[k] * expElnbeta[(size_t)k * V + v] / s;
        }
    }
}

// theta = gamma / sum gamma (LDA.jl:92-94) and the log-likelihood numerator (LDA.jl:174-188); dynamic LDS: [waves][K]
__global__ __launch_bounds__(kBlock) void k_lda_loglik_big(LdaDev c, const double* gamma, const double* beta, double* theta, double* llpart, int compute_ll)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesPerBlock];
    const int K = c.K, V = c.V;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double* wth = smem + (size_t)wid * K;
    double wave_ll = 0.0;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        double g[kLdaSlots], gs = 0.0;
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) { const int k = lane + 64 * s; g[s] = (k < K) ? gamma[(size_t)d * K + k] : 0.0; gs += g[s]; }
        const double S = wave_sum(gs);
        lds_wave_sync();
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) {
            const int k = lane + 64 * s;
            if (k < K) { const double th = g[s] / S; wth[k] = th; if (theta) theta[(size_t)d * K + k] = th; }
        }
        lds_wave_sync();
        if (!compute_ll) continue;
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        double acc = 0.0;
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int2 t = c.tc[start + w];
            double p = 0.0;
            for (int k = 0; k < K; ++k) p = fma(wth[k], beta[(size_t)k * V + t.x], p);
            acc += (double)t.y * log(p);
        }
        wave_ll += wave_sum(acc);
    }
    if (compute_ll) {
        if (lane == 0) shw[wid] = wave_ll;
        __syncthreads();
        if (tid == 0) llpart[blockIdx.x] = shw[0] + shw[1] + shw[2] + shw[3];
    }
}

// block v < V: the statistics of term v (LDA.jl:103-105), its postings split into blockDim.x / 64 contiguous segments, one per
// wave, lanes over a segment's postings in order, segment sums added in segment order.  Block V: the E-step's ll partials
// summed into stats[V*K] (what lda_reduce_block's wave 1 does).
template <int KP>
__global__ __launch_bounds__(512) void k_lda_stats_terms(int V, int K, const int64_t* __restrict__ term_ptr, const int2* __restrict__ tpost,
                                                          const double* __restrict__ aexp, const double* __restrict__ eB, ReduceArgs r)
{
    __shared__ double sh[8][KP];
    if (r.ctl->stop) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int v = blockIdx.x;
    if (v == V) {
        if (wid == 0) {
            double s = 0.0;
            for (int i = lane; i < r.nslab; i += MMM_WAVE) s += r.llpart[i];
            s = wave_sum(s);
            if (lane == 0) r.stats[r.VK] = s;
        }
        return;
    }
    double eb[KP], acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { eb[k] = (k < K) ? eB[(size_t)k * V + v] : 0.0; acc[k] = 0.0; }
    const int64_t p0 = term_ptr[v], p1 = term_ptr[v + 1];
    const int64_t seg = (p1 - p0 + nw - 1) / nw;
    const int64_t q0 = p0 + wid * seg, q1 = (q0 + seg < p1) ? q0 + seg : p1;
    for (int64_t j = q0 + lane; j < q1; j += MMM_WAVE) {
        const int2 dn = tpost[j];
        const double2* __restrict__ ad = (const double2*)(aexp + (size_t)dn.x * KP);      // one contiguous run per posting
        double e[KP], s = 0.0;
#pragma unroll
        for (int k = 0; k < KP; k += 2) { const double2 x = ad[k / 2]; e[k] = x.x * eb[k]; e[k + 1] = x.y * eb[k + 1]; s += e[k]; s += e[k + 1]; }
        const double rn = (double)dn.y / s;
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] = fma(e[k], rn, acc[k]);
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) { const double tot = wave_sum(acc[k]); if (lane == 0) sh[wid][k] = tot; }
    __syncthreads();
    if ((int)threadIdx.x < K) {
        double tot = 0.0;
        for (int w = 0; w < nw; ++w) tot += sh[w][threadIdx.x];
        r.stats[(size_t)threadIdx.x * V + v] = tot;
    }
}

// out[j] = sum_i part[i*stride + j], j < gridDim.x  (one wave per j)
__global__ __launch_bounds__(64) void k_sum_columns(const double* part, int n, int stride, double* out)
{
    const int j = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += part[(size_t)i * stride + j];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[j] = acc;
}

// push ll = num/N onto the device history (standalone ll of the last pass)
__global__ void k_ll_push(LdaCtl* ctl, const double* num, double N, double* hist, double* also)
{
    const double ll = *num / N;
    if (hist) { hist[ctl->n_hist] = ll; ctl->n_hist += 1; }
    if (also) *also = ll;
}

__global__ void k_ctl_clear_stop(LdaCtl* ctl) { ctl->stop = 0; ctl->stop_iter = 0; ctl->ticket = 0; }

// gamma[:,d] = alpha + phi[d] * n_d (LDA.jl:83-87) from a resident phi, then Elntheta (if asked)
__global__ __launch_bounds__(kBlock) void k_lda_gamma_from_phi(LdaDev c, const double* phi, double* gamma, double* Elntheta)
{
    const int K = c.K;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        double mine = 0.0;
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
            for (int w = lane; w < W; w += MMM_WAVE) acc += phi[(size_t)(start + w) * K + k] * (double)c.tc[start + w].y;
            acc = wave_sum(acc);
            if (lane == k) mine = acc;
        }
        const double g = (lane < K) ? c.alpha + mine : 0.0;
        if (lane < K) gamma[(size_t)d * K + lane] = g;
        if (Elntheta) {
            const double S = wave_sum(g);
            const double ps = dev_digamma(lane < K ? g : S);
            const double el = ps - (K < MMM_WAVE ? wave_bcast(ps, K) : dev_digamma(S));
            if (lane < K) Elntheta[(size_t)d * K + lane] = el;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_lda_Elntheta(LdaDev c, const double* gamma, double* Elntheta)
{
    const int K = c.K;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        const double g = (lane < K) ? gamma[(size_t)d * K + lane] : 0.0;
        const double S = wave_sum(g);
        const double ps = dev_digamma(lane < K ? g : S);
        const double el = ps - (K < MMM_WAVE ? wave_bcast(ps, K) : dev_digamma(S));
        if (lane < K) Elntheta[(size_t)d * K + lane] = el;
    }
}

// sums[k][v] += phi[k,w] n_w (LDA.jl:103-105) from a resident phi; global f64 atomics (stage API only)
__global__ void k_lda_lambda_from_phi(LdaDev c, int64_t nnz, const double* phi, double* sums)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int2 t = c.tc[e];
    for (int k = 0; k < c.K; ++k) unsafeAtomicAdd(&sums[(size_t)k * c.V + t.x], phi[(size_t)e * c.K + k] * (double)t.y);
}

// per-document ELBO pieces (LDA.jl:120-160): out[block][5] = {sum Elntheta, ElnPZ, ElnPX, ElnQZ, ElnQtheta}
__global__ __launch_bounds__(kBlock) void k_lda_elbo_docs(LdaDev c, const double* phi, const double* gamma, const double* Elntheta,
                                                          const double* Elnbeta, double* out)
{
    __shared__ double shw[kWavesPerBlock][5];
    const int K = c.K, V = c.V;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double t[5] = {0, 0, 0, 0, 0};
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        const double g = (lane < K) ? gamma[(size_t)d * K + lane] : 0.0;
        const double el = (lane < K) ? Elntheta[(size_t)d * K + lane] : 0.0;
        const double S = wave_sum(g);
        t[0] += wave_sum(el);
        // ElnQtheta = sum lgamma(gamma) - lgamma(sum gamma) - sum (gamma-1) Elntheta   (LDA.jl:148-152)
        t[4] += wave_sum(lane < K ? lgamma(g) - (g - 1.0) * el : 0.0) - lgamma(S);
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        double pz = 0.0, px = 0.0, qz = 0.0;
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int2 tc = c.tc[start + w];
            const double n = (double)tc.y;
            for (int k = 0; k < K; ++k) {
                const double p = phi[(size_t)(start + w) * K + k];
                pz += p * Elntheta[(size_t)d * K + k] * n;
                px += p * Elnbeta[(size_t)k * V + tc.x] * n;
                qz += dev_xlogx(p);
            }
        }
        t[1] += wave_sum(pz); t[2] += wave_sum(px); t[3] += wave_sum(qz);
    }
    if (lane == 0) for (int j = 0; j < 5; ++j) shw[wid][j] = t[j];
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) s += shw[w][threadIdx.x];
        out[(size_t)blockIdx.x * 5 + threadIdx.x] = s;
    }
}

// the same for more than 64 topics: lane l holds topics l + 64 s
__global__ __launch_bounds__(kBlock) void k_lda_elbo_docs_big(LdaDev c, const double* phi, const double* gamma, const double* Elntheta,
                                                              const double* Elnbeta, double* out)
{
    __shared__ double shw[kWavesPerBlock][5];
    const int K = c.K, V = c.V;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double t[5] = {0, 0, 0, 0, 0};
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        double gs = 0.0, es = 0.0, qs = 0.0;
#pragma unroll
        for (int s = 0; s < kLdaSlots; ++s) {
            const int k = lane + 64 * s;
            if (k < K) {
                const double g = gamma[(size_t)d * K + k], el = Elntheta[(size_t)d * K + k];
                gs += g; es += el; qs += lgamma(g) - (g - 1.0) * el;
            }
        }
        const double S = wave_sum(gs);
        t[0] += wave_sum(es);
        t[4] += wave_sum(qs) - lgamma(S);
        const int64_t start = c.doc_ptr[d];
        const int W = (int)(c.doc_ptr[d + 1] - start);
        double pz = 0.0, px = 0.0, qz = 0.0;
        for (int w = lane; w < W; w += MMM_WAVE) {
            const int2 tc = c.tc[start + w];
            const double n = (double)tc.y;
            for (int k = 0; k < K; ++k) {
                const double p = phi[(size_t)(start + w) * K + k];
                pz += p * Elntheta[(size_t)d * K + k] * n;
                px += p * Elnbeta[(size_t)k * V + tc.x] * n;
                qz += dev_xlogx(p);
            }
        }
        t[1] += wave_sum(pz); t[2] += wave_sum(px); t[3] += wave_sum(qz);
    }
    if (lane == 0) for (int j = 0; j < 5; ++j) shw[wid][j] = t[j];
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) s += shw[w][threadIdx.x];
        out[(size_t)blockIdx.x * 5 + threadIdx.x] = s;
    }
}

// topic-side ELBO pieces (LDA.jl:114-118,142-146): out = {sum Elnbeta, ElnQbeta}
__global__ __launch_bounds__(256) void k_lda_elbo_topics(int V, int K, const double* lambda, const double* Elnbeta, double* out)
{
    __shared__ double sh[4];
    double sE = 0.0, q = 0.0;
    for (int k = 0; k < K; ++k) {
        double cs = 0.0, a = 0.0;
        for (int v = threadIdx.x; v < V; v += 256) {
            const double l = lambda[(size_t)k * V + v], e = Elnbeta[(size_t)k * V + v];
            cs += l; a += lgamma(l) - (l - 1.0) * e; sE += e;
        }
        cs = block_sum_256(cs, sh);
        q += block_sum_256(a, sh) - lgamma(cs);
    }
    sE = block_sum_256(sE, sh);
    if (threadIdx.x == 0) { out[0] = sE; out[1] = q; }
}

// topic-side ELBO pieces of ILDA: out[0] = sum_i (eta_i - 1) sum Elnbeta[i]  (ElnPβ without its constant, ILDA.jl:132-141);
// out[1] = ElnQβ as the reference computes it -- `lnq =` inside the loop (ILDA.jl:175-182) keeps only the LAST feature.
__global__ __launch_bounds__(256) void k_ilda_elbo_topics(IldaDesc ds, const double* ilam, const double* iEln, double* out)
{
    __shared__ double sh[4];
    double p = 0.0, q = 0.0;
    for (int i = 0; i < ds.I; ++i) {
        const int Ji = ds.J[i];
        double qi = 0.0, pe = 0.0;
        for (int k = 0; k < ds.K; ++k) {
            const size_t base = (size_t)ds.K * ds.joff[i] + (size_t)Ji * k;
            double cs = 0.0, a = 0.0;
            for (int j = threadIdx.x; j < Ji; j += 256) {
                const double l = ilam[base + j], e = iEln[base + j];
                cs += l; a += lgamma(l) - (l - 1.0) * e; pe += e;
            }
            cs = block_sum_256(cs, sh);
            qi += block_sum_256(a, sh) - lgamma(cs);
        }
        p += (ds.eta[i] - 1.0) * block_sum_256(pe, sh);
        q = qi;
    }
    if (threadIdx.x == 0) { out[0] = p; out[1] = q; }
}

__global__ void k_fill(double* p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_doc_counts(LdaDev c, double* out)
{
    double acc = 0.0;
    const int64_t nnz = c.doc_ptr[c.D];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) acc += (double)c.tc[e].y;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(out, acc);
}

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
            const int nred_ = (r.VK + 15) / 16;
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
            ms.ll_join = (!mmm_off(m->tune, MMM_OFF_LDA_LL_JOIN) && r.n_ll > 0 && (int64_t)m->D > (int64_t)r.n_ll * docs_per_ll_block && r.n_ll + nred - 1 <= 512) ? 1 : 0;
            ms.n_ll = r.n_ll;
            // The next pass's prologue beside this pass's reduction, in the ll blocks: single-step build (every wave of the E-step kernel walks
            // its chain once, the prologue is 2 us of it), every document in exactly one ll block's single step, plain LDA
            ms.pro = (m->single_step && !m->ilda && m->KP <= 12 && m->L == 16 && m->aexp_next.p && r.n_ll > 0 && !ms.ll_join &&
                      (int64_t)r.n_ll * docs_per_ll_block >= (int64_t)m->D && !mmm_off(m->tune, MMM_OFF_LDA_EARLY_PROLOGUE)) ? 1 : 0;
            ms.pro_gamma = m->gamma[(t + 1) % 3].p; ms.pro_Eln = m->Elntheta[(t + 1) % 3].p; ms.pro_a = m->aexp_next.p;
            const int c3 = t % 3;
            IldaMerge im{};
            if (m->ilda) im = IldaMerge{m->ids, m->ilam[c3].p, m->iEln[c3].p, m->ibeta[c3].p, m->fcells.p};
            MMM_KP_SWITCH(m, {
                auto k = m->ilda ? k_lda_reduce_ll_mstep<KPV, false, true> : (r.p2p ? k_lda_reduce_ll_mstep<KPV, true, false> : k_lda_reduce_ll_mstep<KPV, false, false>);
                const int ai = m->ilda ? 2 : r.p2p;
                if (!m->attr_mm[ai]) { if ((rc = set_lds(ctx, k, lds))) return rc; m->attr_mm[ai] = true; }
                hipLaunchKernelGGL(k, dim3(nred + r.n_ll), dim3(16, 64), lds, ctx->stream, r, m->dev(), m->gamma[(t + 2) % 3].p, m->beta[(t + 2) % 3].p, ms, im);
            })
            MMM_LAUNCH_CHECK(ctx);
            m->aexp_for = ms.pro ? t + 1 : -1;
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
extern "C" int mmm_diag_red_stamps(unsigned long long out[32])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_red_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
}
#endif

extern "C" {

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
    const bool small = !dense && (V <= 96) && KP <= 12 && ((D + 12 * G - 1) / (12 * G) <= ncu) && ctx->tune.grid_blocks == 0 &&
                       ctx->tune.waves_per_block == 0;
    // single-step build: just enough waves per block to cover the corpus with one block per CU (fewer co-resident waves
    // per SIMD = shorter step)
    const int swaves = std::max(4, std::min(12, (D + G * ncu - 1) / (G * ncu)));
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
    m->single_step = !wide && small && (int64_t)m->waves_e * G * ncu >= D;
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
