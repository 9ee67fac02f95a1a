// ctm_mstep.cuh -- device code of everything after the E-step of ctm.hip (included there, inside its anonymous namespace): the per-document
// probes of the stage API, gamma statistics and moments, the Gaussian and topic M-steps, update_alpha, props and log-likelihoods, the ELBO
// kernels.  MMCTM.jl:200-448, IMMCTM.jl:188-385, common.jl:38-56.
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
    int gauss_wide;         // 1: keep the three-barrier inversion (block_inverse_wide) where the pipelined one would run (MMM_OFF_CTM_PIPE_GAUSS: A/B, tests)
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

#ifdef MMM_DIAG_STAMPS
// diagnostic build only (make diag): s_memtime stamps of thread 0 through the Gaussian M-step -- [0] entry, [1] mu stored, [2] Sigma filled,
// [3 + 3c .. 5 + 3c] column c: after the pivot barrier, after the scale barrier, after the eliminate barrier, [90] inverse done,
// [91] invSigma stored; [94] / [95] s_memrealtime (100 MHz) at entry / exit
__device__ unsigned long long g_gauss_stamps[96];
#define MMM_GSTAMP(i)                                                                                        \
    do {                                                                                                     \
        unsigned long long t_;                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (threadIdx.x == 0 && blockIdx.y == 0 && (i) < 94) g_gauss_stamps[i] = t_;                        \
    } while (0)
#else
#define MMM_GSTAMP(i) do { } while (0)
#endif

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
            // an exactly zero pivot column is LAPACK's SingularException upstream (`inv(Σ)`, MMCTM.jl:211); a NaN one is not -- the reference
            // carries on with a NaN inverse, and so does this (counted: mmm_ctm_events)
            if (tid == 0) { s_best[c] = best; if (best == 0.0) *singular = 1; }
        }
        __syncthreads();
        MMM_GSTAMP(3 + 3 * c);
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
        MMM_GSTAMP(4 + 3 * c);
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
        MMM_GSTAMP(5 + 3 * c);
    }
    if (tid < n) s_col[tid] = log(s_best[tid]);
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int c = 0; c < n; ++c) s += s_col[c]; *logdet = s; }
}

// Round 5, n <= 32: the same elimination with ONE block barrier per column instead of three (35-45 us -> see DESIGN section 4.2 at n = 28).
// The matrix is the augmented [A | A^-1] of n rows x W = 2n columns and is DOUBLE-BUFFERED: the sweep of column c reads buffer `cur` and
// writes `nxt`, so nothing waits between reading the old rows and writing the new ones.  Every update thread forms the scaled pivot entry
// of its own column itself (low / piv: the same division, no hand-off through LDS).  Wave 0 does not sweep: while the other waves
// eliminate column c it forms column c + 1 AS THAT SWEEP LEAVES IT (same operands, same operations: the same bits) and finds its pivot
// (largest magnitude by DPP, lowest row by ballot -- the rule above), leaving pivot row, pivot and A[c+1][c+1] in cells of the other parity.
// Per element the operations are exactly those of block_inverse_wide (and of the order-matched CPU restatement, orc_twin_gauss).
// Block of >= 128 threads; M0 holds the augmented matrix on entry; returns the buffer that holds it on exit (A^-1 in columns n .. 2n-1).
__device__ const double* block_inverse_pipelined(int n, double* M0, double* M1, double* logdet, int* singular)
{
    __shared__ double s_best[64], s_lg[64], s_pv[2][2];
    __shared__ int s_p[2];
    const int tid = threadIdx.x, nt = blockDim.x, W = 2 * n;
    const int u = tid - 64, nu = nt - 64;
    const int j0 = u >= 0 ? u % W : 0, r0 = u >= 0 ? u / W : 0, rstep = nu / W;      // update thread -> column j0, rows r0, r0 + rstep, ...
    if (tid == 0) *singular = 0;
    __syncthreads();
    if (tid < 64) {      // pivot of column 0
        const double a = tid < n ? M0[tid * W] : 0.0;
        const double mag = tid < n ? fabs(a) : -1.0;
        const double best = wave_max_dpp(mag);
        const unsigned long long eq = __ballot(mag == best);
        const int p = eq ? (int)__builtin_ctzll(eq) : 0;        // (no lane compares equal only if the column holds NaNs)
        if (tid == p) { s_p[0] = p; s_pv[0][0] = a; }
        if (tid == 0) { s_pv[0][1] = a; s_best[0] = best; if (best == 0.0) *singular = 1; }
    }
    __syncthreads();
    double* cur = M0; double* nxt = M1;
    for (int c = 0; c < n; ++c) {
        const int par = c & 1;
        const int p = s_p[par];
        const double piv = s_pv[par][0], acc = s_pv[par][1];      // the pivot A[p][c], and A[c][c]: the multiplier row p carries after the swap
        // (branch-free: rows beyond n are clamped onto row n - 1 -- such a lane loads that row's operands and stores that row's value, the
        // same bits its owner stores -- so that no load waits behind a branch: exec-masked loops had made every row its own LDS round trip)
        if (tid < 64) {
            const int j = c + 1 < n ? c + 1 : c, r = tid, rc = r < n ? r : n - 1;
            const double q1 = cur[p * W + j] / piv;
            const double xl = cur[rc * W + j], fl = cur[rc * W + c], topj = cur[c * W + j];
            const double x = rc == p ? topj : xl, f = rc == p ? acc : fl;
            const double v = rc == c ? q1 : x - f * q1;
            const double mag = (r > c && r < n) ? fabs(v) : -1.0;
            const double best = wave_max_dpp(mag);
            const unsigned long long eq = __ballot(mag == best);
            const int pn = eq ? (int)__builtin_ctzll(eq) : j;
            if (c + 1 < n) {
                if (r == pn) { s_p[par ^ 1] = pn; s_pv[par ^ 1][0] = v; }
                if (r == j) s_pv[par ^ 1][1] = v;
                if (r == 0) { s_best[j] = best; if (best == 0.0) *singular = 1; }
            }
        } else if (r0 < rstep) {
            constexpr int MAXR = 11;                                // rows per thread: ceil(32 / 3)
            double xr[MAXR], fr[MAXR];
            int rr[MAXR];
            const double low = cur[p * W + j0], top = cur[c * W + j0];
#pragma unroll
            for (int i = 0; i < MAXR; ++i) {
                const int r = r0 + i * rstep;
                rr[i] = r < n ? r : n - 1;
                xr[i] = cur[rr[i] * W + j0];
                fr[i] = cur[rr[i] * W + c];
            }
            const double q = low / piv;
#pragma unroll
            for (int i = 0; i < MAXR; ++i) {
                const double x = rr[i] == p ? top : xr[i];
                const double f = rr[i] == p ? acc : fr[i];
                const double e = j0 == c ? 0.0 : x - f * q;
                nxt[rr[i] * W + j0] = rr[i] == c ? q : e;
            }
        }
        __syncthreads();
        MMM_GSTAMP(3 + 3 * c);
        double* t = cur; cur = nxt; nxt = t;
    }
    if (tid < n) s_lg[tid] = log(s_best[tid]);
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int c = 0; c < n; ++c) s += s_lg[c]; *logdet = s; }
    return cur;
}

// update_μ! / update_Σ! of one replica by the calling block (>= 128 threads); smem: 4 MK^2 doubles (sum K <= 32: block_inverse_pipelined) or 2 MK^2.  BIG (sum K > 64): the matrices live
// in device memory -- a compile-time switch, so that the LDS build keeps LDS addressing (a run-time choice of the base pointer turned every
// access of the inversion into a flat one: 54 -> 84 us for the Gaussian block at sum K = 28)
template <bool BIG>
__device__ void ctm_gauss_mstep(const MstepArgs& a, const MstepPtrs& q, double* smem)
{
    __shared__ double s_logdet; __shared__ int s_sing, s_piv;
    const CtmDims& dm = a.dm;
    const int MK = dm.MK, tid = threadIdx.x, nt = blockDim.x;
    const double* sLam = q.stats; const double* sNu = q.stats + MK; const double* sLL = q.stats + 2 * MK;
#ifdef MMM_DIAG_STAMPS
    if (tid == 0 && blockIdx.y == 0) g_gauss_stamps[94] = __builtin_amdgcn_s_memrealtime();
#endif
    MMM_GSTAMP(0);
    // update_μ! (MMCTM.jl:200-202)
    if (a.do_mu) { for (int i = tid; i < MK; i += nt) q.mu[i] = sLam[i] / a.Dglobal; }
    __syncthreads();
    MMM_GSTAMP(1);
    // update_Σ! (MMCTM.jl:204-212) from raw moments: (diag Σν + Σ (λ-μ)(λ-μ)') / D with the NEW μ
    if (a.do_sigma) {
        const bool pipe = !BIG && MK <= 32 && nt >= 128 && !a.gauss_wide;      // the one-barrier-per-column inversion over the augmented, double-buffered matrix
        double* A = BIG ? a.big_scratch + (size_t)blockIdx.y * 2 * MK * MK : smem;
        double* Ai = A + MK * MK;
        const int ld = pipe ? 2 * MK : MK;                    // row stride of A in its buffer
        // (every operand of an entry is requested before any entry is formed, four entries per thread at a time: the loop used to pay one
        // L2 round trip per entry after the other, 2.5 us at sum K = 28; with do_mu the thread forms mu_i = sum lambda_i / D itself -- the
        // division update_μ! has just made -- instead of reading it back)
        for (int e0 = tid; e0 < MK * MK; e0 += 4 * nt) {
            double ll[4], li[4], lj[4], nui[4], mi[4], mj[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * nt < MK * MK ? e0 + k * nt : MK * MK - 1;
                const int i = e % MK, j = e / MK;
                ll[k] = sLL[e]; li[k] = sLam[i]; lj[k] = sLam[j]; nui[k] = sNu[i];
                mi[k] = a.do_mu ? 0.0 : q.mu[i]; mj[k] = a.do_mu ? 0.0 : q.mu[j];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * nt;
                const int ec = e < MK * MK ? e : MK * MK - 1;
                const int i = ec % MK, j = ec / MK;
                const double mui = a.do_mu ? li[k] / a.Dglobal : mi[k], muj = a.do_mu ? lj[k] / a.Dglobal : mj[k];
                // Σ_d (λ_i-μ_i)(λ_j-μ_j) = Σλλ' - μ_i Σλ_j - μ_j Σλ_i + D μ_i μ_j
                double v = ll[k] - mui * lj[k] - muj * li[k] + a.Dglobal * mui * muj;
                if (i == j) v += nui[k];
                v /= a.Dglobal;
                if (e < MK * MK) {
                    q.Sigma[e] = v; A[i * ld + j] = v;
                    if (pipe) A[i * ld + MK + j] = i == j ? 1.0 : 0.0;
                }
            }
        }
        __syncthreads();
        const double* R = Ai; int ldr = MK, offr = 0;
        MMM_GSTAMP(2);
        if constexpr (BIG) block_inverse_big(MK, A, Ai, &s_logdet, &s_sing);
        else if (pipe) { R = block_inverse_pipelined(MK, A, A + 2 * MK * MK, &s_logdet, &s_sing); ldr = 2 * MK; offr = MK; }
        else block_inverse(MK, A, Ai, &s_logdet, &s_sing, &s_piv);
        __syncthreads();
        MMM_GSTAMP(90);
        for (int e = tid; e < MK * MK; e += nt) { const int i = e % MK, j = e / MK; q.invSigma[e] = R[i * ldr + offr + j]; }
        if (tid == 0 && s_sing) *q.status = 1;
        __syncthreads();
        MMM_GSTAMP(91);
#ifdef MMM_DIAG_STAMPS
        if (tid == 0 && blockIdx.y == 0) g_gauss_stamps[95] = __builtin_amdgcn_s_memrealtime();
#endif
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
        // the inversion is one dependent chain beside a sweep that is bound by vector issue: its waves go first at every arbitration
        __builtin_amdgcn_s_setprio(3);
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
        // the inversion is one dependent chain beside a sweep that is bound by vector issue: its waves go first at every arbitration
        __builtin_amdgcn_s_setprio(3);
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
