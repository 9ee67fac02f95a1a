// lda.hip -- LDA variational EM on gfx950 (replaces the hot path of src/LDA.jl)
//
// Data layout in HBM (per context / model handle)
//   corpus     doc_ptr int64[D+1]; tc int2[nnz] = (term0, count) interleaved -> one 8-byte load per nonzero
//   topics     lambda, Elnbeta, beta, expElnbeta: [k][v] (= the reference's V x K column-major), V*K doubles
//   documents  gamma, gamma_next, Elntheta, theta: [d][k] (= K x D column-major)
//   phi        [nnz][K] (= per-doc K x W_d blocks, k fastest); written only when asked for (see below)
//
// Hot path = k_lda_estep<KP, FUSED>: one wavefront per document, lanes over the document's nonzero terms.
//   * Elntheta_k = psi(gamma_k) - psi(sum gamma) on lanes 0..K (LDA.jl:78-80)
//   * phi_kw  = a_k * B_vk / sum_k a_k B_vk with a_k = exp(Elntheta_k) (K exps per document) and
//     B = exp(Elnbeta) (V*K exps per iteration, staged in LDS): algebraically exp(Elntheta_k + Elnbeta_vk) of
//     LDA.jl:71-74 without one exp per (term, topic)
//   * lambda scatter (LDA.jl:103-105): term ids are unique inside a document, so a wave adds into its private
//     LDS slab [K][V] with plain read-modify-write (no atomics, deterministic); slabs are reduced per block and
//     written as one partial per block; k_reduce_slabs sums the partials in fixed order
//   * gamma of the NEXT iteration (LDA.jl:85-87 uses the previous phi) = alpha + sum_w phi_kw n_w is formed in
//     the same pass, so phi never round-trips through HBM inside the loop.  phi is materialised on demand
//     (mmm_lda_get(PHI), ELBO) from (Elntheta, previous Elnbeta), which reproduces the stored phi exactly.
#include "dev_math.h"
#include "mmm_internal.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * MMM_WAVE;

struct LdaDev {
    int D, V, K;
    const int64_t* doc_ptr;
    const int2* tc;
    double alpha, eta;
};

enum { MODE_FUSED = 0, MODE_PHI = 1 };

struct EstepArgs {
    LdaDev c;
    const double* gamma;      // FUSED: in
    double* Elntheta;         // FUSED: out, PHI: in
    double* gamma_next;       // FUSED: out
    const double* expElnbeta; // [K][V]
    double* partial;          // FUSED: [gridDim][K*V]
    double* phi;              // PHI: out
};

template <int KP, int MODE>
__global__ __launch_bounds__(kBlock) void k_lda_estep(EstepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int K = a.c.K, V = a.c.V, D = a.c.D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double* sB = smem;                                  // [KP][V]
    double* sSlab = smem + (size_t)KP * V;              // [waves][KP][V] (FUSED only)
    for (int i = tid; i < KP * V; i += kBlock) sB[i] = (i < K * V) ? a.expElnbeta[i] : 0.0;
    if (MODE == MODE_FUSED)
        for (int i = tid; i < kWavesPerBlock * KP * V; i += kBlock) sSlab[i] = 0.0;
    __syncthreads();
    double* slab = sSlab + (size_t)wid * KP * V;

    for (int d = blockIdx.x * kWavesPerBlock + wid; d < D; d += gridDim.x * kWavesPerBlock) {
        double el;
        if (MODE == MODE_FUSED) {
            const double g = (lane < K) ? a.gamma[(size_t)d * K + lane] : 0.0;
            const double S = wave_sum(g);
            const double ps = dev_digamma(lane < K ? g : S);          // lane K (and above) holds psi(S)
            el = ps - wave_bcast(ps, K);
            if (lane < K) a.Elntheta[(size_t)d * K + lane] = el;
        } else {
            el = (lane < K) ? a.Elntheta[(size_t)d * K + lane] : 0.0;
        }
        const double ak = (lane < K) ? exp(el) : 0.0;
        double av[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) av[k] = wave_bcast(ak, k);

        double acc[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] = 0.0;
        const int64_t start = a.c.doc_ptr[d];
        const int W = (int)(a.c.doc_ptr[d + 1] - start);
        for (int w0 = 0; w0 < W; w0 += MMM_WAVE) {
            const int w = w0 + lane;
            const bool act = w < W;
            int2 t = act ? a.c.tc[start + w] : make_int2(0, 0);
            const int v = t.x;
            double e[KP], s = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) { e[k] = av[k] * sB[k * V + v]; s += e[k]; }
            if (MODE == MODE_FUSED) {
                const double r = act ? (double)t.y / s : 0.0;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    if (k < K) {
                        const double pn = e[k] * r;
                        acc[k] += pn;
                        if (act) slab[k * V + v] += pn;
                    }
                }
            } else if (act) {
                double* ph = a.phi + (size_t)(start + w) * K;
#pragma unroll
                for (int k = 0; k < KP; ++k) if (k < K) ph[k] = e[k] / s;
            }
        }
        if (MODE == MODE_FUSED) {
            double mine = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (k < K) { const double tot = wave_sum(acc[k]); if (lane == k) mine = tot; }
            }
            if (lane < K) a.gamma_next[(size_t)d * K + lane] = a.c.alpha + mine;
        }
    }
    if (MODE == MODE_FUSED) {
        __syncthreads();
        double* out = a.partial + (size_t)blockIdx.x * K * V;
        for (int i = tid; i < K * V; i += kBlock) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) s += sSlab[(size_t)w * KP * V + i];
            out[i] = s;
        }
    }
}

// partial[nslab][n] -> out[n], fixed summation order (deterministic)
__global__ __launch_bounds__(1024) void k_reduce_slabs(const double* __restrict__ part, int nslab, int n, double* __restrict__ out)
{
    __shared__ double s[16][64];
    const int e = blockIdx.x * 64 + threadIdx.x, y = threadIdx.y;
    double acc = 0.0;
    if (e < n) for (int sl = y; sl < nslab; sl += 16) acc += part[(size_t)sl * n + e];
    s[y][threadIdx.x] = acc;
    __syncthreads();
    if (y == 0 && e < n) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += s[j][threadIdx.x];
        out[e] = t;
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

// M-step per topic k (one block per topic): lambda = eta + sums (LDA.jl:101-105), Elnbeta (LDA.jl:96-98),
// beta (LDA.jl:110-112), and the exp(Elnbeta) table of the next E-step.
__global__ __launch_bounds__(256) void k_lda_mstep(int V, double eta, const double* sums, double* lambda, double* Elnbeta,
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

// theta = gamma / sum gamma (LDA.jl:92-94) and the per-iteration log-likelihood numerator (LDA.jl:174-188)
// one wave per document; beta staged in LDS.  llpart[blockIdx] = sum over the block's documents.
template <int KP>
__global__ __launch_bounds__(kBlock) void k_lda_loglik(LdaDev c, const double* gamma, const double* beta, double* theta,
                                                       double* llpart, int compute_ll)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double shw[kWavesPerBlock];
    const int K = c.K, V = c.V;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (compute_ll) {
        for (int i = tid; i < KP * V; i += kBlock) smem[i] = (i < K * V) ? beta[i] : 0.0;
        __syncthreads();
    }
    double wave_ll = 0.0;
    for (int d = blockIdx.x * kWavesPerBlock + wid; d < c.D; d += gridDim.x * kWavesPerBlock) {
        const double g = (lane < K) ? gamma[(size_t)d * K + lane] : 0.0;
        const double S = wave_sum(g);
        const double th = g / S;
        if (lane < K) theta[(size_t)d * K + lane] = th;
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
            for (int k = 0; k < KP; ++k) p += tv[k] * smem[k * V + t.x];
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

// out[j] = sum_i part[i*stride + j], j < nvals  (one wave per j)
__global__ __launch_bounds__(64) void k_sum_columns(const double* part, int n, int stride, double* out)
{
    const int j = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += part[(size_t)i * stride + j];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[j] = acc;
}

__global__ void k_ll_store(const double* num, double N, double* dst) { *dst = *num / N; }

// ---- stage kernels (reference-granularity entry points; not on the fused path) ----------------------------
// gamma[:,d] = alpha + phi[d] * n_d (LDA.jl:83-87) from a resident phi, then Elntheta
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
        const double S = wave_sum(g);
        const double ps = dev_digamma(lane < K ? g : S);
        const double el = ps - wave_bcast(ps, K);
        if (lane < K) { gamma[(size_t)d * K + lane] = g; Elntheta[(size_t)d * K + lane] = el; }
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
        const double el = ps - wave_bcast(ps, K);
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

__global__ void k_fill(double* p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_doc_counts(LdaDev c, double* out)   // out[0] += sum of counts (via per-thread partials)
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
    int D = 0, V = 0, K = 0, KP = 0;
    int64_t nnz = 0;
    double alpha = 0, eta = 0;
    double Nglobal = 0, Dglobal = 0;
    DevBuf<int64_t> doc_ptr; DevBuf<int2> tc;
    DevBuf<double> lambda, Elnbeta, Elnbeta_prev, expElnbeta, expElnbeta_prev, beta;
    DevBuf<double> gamma, gamma_next, Elntheta, theta, phi;
    DevBuf<double> partial, stats, llpart, elbopart, ll_hist;
    bool phi_valid = false, gnext_valid = false;
    int n_hist = 0, cap_hist = 0;
    int grid_e = 1;
    size_t lds_e = 0, lds_ll = 0;
    LdaDev dev() const { return LdaDev{D, V, K, doc_ptr.p, tc.p, alpha, eta}; }
};

namespace {

int pick_kp(int K)
{
    static const int opts[] = {2, 4, 6, 8, 10, 12, 16, 20, 24, 32};
    for (int o : opts) if (K <= o) return o;
    return -1;
}

template <int MODE>
int launch_estep(mmm_lda* m, const EstepArgs& a, size_t lds)
{
    mmm_ctx* ctx = m->ctx;
#define MMM_CASE(KPV)                                                                                          \
    case KPV: {                                                                                                \
        auto kern = k_lda_estep<KPV, MODE>;                                                                    \
        if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(m->grid_e), dim3(kBlock), lds, ctx->stream, a);                           \
        break;                                                                                                 \
    }
    switch (m->KP) {
        MMM_CASE(2) MMM_CASE(4) MMM_CASE(6) MMM_CASE(8) MMM_CASE(10) MMM_CASE(12) MMM_CASE(16) MMM_CASE(20) MMM_CASE(24) MMM_CASE(32)
        default: return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "LDA: K=%d not supported (max 32)", m->K);
    }
#undef MMM_CASE
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

int launch_loglik(mmm_lda* m, int compute_ll)
{
    mmm_ctx* ctx = m->ctx;
    size_t lds = compute_ll ? m->lds_ll : 0;
#define MMM_CASE(KPV)                                                                                          \
    case KPV: {                                                                                                \
        auto kern = k_lda_loglik<KPV>;                                                                         \
        if (lds > 48 * 1024) MMM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(m->grid_e), dim3(kBlock), lds, ctx->stream, m->dev(), m->gamma.p, m->beta.p, m->theta.p, m->llpart.p, compute_ll); \
        break;                                                                                                 \
    }
    switch (m->KP) {
        MMM_CASE(2) MMM_CASE(4) MMM_CASE(6) MMM_CASE(8) MMM_CASE(10) MMM_CASE(12) MMM_CASE(16) MMM_CASE(20) MMM_CASE(24) MMM_CASE(32)
        default: return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "LDA: K=%d not supported (max 32)", m->K);
    }
#undef MMM_CASE
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

// lambda = eta + all-reduced sums; Elnbeta (previous one kept for phi materialisation); beta; exp table
int run_mstep(mmm_lda* m, bool from_sums)
{
    mmm_ctx* ctx = m->ctx;
    if (from_sums) {
        int rc = mmm_allreduce_sum(ctx, m->stats.p, (size_t)m->V * m->K);
        if (rc) return rc;
    }
    m->Elnbeta.swap(m->Elnbeta_prev);
    m->expElnbeta.swap(m->expElnbeta_prev);
    hipLaunchKernelGGL(k_lda_mstep, dim3(m->K), dim3(256), 0, ctx->stream, m->V, m->eta, from_sums ? m->stats.p : nullptr,
                       m->lambda.p, m->Elnbeta.p, m->expElnbeta.p, m->beta.p, 0);
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

int run_beta(mmm_lda* m)
{
    hipLaunchKernelGGL(k_lda_mstep, dim3(m->K), dim3(256), 0, m->ctx->stream, m->V, m->eta, (const double*)nullptr, m->lambda.p,
                       (double*)nullptr, (double*)nullptr, m->beta.p, 1);
    MMM_LAUNCH_CHECK(m->ctx);
    return MMM_OK;
}

// materialise phi of the last fused iteration: softmax_k(Elntheta + previous Elnbeta) (LDA.jl:69-76)
int materialise_phi(mmm_lda* m)
{
    if (m->phi_valid) return MMM_OK;
    EstepArgs a{m->dev(), nullptr, m->Elntheta.p, nullptr, m->expElnbeta_prev.p, nullptr, m->phi.p};
    int rc = launch_estep<MODE_PHI>(m, a, (size_t)m->KP * m->V * sizeof(double));
    if (rc) return rc;
    m->phi_valid = true;
    return MMM_OK;
}

int ll_to_history(mmm_lda* m, double* dst_dev)
{
    mmm_ctx* ctx = m->ctx;
    double* num = m->stats.p + (size_t)m->V * m->K;
    hipLaunchKernelGGL(k_sum_columns, dim3(1), dim3(64), 0, ctx->stream, m->llpart.p, m->grid_e, 1, num);
    MMM_LAUNCH_CHECK(ctx);
    int rc = mmm_allreduce_sum(ctx, num, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ll_store, dim3(1), dim3(1), 0, ctx->stream, num, m->Nglobal, dst_dev);
    MMM_LAUNCH_CHECK(ctx);
    return MMM_OK;
}

int ensure_hist(mmm_lda* m, int extra)
{
    if (m->n_hist + extra <= m->cap_hist) return MMM_OK;
    int cap = std::max(2 * m->cap_hist, m->n_hist + extra + 64);
    DevBuf<double> nb;
    MMM_HIP(m->ctx, nb.alloc(cap));
    if (m->n_hist) MMM_HIP(m->ctx, hipMemcpyAsync(nb.p, m->ll_hist.p, sizeof(double) * m->n_hist, hipMemcpyDeviceToDevice, m->ctx->stream));
    MMM_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
    m->ll_hist.swap(nb);
    m->cap_hist = cap;
    return MMM_OK;
}

} // namespace

extern "C" {

int mmm_lda_create(mmm_ctx* ctx, int D, int V, int K, double alpha, double eta, const int64_t* doc_ptr, const int32_t* term,
                   const int32_t* count, const double* lambda0, mmm_lda** out)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_CHECK(ctx, out && doc_ptr && lambda0, "mmm_lda_create: NULL argument");
    MMM_CHECK(ctx, D >= 0 && V >= 1 && K >= 1, "mmm_lda_create: bad sizes D=%d V=%d K=%d", D, V, K);
    MMM_CHECK(ctx, K < 64, "mmm_lda_create: K=%d must be < 64", K);
    *out = nullptr;
    const int KP = pick_kp(K);
    if (KP < 0) return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_lda_create: K=%d not supported (max 32)", K);
    const int64_t nnz = doc_ptr[D];
    MMM_CHECK(ctx, doc_ptr[0] == 0 && nnz >= 0 && (nnz == 0 || (term && count)), "mmm_lda_create: bad CSR");
    std::vector<int2> tc((size_t)nnz);
    for (int d = 0; d < D; ++d) MMM_CHECK(ctx, doc_ptr[d + 1] >= doc_ptr[d], "mmm_lda_create: doc_ptr not monotone at %d", d);
    for (int64_t e = 0; e < nnz; ++e) {
        MMM_CHECK(ctx, term[e] >= 0 && term[e] < V && count[e] >= 0, "mmm_lda_create: entry %lld out of range (term %d, count %d)", (long long)e, term[e], count[e]);
        tc[(size_t)e] = make_int2(term[e], count[e]);
    }
    const size_t lds_e = (size_t)KP * V * (1 + kWavesPerBlock) * sizeof(double);
    if (lds_e > 160 * 1024)
        return mmm_fail(ctx, MMM_ERR_UNSUPPORTED, "mmm_lda_create: K*V = %d*%d needs %zu B of LDS (> 160 KiB)", K, V, lds_e);

    MMM_HIP(ctx, hipSetDevice(ctx->device));
    mmm_lda* m = new mmm_lda();
    m->ctx = ctx; m->D = D; m->V = V; m->K = K; m->KP = KP; m->nnz = nnz; m->alpha = alpha; m->eta = eta;
    m->lds_e = lds_e; m->lds_ll = (size_t)KP * V * sizeof(double);
    const size_t VK = (size_t)V * K, KD = (size_t)K * D;
    const int max_blocks_lds = (int)std::max<size_t>(1, (160 * 1024) / lds_e);
    const int per_cu = std::min(2, max_blocks_lds);
    m->grid_e = std::max(1, std::min((D + kWavesPerBlock - 1) / kWavesPerBlock, ctx->num_cu * per_cu));
    if (const char* g = getenv("MMM_LDA_GRID")) m->grid_e = std::max(1, atoi(g));
#define A(buf, n) do { hipError_t e_ = m->buf.alloc(n); if (e_ != hipSuccess) { int rc = mmm_fail(ctx, MMM_ERR_HIP, "hipMalloc(" #buf "): %s", hipGetErrorString(e_)); delete m; return rc; } } while (0)
    A(doc_ptr, (size_t)D + 1); A(tc, (size_t)nnz);
    A(lambda, VK); A(Elnbeta, VK); A(Elnbeta_prev, VK); A(expElnbeta, VK); A(expElnbeta_prev, VK); A(beta, VK);
    A(gamma, KD); A(gamma_next, KD); A(Elntheta, KD); A(theta, KD); A(phi, (size_t)K * nnz);
    A(partial, (size_t)m->grid_e * VK); A(stats, VK + 16); A(llpart, (size_t)m->grid_e); A(elbopart, (size_t)m->grid_e * 5 + 8);
#undef A
    hipStream_t st = ctx->stream;
    MMM_HIP(ctx, hipMemcpyAsync(m->doc_ptr.p, doc_ptr, sizeof(int64_t) * (D + 1), hipMemcpyHostToDevice, st));
    if (nnz) MMM_HIP(ctx, hipMemcpyAsync(m->tc.p, tc.data(), sizeof(int2) * nnz, hipMemcpyHostToDevice, st));
    MMM_HIP(ctx, hipMemcpyAsync(m->lambda.p, lambda0, sizeof(double) * VK, hipMemcpyHostToDevice, st));
    MMM_HIP(ctx, hipStreamSynchronize(st));   // tc (host vector) must outlive the copy
    // constructor state (LDA.jl:36-49): Elnbeta from lambda0; gamma = 1 -> Elntheta; phi = 1/K
    hipLaunchKernelGGL(k_lda_mstep, dim3(K), dim3(256), 0, st, V, eta, (const double*)nullptr, m->lambda.p, m->Elnbeta.p, m->expElnbeta.p, m->beta.p, 0);
    if (KD) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((KD + 255) / 256)), dim3(256), 0, st, m->gamma.p, KD, 1.0);
        hipLaunchKernelGGL(k_lda_Elntheta, dim3(m->grid_e), dim3(kBlock), 0, st, m->dev(), m->gamma.p, m->Elntheta.p);
    }
    if (nnz) hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)K * nnz + 255) / 256)), dim3(256), 0, st, m->phi.p, (size_t)K * nnz, 1.0 / K);
    MMM_HIP(ctx, hipMemsetAsync(m->stats.p, 0, sizeof(double) * (VK + 16), st));
    double* ncount = m->stats.p + VK + 1;
    if (nnz) hipLaunchKernelGGL(k_doc_counts, dim3(64), dim3(256), 0, st, m->dev(), ncount);
    MMM_LAUNCH_CHECK(ctx);
    // global N and D (sum over ranks)
    double hd[2] = {0.0, (double)D};
    MMM_HIP(ctx, hipMemcpyAsync(&hd[0], ncount, sizeof(double), hipMemcpyDeviceToHost, st));
    MMM_HIP(ctx, hipStreamSynchronize(st));
    if (ctx->nranks > 1) {
        MMM_HIP(ctx, hipMemcpyAsync(ncount, hd, sizeof hd, hipMemcpyHostToDevice, st));
        int rc = mmm_allreduce_sum(ctx, ncount, 2);
        if (rc) { delete m; return rc; }
        MMM_HIP(ctx, hipMemcpyAsync(hd, ncount, sizeof hd, hipMemcpyDeviceToHost, st));
        MMM_HIP(ctx, hipStreamSynchronize(st));
    }
    m->Nglobal = hd[0]; m->Dglobal = hd[1];
    m->phi_valid = true; m->gnext_valid = false;
    *out = m;
    return MMM_OK;
}

int mmm_lda_destroy(mmm_lda* m)
{
    if (!m) return MMM_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    delete m;
    return MMM_OK;
}

static int lda_field(mmm_lda* m, int field, double** p, size_t* n)
{
    const size_t VK = (size_t)m->V * m->K, KD = (size_t)m->K * m->D;
    switch (field) {
        case MMM_LDA_LAMBDA: *p = m->lambda.p; *n = VK; break;
        case MMM_LDA_ELNBETA: *p = m->Elnbeta.p; *n = VK; break;
        case MMM_LDA_BETA: *p = m->beta.p; *n = VK; break;
        case MMM_LDA_GAMMA: *p = m->gamma.p; *n = KD; break;
        case MMM_LDA_ELNTHETA: *p = m->Elntheta.p; *n = KD; break;
        case MMM_LDA_THETA: *p = m->theta.p; *n = KD; break;
        case MMM_LDA_PHI: *p = m->phi.p; *n = (size_t)m->K * m->nnz; break;
        default: return mmm_fail(m->ctx, MMM_ERR_ARG, "unknown LDA field %d", field);
    }
    return MMM_OK;
}

int mmm_lda_get(mmm_lda* m, int field, double* host, size_t n)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    double* p; size_t cnt;
    int rc = lda_field(m, field, &p, &cnt);
    if (rc) return rc;
    MMM_CHECK(ctx, host && n == cnt, "mmm_lda_get(field %d): expected %zu doubles, got %zu", field, cnt, n);
    if (field == MMM_LDA_PHI && (rc = materialise_phi(m))) return rc;
    if (n) MMM_HIP(ctx, hipMemcpyAsync(host, p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_lda_set(mmm_lda* m, int field, const double* host, size_t n)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    double* p; size_t cnt;
    int rc = lda_field(m, field, &p, &cnt);
    if (rc) return rc;
    MMM_CHECK(ctx, host && n == cnt, "mmm_lda_set(field %d): expected %zu doubles, got %zu", field, cnt, n);
    if ((rc = materialise_phi(m))) return rc;     // make the implicit phi explicit before state is overwritten
    m->gnext_valid = false;
    if (n) MMM_HIP(ctx, hipMemcpyAsync(p, host, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (field == MMM_LDA_ELNBETA && n) {
        hipLaunchKernelGGL(k_exp_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, m->Elnbeta.p, m->expElnbeta.p);
        MMM_LAUNCH_CHECK(ctx);
    }
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_lda_update_gamma(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    MMM_HIP(m->ctx, hipSetDevice(m->ctx->device));
    int rc = materialise_phi(m);
    if (rc) return rc;
    hipLaunchKernelGGL(k_lda_gamma_from_phi, dim3(m->grid_e), dim3(kBlock), 0, m->ctx->stream, m->dev(), m->phi.p, m->gamma.p, m->Elntheta.p);
    MMM_LAUNCH_CHECK(m->ctx);
    m->gnext_valid = false;
    return MMM_OK;
}

int mmm_lda_update_phi(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    MMM_HIP(m->ctx, hipSetDevice(m->ctx->device));
    EstepArgs a{m->dev(), nullptr, m->Elntheta.p, nullptr, m->expElnbeta.p, nullptr, m->phi.p};
    int rc = launch_estep<MODE_PHI>(m, a, (size_t)m->KP * m->V * sizeof(double));
    if (rc) return rc;
    m->phi_valid = true; m->gnext_valid = false;
    return MMM_OK;
}

int mmm_lda_update_lambda(mmm_lda* m)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    int rc = materialise_phi(m);
    if (rc) return rc;
    MMM_HIP(ctx, hipMemsetAsync(m->stats.p, 0, sizeof(double) * (size_t)m->V * m->K, ctx->stream));
    if (m->nnz) hipLaunchKernelGGL(k_lda_lambda_from_phi, dim3((unsigned)((m->nnz + 255) / 256)), dim3(256), 0, ctx->stream, m->dev(), m->nnz, m->phi.p, m->stats.p);
    MMM_LAUNCH_CHECK(ctx);
    m->gnext_valid = false;
    return run_mstep(m, true);
}

int mmm_lda_update_beta(mmm_lda* m) { if (!m) return MMM_ERR_ARG; MMM_HIP(m->ctx, hipSetDevice(m->ctx->device)); return run_beta(m); }

int mmm_lda_update_theta(mmm_lda* m) { if (!m) return MMM_ERR_ARG; MMM_HIP(m->ctx, hipSetDevice(m->ctx->device)); return launch_loglik(m, 0); }

int mmm_lda_loglik(mmm_lda* m, double* ll)
{
    if (!m || !ll) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    // the reference evaluates with the stored theta and beta (LDA.jl:194-196); this entry point recomputes
    // theta from gamma first, which is what fit! has just done (LDA.jl:207) -- beta must be current.
    int rc = launch_loglik(m, 1);
    if (rc) return rc;
    double* dst = m->stats.p + (size_t)m->V * m->K + 4;
    if ((rc = ll_to_history(m, dst))) return rc;
    MMM_HIP(ctx, hipMemcpyAsync(ll, dst, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int mmm_lda_iterate(mmm_lda* m, int n_iter)
{
    if (!m) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    MMM_CHECK(ctx, n_iter >= 0, "mmm_lda_iterate: n_iter < 0");
    int rc = ensure_hist(m, n_iter);
    if (rc) return rc;
    const int VK = m->V * m->K;
    for (int it = 0; it < n_iter; ++it) {
        // update_γ! (LDA.jl:82-90): gamma for this pass was formed from the previous pass's phi
        if (m->gnext_valid) m->gamma.swap(m->gamma_next);
        else {
            if ((rc = materialise_phi(m))) return rc;
            hipLaunchKernelGGL(k_lda_gamma_from_phi, dim3(m->grid_e), dim3(kBlock), 0, ctx->stream, m->dev(), m->phi.p, m->gamma.p, m->Elntheta.p);
            MMM_LAUNCH_CHECK(ctx);
        }
        // update_ϕ! + the document loop of update_λ! + next pass's update_γ!, fused (LDA.jl:69-76,103-105,85-87)
        EstepArgs a{m->dev(), m->gamma.p, m->Elntheta.p, m->gamma_next.p, m->expElnbeta.p, m->partial.p, nullptr};
        { ProfSpan span(ctx); rc = launch_estep<MODE_FUSED>(m, a, m->lds_e); }
        if (rc) return rc;
        m->phi_valid = false; m->gnext_valid = true;
        hipLaunchKernelGGL(k_reduce_slabs, dim3((VK + 63) / 64), dim3(64, 16), 0, ctx->stream, m->partial.p, m->grid_e, VK, m->stats.p);
        MMM_LAUNCH_CHECK(ctx);
        // update_λ! tail, update_Elnβ! (LDA.jl:96-108)
        if ((rc = run_mstep(m, true))) return rc;
        // update_β!, update_θ!, log-likelihood (LDA.jl:206-209)
        if ((rc = run_beta(m))) return rc;
        if ((rc = launch_loglik(m, 1))) return rc;
        if ((rc = ll_to_history(m, m->ll_hist.p + m->n_hist))) return rc;
        m->n_hist++;
    }
    return MMM_OK;
}

int mmm_lda_ll_history(mmm_lda* m, double* ll, int max_n, int* n)
{
    if (!m || !n) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    const int cnt = std::min(max_n, m->n_hist);
    if (cnt > 0 && ll) MMM_HIP(ctx, hipMemcpyAsync(ll, m->ll_hist.p + (m->n_hist - cnt), sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n = cnt;
    return MMM_OK;
}

int mmm_lda_elbo(mmm_lda* m, double* elbo, double terms[7])
{
    if (!m || !elbo) return MMM_ERR_ARG;
    mmm_ctx* ctx = m->ctx;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    int rc = materialise_phi(m);
    if (rc) return rc;
    double* acc = m->elbopart.p + (size_t)m->grid_e * 5;      // [0..4] doc sums, [5..6] topic sums
    hipLaunchKernelGGL(k_lda_elbo_docs, dim3(m->grid_e), dim3(kBlock), 0, ctx->stream, m->dev(), m->phi.p, m->gamma.p, m->Elntheta.p, m->Elnbeta.p, m->elbopart.p);
    hipLaunchKernelGGL(k_sum_columns, dim3(5), dim3(64), 0, ctx->stream, m->elbopart.p, m->grid_e, 5, acc);
    hipLaunchKernelGGL(k_lda_elbo_topics, dim3(1), dim3(256), 0, ctx->stream, m->V, m->K, m->lambda.p, m->Elnbeta.p, acc + 5);
    MMM_LAUNCH_CHECK(ctx);
    if ((rc = mmm_allreduce_sum(ctx, acc, 5))) return rc;
    double h[7];
    MMM_HIP(ctx, hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double K = m->K, V = m->V, al = m->alpha, et = m->eta;
    double t[7];
    t[0] = K * (lgamma(V * et) - V * lgamma(et)) + (et - 1.0) * h[5];            // LDA.jl:114-118
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
    *converged = 0;
    const int base = m->n_hist;
    int done = 0;
    std::vector<double> ll((size_t)maxiter);
    // the convergence test needs > 10 values (LDA.jl:215): run the first 11 passes unsynchronised, then one
    // pass per host check
    while (done < maxiter) {
        const int chunk = (done == 0) ? std::min(maxiter, 11) : 1;
        int rc = mmm_lda_iterate(m, chunk);
        if (rc) return rc;
        MMM_HIP(ctx, hipMemcpyAsync(ll.data() + done, m->ll_hist.p + base + done, sizeof(double) * chunk, hipMemcpyDeviceToHost, ctx->stream));
        MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        done += chunk;
        if (done > 10) {   // common.jl:53-56
            const double rel = fabs(ll[done - 2] - ll[done - 1]) / fabs(ll[done - 1]);
            if (rel < tol) { *converged = 1; break; }
        }
    }
    *n_iter = done;
    if (ll_hist) memcpy(ll_hist, ll.data(), sizeof(double) * done);
    if (elbo) return mmm_lda_elbo(m, elbo, nullptr);
    return MMM_OK;
}

} // extern "C"
