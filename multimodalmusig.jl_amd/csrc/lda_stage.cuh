// lda_stage.cuh -- stage / on-demand kernels of lda.hip (included there, inside its anonymous namespace): phi, ll, the wide-vocabulary sweeps
// (document-major and term-major), the builds for more than 32 / 64 topics, ELBO terms.  LDA.jl:69-172; DESIGN section 4.8.
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
