// free.hip -- the reference's FREE functions: their arguments are the caller's arrays, not a model.  The reference's own tests call them
// directly (test/common.jl:79-97; test/mmctm.jl:135-148,268-293,349-388; test/immctm.jl:122-160,273-294,350-386), so the drop-in
// boundary carries them too; each is one entry point here and one `ccall` in the Julia shim.
//   λ_objective / ν_objective / α_objective      src/common.jl:11-46   (maximisation form, as the reference states them)
//   calculate_loglikelihood(X, θ, β)             src/LDA.jl:174-188
//   calculate_(doc)modality_loglikelihood        src/MMCTM.jl:384-418 (props, ϕ) and src/IMMCTM.jl:362-407 (η, factor ϕ, features)
// Small, latency-bound launches: one block per call for the objectives, one wave per document for the log-likelihood with a fixed-order
// reduction across documents (same bits run to run).  The scalar functions are those of the fit kernels (mmm_arith.h / dev_math.h).
#include "mmm_internal.h"
#include "dev_math.h"
#include "mmm_arith.h"

namespace {

// mode 0: λ_objective(x = λ, other = ν, c = Ndivζ, sumθ, μ, S = invΣ);  mode 1: ν_objective(x = ν, other = λ, c, μ unused, S)
// work[0..n): gradient; work[n..4n): the addends of the value's three sums; out[0] = value.  One block; coordinates strided over threads;
// thread 0 adds the addends in index order -- the order of the reference's sum(...) over a Vector.
__global__ __launch_bounds__(256) void k_free_objective(int mode, int n, const double* x, const double* other, const double* c, const double* sumth,
                                                        const double* mu, const double* S, double* work, double* out)
{
    double* grad = work; double* t0 = work + n; double* t1 = work + 2 * (size_t)n; double* t2 = work + 3 * (size_t)n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (mode == 0) {
            double Sd = 0.0;                                             // (invΣ · (λ − μ))_i, j ascending
            for (int j = 0; j < n; ++j) Sd = fma(S[(size_t)j * n + i], x[j] - mu[j], Sd);
            const double E = ar_exp(x[i] + 0.5 * other[i]);
            grad[i] = -Sd + sumth[i] - c[i] * E;                         // common.jl:19
            t0[i] = (x[i] - mu[i]) * Sd; t1[i] = x[i] * sumth[i]; t2[i] = c[i] * E;
        } else {
            const double E = ar_exp(other[i] + 0.5 * x[i]);
            const double Sii = S[(size_t)i * n + i];
            grad[i] = -0.5 * Sii - (c[i] * 0.5) * E + dev_div(1.0, 2.0 * x[i]);   // common.jl:32
            t0[i] = x[i] * Sii; t1[i] = c[i] * E; t2[i] = ar_log(x[i]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, d = 0.0;
        for (int i = 0; i < n; ++i) { a += t0[i]; b += t1[i]; d += t2[i]; }
        out[0] = mode == 0 ? -0.5 * a + b - d          // common.jl:22
                           : -0.5 * a - b + d / 2.0;   // common.jl:35
    }
}

// α_objective (common.jl:38-46): out[0] = K (lnΓ(Vα) − V lnΓ(α)) + α ΣElnϕ, out[1] = K V (ψ(Vα) − ψ(α)) + ΣElnϕ -- the expressions of k_ctm_update_alpha
__global__ void k_free_alpha_objective(double alpha, double s, double K, double V, double* out)
{
    out[0] = K * (lgamma(V * alpha) - V * lgamma(alpha)) + alpha * s;
    out[1] = K * V * (dev_digamma(V * alpha) - dev_digamma(alpha)) + s;
}

// props[k + K d] = softmax(eta[:, d])_k  (IMMCTM.jl:365: exp.(η) ./ sum(exp.(η)); no max-subtraction there either)
__global__ void k_free_softmax(int D, int K, const double* eta, double* props)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += ar_exp(eta[(size_t)d * K + k]);
    for (int k = 0; k < K; ++k) props[(size_t)d * K + k] = dev_div(ar_exp(eta[(size_t)d * K + k]), s);
}

// phieff[k V + v] = prod_i phi[k SJ + joff_i + f_vi]  (the inner loop of IMMCTM.jl:374-378, factors multiplied in feature order)
__global__ void k_free_phieff(int K, int V, int I, int SJ, const int* joff, const int32_t* features, const double* phi, double* phieff)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * V) return;
    const int k = idx / V, v = idx - k * V;
    double p = 1.0;
    for (int i = 0; i < I; ++i) p *= phi[(size_t)k * SJ + joff[i] + features[(size_t)i * V + v]];
    phieff[idx] = p;
}

// one wave per document: docsum[d] = sum_w n_w log(sum_k props[k, d] phi[k V + v_w])  (k ascending as MMCTM.jl:392-396 / LDA.jl:183), docN[d] = N_d
__global__ __launch_bounds__(256) void k_free_loglik_docs(int D, int K, int V, const int64_t* doc_ptr, const int32_t* term, const int32_t* count,
                                                          const double* props, const double* phi, double* docsum, double* docN)
{
    const int lane = threadIdx.x & 63;
    const int d = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (d >= D) return;
    const double* p = props + (size_t)d * K;
    double s = 0.0, N = 0.0;
    for (int64_t e = doc_ptr[d] + lane; e < doc_ptr[d + 1]; e += 64) {
        const int v = term[e];
        const double n = (double)count[e];
        double pw = 0.0;
        for (int k = 0; k < K; ++k) pw += p[k] * phi[(size_t)k * V + v];
        s += n * ar_log(pw);
        N += n;
    }
    s = wave_sum(s); N = wave_sum(N);
    if (lane == 0) { docsum[d] = s; docN[d] = N; }
}

// out[0] = sum_d docsum[d] / sum_d docN[d] over the documents with N_d > 0 (MMCTM.jl:406-417): 256 strided partial sums, then a fixed tree
__global__ __launch_bounds__(256) void k_free_loglik_total(int D, const double* docsum, const double* docN, double* out)
{
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (int d = threadIdx.x; d < D; d += 256) if (docN[d] > 0.0) { a += docsum[d]; b += docN[d]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sh[0][0] / sh[1][0]; out[1] = sh[0][0]; out[2] = sh[1][0]; }
}

int check_csr(mmm_ctx* ctx, const char* who, int D, int V, const int64_t* doc_ptr, const int32_t* term, const int32_t* count)
{
    MMM_CHECK(ctx, D >= 0 && doc_ptr, "%s: D < 0 or doc_ptr == NULL", who);
    MMM_CHECK(ctx, doc_ptr[0] == 0, "%s: doc_ptr[0] != 0", who);
    for (int d = 0; d < D; ++d) MMM_CHECK(ctx, doc_ptr[d + 1] >= doc_ptr[d], "%s: doc_ptr decreases at document %d", who, d);
    const int64_t nnz = doc_ptr[D];
    MMM_CHECK(ctx, nnz == 0 || (term && count), "%s: NULL term / count", who);
    for (int64_t e = 0; e < nnz; ++e)
        MMM_CHECK(ctx, term[e] >= 0 && term[e] < V && count[e] >= 0, "%s: entry %lld has term %d (V = %d), count %d", who, (long long)e, term[e], V, count[e]);
    return MMM_OK;
}

// device copies of the CSR arrays + log-likelihood from device-resident props [K x D] and phi [k V + v]
int loglik_from_tables(mmm_ctx* ctx, int D, int K, int V, const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const double* props_dev,
                       const double* phi_dev, double* ll)
{
    const int64_t nnz = doc_ptr[D];
    DevBuf<int64_t> dp; DevBuf<int32_t> t, c; DevBuf<double> tmp;
    MMM_HIP(ctx, dp.alloc((size_t)D + 1)); MMM_HIP(ctx, t.alloc((size_t)nnz)); MMM_HIP(ctx, c.alloc((size_t)nnz)); MMM_HIP(ctx, tmp.alloc(2 * (size_t)D + 4));
    MMM_HIP(ctx, hipMemcpyAsync(dp.p, doc_ptr, sizeof(int64_t) * ((size_t)D + 1), hipMemcpyHostToDevice, ctx->stream));
    if (nnz) {
        MMM_HIP(ctx, hipMemcpyAsync(t.p, term, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, ctx->stream));
        MMM_HIP(ctx, hipMemcpyAsync(c.p, count, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, ctx->stream));
    }
    if (D) hipLaunchKernelGGL(k_free_loglik_docs, dim3((unsigned)((D + 3) / 4)), dim3(256), 0, ctx->stream, D, K, V, dp.p, t.p, c.p, props_dev, phi_dev, tmp.p, tmp.p + D);
    hipLaunchKernelGGL(k_free_loglik_total, dim3(1), dim3(256), 0, ctx->stream, D, tmp.p, tmp.p + D, tmp.p + 2 * (size_t)D);
    MMM_LAUNCH_CHECK(ctx);
    MMM_HIP(ctx, hipMemcpyAsync(ll, tmp.p + 2 * (size_t)D, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

int objective(mmm_ctx* ctx, int mode, int n, const double* x, const double* other, const double* c, const double* sumth, const double* mu,
              const double* S, double* val, double* grad)
{
    const char* who = mode == 0 ? "mmm_lambda_objective" : "mmm_nu_objective";
    if (!ctx) return MMM_ERR_ARG;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    MMM_CHECK(ctx, n >= 1 && x && other && c && S && val && (mode == 1 || (sumth && mu)), "%s: NULL argument or n < 1", who);
    const size_t N = (size_t)n;
    // one upload: [x | other | c | sumth | mu | S], then [grad | 3n addends | value]
    std::vector<double> h(5 * N + N * N, 0.0);
    memcpy(h.data(), x, 8 * N); memcpy(h.data() + N, other, 8 * N); memcpy(h.data() + 2 * N, c, 8 * N);
    if (sumth) memcpy(h.data() + 3 * N, sumth, 8 * N);
    if (mu) memcpy(h.data() + 4 * N, mu, 8 * N);
    memcpy(h.data() + 5 * N, S, 8 * N * N);
    DevBuf<double> in, work;
    MMM_HIP(ctx, in.alloc(h.size())); MMM_HIP(ctx, work.alloc(4 * N + 1));
    MMM_HIP(ctx, hipMemcpyAsync(in.p, h.data(), 8 * h.size(), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_free_objective, dim3(1), dim3(256), 0, ctx->stream, mode, n, in.p, in.p + N, in.p + 2 * N, in.p + 3 * N, in.p + 4 * N, in.p + 5 * N,
                       work.p, work.p + 4 * N);
    MMM_LAUNCH_CHECK(ctx);
    MMM_HIP(ctx, hipMemcpyAsync(val, work.p + 4 * N, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (grad) MMM_HIP(ctx, hipMemcpyAsync(grad, work.p, 8 * N, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MMM_OK;
}

} // namespace

extern "C" {

int mmm_lambda_objective(mmm_ctx* ctx, int n, const double* lambda, const double* nu, const double* Ndivzeta, const double* sumtheta, const double* mu,
                         const double* invSigma, double* val, double* grad)
{
    return objective(ctx, 0, n, lambda, nu, Ndivzeta, sumtheta, mu, invSigma, val, grad);
}

int mmm_nu_objective(mmm_ctx* ctx, int n, const double* nu, const double* lambda, const double* Ndivzeta, const double* mu, const double* invSigma,
                     double* val, double* grad)
{
    return objective(ctx, 1, n, nu, lambda, Ndivzeta, nullptr, mu, invSigma, val, grad);
}

int mmm_alpha_objective(mmm_ctx* ctx, double alpha, double sum_Elnphi, int K, int V, double* val, double* grad)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    MMM_CHECK(ctx, val && alpha > 0.0 && K >= 1 && V >= 1, "mmm_alpha_objective: val == NULL, alpha <= 0, K < 1 or V < 1");
    DevBuf<double> out;
    MMM_HIP(ctx, out.alloc(2));
    hipLaunchKernelGGL(k_free_alpha_objective, dim3(1), dim3(1), 0, ctx->stream, alpha, sum_Elnphi, (double)K, (double)V, out.p);
    MMM_LAUNCH_CHECK(ctx);
    double h[2];
    MMM_HIP(ctx, hipMemcpyAsync(h, out.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    MMM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *val = h[0];
    if (grad) *grad = h[1];
    return MMM_OK;
}

int mmm_mixture_loglik(mmm_ctx* ctx, int D, int K, int V, const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const double* props,
                       const double* phi, double* ll)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    MMM_CHECK(ctx, K >= 1 && V >= 1 && props && phi && ll, "mmm_mixture_loglik: NULL argument, K < 1 or V < 1");
    if (int rc = check_csr(ctx, "mmm_mixture_loglik", D, V, doc_ptr, term, count)) return rc;
    DevBuf<double> p, f;
    MMM_HIP(ctx, p.alloc((size_t)K * D)); MMM_HIP(ctx, f.alloc((size_t)K * V));
    if (D) MMM_HIP(ctx, hipMemcpyAsync(p.p, props, sizeof(double) * K * D, hipMemcpyHostToDevice, ctx->stream));
    MMM_HIP(ctx, hipMemcpyAsync(f.p, phi, sizeof(double) * K * V, hipMemcpyHostToDevice, ctx->stream));
    return loglik_from_tables(ctx, D, K, V, doc_ptr, term, count, p.p, f.p, ll);
}

int mmm_mixture_loglik_features(mmm_ctx* ctx, int D, int K, int V, int I, const int* J, const int32_t* features, const int64_t* doc_ptr,
                                const int32_t* term, const int32_t* count, const double* eta, int softmax, const double* phi, double* ll)
{
    if (!ctx) return MMM_ERR_ARG;
    MMM_HIP(ctx, hipSetDevice(ctx->device));
    MMM_CHECK(ctx, K >= 1 && V >= 1 && I >= 1 && J && features && eta && phi && ll, "mmm_mixture_loglik_features: NULL argument, K < 1, V < 1 or I < 1");
    if (int rc = check_csr(ctx, "mmm_mixture_loglik_features", D, V, doc_ptr, term, count)) return rc;
    std::vector<int> joff((size_t)I + 1, 0);
    for (int i = 0; i < I; ++i) { MMM_CHECK(ctx, J[i] >= 1, "mmm_mixture_loglik_features: J[%d] < 1", i); joff[i + 1] = joff[i] + J[i]; }
    for (int i = 0; i < I; ++i)
        for (int v = 0; v < V; ++v)
            MMM_CHECK(ctx, features[(size_t)i * V + v] >= 0 && features[(size_t)i * V + v] < J[i], "mmm_mixture_loglik_features: feature %d of term %d out of range", i, v);
    const int SJ = joff[I];
    DevBuf<double> e, p, f, fe; DevBuf<int> jo; DevBuf<int32_t> ft;
    MMM_HIP(ctx, e.alloc((size_t)K * D)); MMM_HIP(ctx, p.alloc((size_t)K * D)); MMM_HIP(ctx, f.alloc((size_t)K * SJ)); MMM_HIP(ctx, fe.alloc((size_t)K * V));
    MMM_HIP(ctx, jo.alloc((size_t)I + 1)); MMM_HIP(ctx, ft.alloc((size_t)I * V));
    if (D) MMM_HIP(ctx, hipMemcpyAsync(e.p, eta, sizeof(double) * K * D, hipMemcpyHostToDevice, ctx->stream));
    MMM_HIP(ctx, hipMemcpyAsync(f.p, phi, sizeof(double) * K * SJ, hipMemcpyHostToDevice, ctx->stream));
    MMM_HIP(ctx, hipMemcpyAsync(jo.p, joff.data(), sizeof(int) * (I + 1), hipMemcpyHostToDevice, ctx->stream));
    MMM_HIP(ctx, hipMemcpyAsync(ft.p, features, sizeof(int32_t) * I * V, hipMemcpyHostToDevice, ctx->stream));
    if (D && softmax) hipLaunchKernelGGL(k_free_softmax, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, D, K, e.p, p.p);
    hipLaunchKernelGGL(k_free_phieff, dim3((unsigned)((K * V + 255) / 256)), dim3(256), 0, ctx->stream, K, V, I, SJ, jo.p, ft.p, f.p, fe.p);
    MMM_LAUNCH_CHECK(ctx);
    return loglik_from_tables(ctx, D, K, V, doc_ptr, term, count, softmax ? p.p : e.p, fe.p, ll);
}

} // extern "C"
