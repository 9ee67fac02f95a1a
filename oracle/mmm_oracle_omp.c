/*
 * mmm_oracle_omp.c -- TEST INFRASTRUCTURE ONLY: the LDA iteration of mmm_oracle.c with its document loops run by OpenMP
 * threads.  It exists for one purpose: bench.py's `cpu_baseline_all_cores` (the CPU figure on all host cores
 * beside the single-thread one).  The arithmetic per document is the sequential oracle's; only the lambda statistics and the
 * log-likelihood are summed per thread first (different summation order, same values to ~1e-13; tests/test_oracle_crosscheck.py).
 * Reference lines: LDA.jl:69-112,174-188,201-209.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mmm_oracle.h"

int orc_omp_threads(void) { return omp_get_max_threads(); }
void orc_omp_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* one pass of the body of fit! (LDA.jl:201-209); returns the log-likelihood */
double orc_lda_pass_omp(int D, int V, int K, double alpha, double eta, const int64_t* doc_ptr, const int32_t* term,
                        const int32_t* count, double* lambda, double* Elnbeta, double* beta, double* gamma, double* Elntheta,
                        double* theta, double* phi)
{
    const int nt = omp_get_max_threads();
    double* acc = (double*)calloc((size_t)nt * V * K, sizeof(double));
    double ll = 0.0; int64_t N = 0;
#pragma omp parallel
    {
        double* mine = acc + (size_t)omp_get_thread_num() * V * K;
#pragma omp for schedule(static)
        for (int d = 0; d < D; ++d) {
            double* ph = phi + (size_t)K * doc_ptr[d];
            const int64_t W = doc_ptr[d + 1] - doc_ptr[d];
            double s = 0.0;
            /* update_γ! LDA.jl:82-90 (+ Elnθ :78-80) */
            for (int k = 0; k < K; ++k) {
                double g = alpha;
                for (int64_t w = 0; w < W; ++w) g += ph[k + (size_t)K * w] * (double)count[doc_ptr[d] + w];
                gamma[k + (size_t)K * d] = g; s += g;
            }
            const double ps = orc_digamma(s);
            for (int k = 0; k < K; ++k) Elntheta[k + (size_t)K * d] = orc_digamma(gamma[k + (size_t)K * d]) - ps;
            /* update_ϕ! LDA.jl:69-76, λ statistics LDA.jl:103-105 */
            for (int64_t w = 0; w < W; ++w) {
                const int v = term[doc_ptr[d] + w];
                double t = 0.0;
                for (int k = 0; k < K; ++k) { double e = exp(Elntheta[k + (size_t)K * d] + Elnbeta[v + (size_t)V * k]); ph[k + (size_t)K * w] = e; t += e; }
                for (int k = 0; k < K; ++k) {
                    ph[k + (size_t)K * w] /= t;
                    mine[v + (size_t)V * k] += ph[k + (size_t)K * w] * (double)count[doc_ptr[d] + w];
                }
            }
            for (int k = 0; k < K; ++k) theta[k + (size_t)K * d] = gamma[k + (size_t)K * d] / s;     /* update_θ! :92-94 */
        }
    }
    /* update_λ!, Elnβ, β  (LDA.jl:96-112) */
    for (size_t i = 0; i < (size_t)V * K; ++i) { double s = eta; for (int t = 0; t < nt; ++t) s += acc[(size_t)t * V * K + i]; lambda[i] = s; }
    orc_lda_update_Elnbeta(V, K, lambda, Elnbeta);
    orc_lda_update_beta(V, K, lambda, beta);
    /* calculate_loglikelihood LDA.jl:174-188 */
#pragma omp parallel for schedule(static) reduction(+ : ll, N)
    for (int d = 0; d < D; ++d)
        for (int64_t e = doc_ptr[d]; e < doc_ptr[d + 1]; ++e) {
            double p = 0.0;
            for (int k = 0; k < K; ++k) p += theta[k + (size_t)K * d] * beta[term[e] + (size_t)V * k];
            ll += (double)count[e] * log(p);
            N += count[e];
        }
    free(acc);
    return ll / (double)N;
}
