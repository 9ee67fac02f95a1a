/*
 * lda_fit.c -- the C ABI of libmmmusig_hip.so driven from plain C (no Python, no Julia): what a binding in any host language
 * does for `model = LDA(K, α, η, V, X); fit!(model)` (reference: src/LDA.jl:24-54, 198-224).
 *
 *   cc -std=c99 -Iinclude examples/lda_fit.c -Lmultimodalmusig.jl_amd/lib -lmmmusig_hip -Wl,-rpath,$PWD/multimodalmusig.jl_amd/lib -lm -o lda_fit
 *   ./lda_fit counts.tsv K [maxiter [tol [seed]]]
 *
 * counts.tsv: the layout of the reference's data/ tables -- a header `term<TAB>sample1<TAB>...`, one row per vocabulary term.
 * Prints the log-likelihood history, the ELBO and the first document's topic proportions as one JSON object.
 * λ0 is drawn by a small LCG in 1..100 (the reference draws `rand(1:100, V, K)`, LDA.jl:36); the test that runs this program
 * reproduces the same λ0 on the Python side and compares the results bit for bit.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mmmusig.h"

static int fail(mmm_ctx* ctx, const char* what)
{
    fprintf(stderr, "%s: %s\n", what, mmm_last_error(ctx));
    return 1;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s counts.tsv K [maxiter [tol [seed]]]\n", argv[0]); return 2; }
    const int K = atoi(argv[2]);
    const int maxiter = argc > 3 ? atoi(argv[3]) : 1000;
    const double tol = argc > 4 ? atof(argv[4]) : 1e-4;
    uint64_t lcg = argc > 5 ? (uint64_t)strtoull(argv[5], NULL, 10) : 1u;

    /* ---- read the table: V rows x D columns of counts ------------------------------------------------------------ */
    FILE* fh = fopen(argv[1], "r");
    if (!fh) { perror(argv[1]); return 2; }
    size_t cap = 1 << 20;
    char* line = (char*)malloc(cap);
    if (!fgets(line, (int)cap, fh)) { fprintf(stderr, "empty table\n"); return 2; }
    int D = 0;
    for (char* p = line; *p; ++p) if (*p == '\t') ++D;
    int V = 0, vcap = 256;
    int32_t* table = (int32_t*)malloc(sizeof(int32_t) * (size_t)vcap * D);
    while (fgets(line, (int)cap, fh)) {
        char* p = strchr(line, '\t');
        if (!p) continue;
        if (V == vcap) { vcap *= 2; table = (int32_t*)realloc(table, sizeof(int32_t) * (size_t)vcap * D); }
        for (int d = 0; d < D; ++d) { table[(size_t)V * D + d] = (int32_t)strtod(p + 1, &p); }
        ++V;
    }
    fclose(fh);

    /* ---- CSR, zero counts dropped (format_counts_lda, src/utils.jl:9-18): 0-based terms, int64 offsets ---------------- */
    int64_t* doc_ptr = (int64_t*)calloc((size_t)D + 1, sizeof(int64_t));
    for (int d = 0; d < D; ++d) {
        int64_t w = 0;
        for (int v = 0; v < V; ++v) w += table[(size_t)v * D + d] > 0;
        doc_ptr[d + 1] = doc_ptr[d] + w;
    }
    const int64_t nnz = doc_ptr[D];
    int32_t* term = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    int32_t* count = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    for (int d = 0; d < D; ++d) {
        int64_t e = doc_ptr[d];
        for (int v = 0; v < V; ++v)
            if (table[(size_t)v * D + d] > 0) { term[e] = v; count[e] = table[(size_t)v * D + d]; ++e; }
    }

    /* ---- λ0 in 1..100, V x K column-major ------------------------------------------------------------------------- */
    double* lambda0 = (double*)malloc(sizeof(double) * (size_t)V * K);
    for (size_t i = 0; i < (size_t)V * K; ++i) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        lambda0[i] = (double)(1 + (lcg >> 33) % 100);
    }

    /* ---- the calls a binding makes ---------------------------------------------------------------------------------- */
    mmm_ctx* ctx = NULL;
    if (mmm_ctx_create(0, &ctx) != MMM_OK) return fail(NULL, "mmm_ctx_create");
    mmm_lda* model = NULL;
    if (mmm_lda_create(ctx, D, V, K, 0.1, 0.1, doc_ptr, term, count, lambda0, &model) != MMM_OK) return fail(ctx, "mmm_lda_create");
    double* ll = (double*)malloc(sizeof(double) * (size_t)maxiter);
    int n_iter = 0, converged = 0;
    double elbo = 0.0;
    if (mmm_lda_fit(model, maxiter, tol, ll, &n_iter, &converged, &elbo) != MMM_OK) return fail(ctx, "mmm_lda_fit");
    double* theta = (double*)malloc(sizeof(double) * (size_t)K * D);
    if (mmm_lda_get(model, MMM_LDA_THETA, theta, (size_t)K * D) != MMM_OK) return fail(ctx, "mmm_lda_get");

    printf("{\"D\": %d, \"V\": %d, \"K\": %d, \"nnz\": %lld, \"n_iter\": %d, \"converged\": %d, \"elbo\": %.17g, \"ll\": [", D, V, K, (long long)nnz,
           n_iter, converged, elbo);
    for (int i = 0; i < n_iter; ++i) printf("%s%.17g", i ? ", " : "", ll[i]);
    printf("], \"theta_doc1\": [");
    for (int k = 0; k < K; ++k) printf("%s%.17g", k ? ", " : "", theta[k]);
    printf("]}\n");

    mmm_lda_destroy(model);
    mmm_ctx_destroy(ctx);
    free(theta); free(ll); free(lambda0); free(count); free(term); free(doc_ptr); free(table); free(line);
    return 0;
}
