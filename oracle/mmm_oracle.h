/*
 * mmm_oracle.h -- CPU ORACLE for the variational-EM hot path of MultiModalMuSig.jl.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's Julia algorithm
 * (src/LDA.jl, src/MMCTM.jl, src/IMMCTM.jl, src/common.jl) plus a restatement of the two third-party
 * pieces the reference calls but does not vendor (NLopt `LD_MMA`, SpecialFunctions `digamma`).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (multimodalmusig.jl_amd/, libmmmusig_hip.so) never links, imports or calls anything in here.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - closed-form updates (phi/gamma/lambda of LDA; zeta/theta/mu/Sigma/gamma/Elnphi/loglik of MMCTM
 *     and IMMCTM; the lambda/nu objectives and gradients): PINNED by the reference's own known-answer
 *     tests (test/lda.jl, test/mmctm.jl, test/immctm.jl, test/common.jl) -> tests/golden/ (JSON).
 *   - MMA optimiser outputs (lambda, nu after update_λ!/update_ν!), full fit! trajectories and every
 *     ELBO value: "PARITY UNPINNED" -- the reference's tests assert only qualitative facts there and the
 *     Julia reference + libnlopt cannot be executed in the build container (no julia, no NLopt).
 *
 * Layout conventions (shared with include/mmmusig.h):
 *   all reals are double, term indices are 0-based int32, counts int32, CSR offsets int64.
 *   LDA:    lambda/Elnbeta/beta  V x K column-major  -> [v + V*k]
 *           gamma/Elntheta/theta K x D column-major  -> [k + K*d]
 *           phi                  per doc K x W_d (k fastest), docs concatenated -> [K*doc_ptr[d] + k + K*w]
 *   MMCTM:  doc_ptr              M*(D+1) absolute offsets into term/count (modality-major concatenation)
 *           lambda/nu/props      MK x D -> [i + MK*d];  zeta M x D -> [m + M*d]
 *           gamma/Elnphi/phi     [goff[m] + k*V[m] + v],  goff[m] = sum_{m'<m} K[m']*V[m']
 *           theta                [toff[m] + (e - doc_ptr[m*(D+1)])*K[m] + k] for absolute entry e of
 *                                modality m, toff[m] = sum_{m'<m} nnz[m']*K[m']
 *           mu MK; Sigma/invSigma MK x MK column-major
 *   IMMCTM: features             [foff[m] + i*V[m] + v] (0-based values), foff[m] = sum_{m'<m} I[m']*V[m']
 *           gamma/Elnphi         [goff[m] + k*SJ[m] + joff[m][i] + j], SJ[m] = sum_i J[m][i],
 *                                goff[m] = sum_{m'<m} K[m']*SJ[m'];  alpha [aoff[m] + i], aoff = sum I
 */
#ifndef MMM_ORACLE_H
#define MMM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar math ---------------------------------------------------------------------------- */
double orc_digamma(double x);                 /* SpecialFunctions.jl digamma algorithm           */
double orc_lgamma(double x);                  /* logabsgamma(x)[1]                                */
void   orc_digamma_vec(int n, const double* x, double* out);
double orc_logmvbeta(int n, const double* vals);        /* common.jl:1-9 */

/* ---- NLopt LD_MMA (m = 0 constraints) restatement ---------------------------------------------- */
typedef double (*orc_objective)(int n, const double* x, double* grad, void* data);
/* xtol_rule: 0 = NLopt >= 2.7 (L1 norm rule), 1 = NLopt <= 2.6 (per-coordinate rule).
 * minimises f. lb/ub may be NULL (= -inf/+inf). Returns number of objective evaluations (>0) or
 * -1 when the evaluation cap was hit. x is overwritten with the best accepted point. */
void orc_mma_set_trace(double* buf, int cap_rows);   /* test hook: one row [rho, gval, wval, fcur, sigma[n], xcur[n]] per inner iteration */
int orc_mma_trace_rows(void);
int orc_mma_minimize(int n, orc_objective f, void* data, const double* lb, const double* ub,
                     double* x, double* minf, double xtol_rel, double xtol_abs, int xtol_rule,
                     int max_eval, int* n_outer);

/* common.jl:11-23 / 25-36 -- value returned, gradient written when grad != NULL (objective is MAXIMISED) */
double orc_lambda_objective(int n, const double* lambda, double* grad, const double* nu,
                            const double* Ndivzeta, const double* sumtheta, const double* mu,
                            const double* invSigma);
double orc_nu_objective(int n, const double* nu, double* grad, const double* lambda,
                        const double* Ndivzeta, const double* mu, const double* invSigma);

/* ---- LDA (src/LDA.jl) -------------------------------------------------------------------------- */
void orc_lda_update_Elntheta(int K, int D, const double* gamma, double* Elntheta);
void orc_lda_update_gamma(int K, int D, double alpha, const int64_t* doc_ptr, const int32_t* count,
                          const double* phi, double* gamma, double* Elntheta);
void orc_lda_update_phi(int K, int D, int V, const int64_t* doc_ptr, const int32_t* term,
                        const double* Elntheta, const double* Elnbeta, double* phi);
void orc_lda_update_Elnbeta(int V, int K, const double* lambda, double* Elnbeta);
void orc_lda_update_lambda(int K, int D, int V, double eta, const int64_t* doc_ptr,
                           const int32_t* term, const int32_t* count, const double* phi,
                           double* lambda, double* Elnbeta);
void orc_lda_update_beta(int V, int K, const double* lambda, double* beta);
void orc_lda_update_theta(int K, int D, const double* gamma, double* theta);
double orc_lda_loglik(int K, int D, int V, const int64_t* doc_ptr, const int32_t* term,
                      const int32_t* count, const double* theta, const double* beta);
/* terms[7] = ElnPbeta, ElnPtheta, ElnPZ, ElnPX, ElnQbeta, ElnQtheta, ElnQZ; returns the ELBO */
double orc_lda_elbo(int K, int D, int V, double alpha, double eta, const int64_t* doc_ptr,
                    const int32_t* term, const int32_t* count, const double* lambda,
                    const double* Elnbeta, const double* gamma, const double* Elntheta,
                    const double* phi, double* terms);
/* constructor (LDA.jl:24-54) + fit! (LDA.jl:198-224). lambda is in/out (lambda0 on entry). */
int orc_lda_fit(int D, int V, int K, double alpha, double eta, const int64_t* doc_ptr,
                const int32_t* term, const int32_t* count, int maxiter, double tol,
                double* lambda, double* gamma, double* Elntheta, double* theta, double* Elnbeta,
                double* beta, double* phi, double* ll_hist, int* n_iter, int* converged,
                double* elbo);

/* frozen-topic inference: unsmoothed_update_ϕ! (LDA.jl:226-231) and the loops of transform (LDA.jl:233-263) /
 * fit_heldout (LDA.jl:265-295) on constructor state with the trained topics (Elnbeta, beta) given */
void orc_lda_unsmoothed_update_phi(int K, int D, int V, const int64_t* doc_ptr, const int32_t* term,
                                   const double* Elntheta, const double* beta, double* phi);
int orc_lda_infer(int D, int V, int K, double alpha, const int64_t* doc_ptr, const int32_t* term,
                  const int32_t* count, int unsmoothed, int maxiter, double tol, const double* Elnbeta,
                  const double* beta, double* gamma, double* Elntheta, double* theta, double* phi,
                  double* ll_hist, int* n_iter, int* converged);

/* ---- ILDA (src/ILDA.jl): lambda[i] J_i x K column-major at K*sum_{q<i} J_q; features [i*V + v], 0-based -------------- */
void orc_ilda_update_Elnbeta(int K, int I, const int* J, const double* lam, double* Eln);
void orc_ilda_update_beta(int K, int I, const int* J, const double* lam, double* beta);
void orc_ilda_update_phi(int K, int D, int V, int I, const int* J, const int32_t* features, const int64_t* doc_ptr,
                         const int32_t* term, const double* Elntheta, const double* Eln_or_beta, int unsmoothed, double* phi);
void orc_ilda_update_lambda(int K, int D, int V, int I, const int* J, const double* eta, const int32_t* features,
                            const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const double* phi,
                            double* lam, double* Eln);
double orc_ilda_loglik(int K, int D, int V, int I, const int* J, const int32_t* features, const int64_t* doc_ptr,
                       const int32_t* term, const int32_t* count, const double* theta, const double* beta);
double orc_ilda_elbo(int K, int D, int V, int I, const int* J, double alpha, const double* eta, const int32_t* features,
                     const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const double* lam, const double* Eln,
                     const double* gamma, const double* Elntheta, const double* phi, double* terms);
int orc_ilda_fit(int D, int V, int K, int I, const int* J, double alpha, const double* eta, const int32_t* features,
                 const int64_t* doc_ptr, const int32_t* term, const int32_t* count, int frozen, int maxiter, double tol,
                 double* lam, double* Eln, double* beta, double* gamma, double* Elntheta, double* theta, double* phi,
                 double* ll_hist, int* n_iter, int* converged, double* elbo);

/* ---- MMCTM / IMMCTM (src/MMCTM.jl, src/IMMCTM.jl) ------------------------------------------------ */
typedef struct {
    int D, M, MK;
    const int* K;             /* M */
    const int* V;             /* M */
    const int64_t* doc_ptr;   /* M*(D+1) absolute */
    const int32_t* term;
    const int32_t* count;
    /* independent-feature extension (IMMCTM); n_feat == NULL for plain MMCTM */
    const int* n_feat;        /* I[m] */
    const int* J;             /* concatenated J[m][i] */
    const int32_t* features;  /* see header comment */
    double* alpha;            /* MMCTM: M ; IMMCTM: sum_m I[m] */
    /* globals */
    double* mu; double* Sigma; double* invSigma;
    double* gamma; double* Elnphi; double* phi;   /* phi unused (may be NULL) for IMMCTM */
    /* per document */
    double* lambda; double* nu; double* zeta; double* props; double* theta;
    /* solver configuration (MMCTM.jl:127-130,156-160) */
    double xtol_rel, xtol_abs, nu_lower; int xtol_rule; int max_eval;
    /* statistics out */
    int64_t n_eval_lambda, n_eval_nu, n_solver_cap;
    /* order-matched variant (mmm_twin.c).  arith = 0: every sum in index order (mmm_oracle.c); arith = 1: every sum
     * associated as the gfx950 kernels associate it, exp/log/digamma from csrc/mmm_arith.h.  The sums across documents
     * follow the launch geometry of the device handle (mmm_ctm_geometry): L lanes per document, grid_e blocks of waves_e
     * waves in the theta phase, grid_m blocks in the moment sums. */
    int arith;
    int L, grid_e, waves_e, grid_m;
    int* nev_nu; int* nev_lambda;   /* objective evaluations per document in the last pass (either variant; may be NULL) */
    double* expE;                   /* arith = 1: exp(Elnphi) in the effective [m][k][v] layout, sum_m K_m V_m doubles */
    int Ls;                         /* lanes per document in the solve phase: L, or sum K for the packed device builds (0: = L) */
    int cpl;                        /* coordinates per lane in the solve phase (device builds with Ls * cpl = sum K; 0 / 1: one) */
    int tdense;                     /* 1: the fused pass's theta phase runs over rows of counts on the device (k_ctm_theta_dense) */
} orc_ctm;

/* the pieces of one pass in device order (mmm_twin.c) */
void orc_twin_topics(orc_ctm* m, const double* sG);
void orc_twin_estep(orc_ctm* m, double* sG);
void orc_twin_estep_fused(orc_ctm* m, double* sG);     /* the E-step of a fused pass: over rows of counts when m->tdense */
void orc_twin_moments(const orc_ctm* m, double* mom);
int  orc_twin_gauss(orc_ctm* m, const double* mom, int do_sigma);
int  orc_twin_pass(orc_ctm* m, int update_sigma);
void orc_twin_tables_from_Elnphi(orc_ctm* m);
int  orc_twin_infer_pass(orc_ctm* m, int flags);
void orc_twin_objectives(int n, const double* lambda, const double* nu, const double* Ndivzeta, const double* sumtheta,
                         const double* mu, const double* invSigma, double* vals, double* grad_lambda, double* grad_nu);
void orc_ar_exp_vec(int n, const double* x, double* out);
void orc_ar_log_vec(int n, const double* x, double* out);
void orc_ar_digamma_vec(int n, const double* x, double* out);
void orc_ar_exptab_vec(int n, const double* x, double* out);      /* ar_exp_tab / ar_log_tab: the table-driven exp / log of the LD_MMA objectives */
void orc_ar_logtab_vec(int n, const double* x, double* out);
void orc_ar_digammatab_vec(int n, const double* x, double* out);  /* ar_digamma_pos_tab: the LDA dense-row prologue's digamma over the log table */
/* debug hook: [exp min, exp max, log min, log max] of the arguments the order-matched objectives have passed to the table-driven exp / log
 * since the last reset (tests/golden/make_table_ranges.py records them for configs 3-5) */
void orc_twin_arg_ranges(double out[4], int reset);

void orc_ctm_update_zeta(orc_ctm* m, int d);
void orc_ctm_update_theta(orc_ctm* m, int d);
void orc_ctm_update_nu(orc_ctm* m, int d);
void orc_ctm_update_lambda(orc_ctm* m, int d);
void orc_ctm_fitdoc(orc_ctm* m, int d);
void orc_ctm_calc_sumtheta(const orc_ctm* m, int d, double* out);
void orc_ctm_calc_Ndivzeta(const orc_ctm* m, int d, double* out);
void orc_ctm_update_mu(orc_ctm* m);
int  orc_ctm_update_Sigma(orc_ctm* m);
void orc_ctm_update_Elnphi(orc_ctm* m);
void orc_ctm_update_gamma(orc_ctm* m);
void orc_ctm_update_props(orc_ctm* m);
void orc_ctm_update_phi(orc_ctm* m);
void orc_ctm_loglik(const orc_ctm* m, double* ll /* M */);
double orc_ctm_elbo(const orc_ctm* m, double* terms /* 7 */);
/* the E-step over a doc range (used by sharding tests) */
void orc_ctm_estep_range(orc_ctm* m, int d0, int d1);
/* fit!: MMCTM.jl:457-494 / IMMCTM.jl:437-466. State must be constructor-initialised by the caller
 * via orc_ctm_init (lambda=0, nu=1, mu=0, Sigma=I, theta=1/K, Elnphi from gamma, zeta). */
void orc_ctm_init(orc_ctm* m);
double orc_alpha_objective(double alpha, double* grad, double sum_Elnphi, int K, int V);   /* common.jl:38-46 */
void orc_ctm_update_alpha(orc_ctm* m);                     /* update_α!  MMCTM.jl:252-269 / IMMCTM.jl:225-244 */
int orc_ctm_fit(orc_ctm* m, int maxiter, double tol, int update_sigma, int auto_alpha, double* ll_hist /* M*maxiter */,
                int* n_iter, int* converged, double* elbo);

/* frozen-topic inference: unsmoothed_update_θ! (MMCTM.jl:496-509); loops of transform (flags & 1 unsmoothed theta,
 * flags & 2 fit_gaussian; MMCTM.jl:511-552) and fit_heldout / predict_modality_η (flags 0; MMCTM.jl:554-634,
 * IMMCTM.jl:468-545) */
void orc_ctm_unsmoothed_update_theta(orc_ctm* m, int d);
int orc_ctm_infer(orc_ctm* m, int flags, int maxiter, double tol, double* ll_hist /* M*maxiter */, int* n_iter,
                  int* converged);

/* dense helper: inverse + log|det| of a column-major n x n matrix by LU with partial pivoting */
int orc_inv_logdet(int n, const double* A, double* Ainv, double* logabsdet, int* sign);

#ifdef __cplusplus
}
#endif
#endif
