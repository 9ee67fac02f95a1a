// ctm.hip -- PLACEHOLDER while the CTM kernels are being written: every entry point reports MMM_ERR_UNSUPPORTED.
#include "mmm_internal.h"
struct mmm_ctm { mmm_ctx* ctx; };
extern "C" {
int mmm_ctm_create(mmm_ctx* ctx, int D, int M, const int* K, const int* V, const double* alpha, const int64_t* doc_ptr, const int32_t* term, const int32_t* count, const int* n_feat, const int* J, const int32_t* features, const double* gamma0, const mmm_solver_opts* opts, mmm_ctm** out) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_destroy(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_get(mmm_ctm* m, int field, double* host, size_t n) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_set(mmm_ctm* m, int field, const double* host, size_t n) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_zeta(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_theta(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_nu(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_lambda(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_mu(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_Sigma(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_gamma(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_Elnphi(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_props(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_update_phi(mmm_ctm* m) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_loglik(mmm_ctm* m, double* ll /* M */) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_elbo(mmm_ctm* m, double* elbo, double terms[7]) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_objectives(mmm_ctm* m, int d, double* lambda_val, double* lambda_grad, double* nu_val, double* nu_grad) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_solver_stats(mmm_ctm* m, int64_t* n_eval_nu, int64_t* n_eval_lambda, int64_t* n_capped, int* per_doc_nu, int* per_doc_lambda) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_iterate(mmm_ctm* m, int n_iter, int update_sigma) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_ll_history(mmm_ctm* m, double* ll /* M*max_n */, int max_n, int* n) { return MMM_ERR_UNSUPPORTED; }
int mmm_ctm_fit(mmm_ctm* m, int maxiter, double tol, int update_sigma, double* ll_hist /* M*maxiter */, int* n_iter, int* converged, double* elbo) { return MMM_ERR_UNSUPPORTED; }
}
