"""ctypes front-end of the CPU oracle (oracle/mmm_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  Nothing under multimodalmusig.jl_amd/ imports this module.

All arrays are numpy, laid out as documented at the top of oracle/mmm_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "build", "libmmm_oracle.so")
    so_omp = os.path.join(_HERE, "build", "libmmm_oracle_omp.so")
    src = [os.path.join(_HERE, f) for f in ("mmm_oracle.c", "mmm_oracle.h", "mmm_oracle_omp.c", "mmm_twin.c")]
    src += [os.path.join(_HERE, "..", "multimodalmusig.jl_amd", "csrc", h) for h in ("mmm_arith.h", "mmm_exptab.h", "mmm_logtab.h")]
    stale = (not os.path.exists(so)) or (not os.path.exists(so_omp)) or any(os.path.getmtime(s) > min(os.path.getmtime(so), os.path.getmtime(so_omp)) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return so


class OrcCtm(C.Structure):
    _fields_ = [
        ("D", C.c_int), ("M", C.c_int), ("MK", C.c_int),
        ("K", C.c_void_p), ("V", C.c_void_p), ("doc_ptr", C.c_void_p), ("term", C.c_void_p),
        ("count", C.c_void_p), ("n_feat", C.c_void_p), ("J", C.c_void_p), ("features", C.c_void_p),
        ("alpha", C.c_void_p), ("mu", C.c_void_p), ("Sigma", C.c_void_p), ("invSigma", C.c_void_p),
        ("gamma", C.c_void_p), ("Elnphi", C.c_void_p), ("phi", C.c_void_p),
        ("lambda_", C.c_void_p), ("nu", C.c_void_p), ("zeta", C.c_void_p), ("props", C.c_void_p),
        ("theta", C.c_void_p),
        ("xtol_rel", C.c_double), ("xtol_abs", C.c_double), ("nu_lower", C.c_double),
        ("xtol_rule", C.c_int), ("max_eval", C.c_int),
        ("n_eval_lambda", C.c_int64), ("n_eval_nu", C.c_int64), ("n_solver_cap", C.c_int64),
        ("arith", C.c_int), ("L", C.c_int), ("grid_e", C.c_int), ("waves_e", C.c_int), ("grid_m", C.c_int),
        ("nev_nu", C.c_void_p), ("nev_lambda", C.c_void_p), ("expE", C.c_void_p), ("Ls", C.c_int), ("cpl", C.c_int), ("tdense", C.c_int),
    ]


OBJ_CB = C.CFUNCTYPE(C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # MMM_ORACLE_SO: another build of the same sources (`make -C oracle asan`, run under LD_PRELOAD=libasan.so: DESIGN section 2)
    L = C.CDLL(os.environ.get("MMM_ORACLE_SO") or build())
    L.orc_digamma.restype = C.c_double; L.orc_digamma.argtypes = [C.c_double]
    L.orc_lgamma.restype = C.c_double; L.orc_lgamma.argtypes = [C.c_double]
    L.orc_digamma_vec.argtypes = [C.c_int, f64p, f64p]
    L.orc_logmvbeta.restype = C.c_double; L.orc_logmvbeta.argtypes = [C.c_int, f64p]
    L.orc_inv_logdet.argtypes = [C.c_int, f64p, f64p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.orc_mma_minimize.restype = C.c_int
    L.orc_mma_set_trace.restype = None; L.orc_mma_set_trace.argtypes = [C.c_void_p, C.c_int]
    L.orc_mma_trace_rows.restype = C.c_int; L.orc_mma_trace_rows.argtypes = []
    L.orc_mma_minimize.argtypes = [C.c_int, OBJ_CB, C.c_void_p, C.c_void_p, C.c_void_p, f64p,
                                   C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.POINTER(C.c_int)]
    L.orc_lambda_objective.restype = C.c_double
    L.orc_lambda_objective.argtypes = [C.c_int, f64p, C.c_void_p, f64p, f64p, f64p, f64p, f64p]
    L.orc_nu_objective.restype = C.c_double
    L.orc_nu_objective.argtypes = [C.c_int, f64p, C.c_void_p, f64p, f64p, f64p, f64p]
    # LDA
    L.orc_lda_update_Elntheta.argtypes = [C.c_int, C.c_int, f64p, f64p]
    L.orc_lda_update_gamma.argtypes = [C.c_int, C.c_int, C.c_double, i64p, i32p, f64p, f64p, f64p]
    L.orc_lda_update_phi.argtypes = [C.c_int, C.c_int, C.c_int, i64p, i32p, f64p, f64p, f64p]
    L.orc_lda_update_Elnbeta.argtypes = [C.c_int, C.c_int, f64p, f64p]
    L.orc_lda_update_lambda.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, i64p, i32p, i32p, f64p, f64p, f64p]
    L.orc_lda_update_beta.argtypes = [C.c_int, C.c_int, f64p, f64p]
    L.orc_lda_update_theta.argtypes = [C.c_int, C.c_int, f64p, f64p]
    L.orc_lda_loglik.restype = C.c_double
    L.orc_lda_loglik.argtypes = [C.c_int, C.c_int, C.c_int, i64p, i32p, i32p, f64p, f64p]
    L.orc_lda_elbo.restype = C.c_double
    L.orc_lda_elbo.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i64p, i32p, i32p,
                               f64p, f64p, f64p, f64p, f64p, f64p]
    L.orc_lda_fit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i64p, i32p, i32p,
                              C.c_int, C.c_double, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p,
                              C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.orc_lda_unsmoothed_update_phi.argtypes = [C.c_int, C.c_int, C.c_int, i64p, i32p, f64p, f64p, f64p]
    L.orc_lda_infer.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, i64p, i32p, i32p, C.c_int, C.c_int, C.c_double,
                                f64p, f64p, f64p, f64p, f64p, f64p, f64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_ilda_update_Elnbeta.argtypes = [C.c_int, C.c_int, i32p, f64p, f64p]
    L.orc_ilda_update_beta.argtypes = [C.c_int, C.c_int, i32p, f64p, f64p]
    L.orc_ilda_update_phi.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i64p, i32p, f64p, f64p, C.c_int, f64p]
    L.orc_ilda_update_lambda.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p, f64p, i32p, i64p, i32p, i32p, f64p, f64p, f64p]
    L.orc_ilda_loglik.restype = C.c_double
    L.orc_ilda_loglik.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i64p, i32p, i32p, f64p, f64p]
    L.orc_ilda_elbo.restype = C.c_double
    L.orc_ilda_elbo.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.c_double, f64p, i32p, i64p, i32p, i32p, f64p, f64p, f64p, f64p, f64p, f64p]
    L.orc_ilda_fit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.c_double, f64p, i32p, i64p, i32p, i32p, C.c_int, C.c_int, C.c_double,
                               f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    P = C.POINTER(OrcCtm)
    for name in ("update_zeta", "update_theta", "update_nu", "update_lambda", "fitdoc"):
        getattr(L, "orc_ctm_" + name).argtypes = [P, C.c_int]
    for name in ("update_mu", "update_Elnphi", "update_gamma", "update_props", "update_phi", "init"):
        getattr(L, "orc_ctm_" + name).argtypes = [P]
    L.orc_ctm_update_Sigma.argtypes = [P]; L.orc_ctm_update_Sigma.restype = C.c_int
    L.orc_ctm_calc_sumtheta.argtypes = [P, C.c_int, f64p]
    L.orc_ctm_calc_Ndivzeta.argtypes = [P, C.c_int, f64p]
    L.orc_ctm_loglik.argtypes = [P, f64p]
    L.orc_ctm_elbo.argtypes = [P, f64p]; L.orc_ctm_elbo.restype = C.c_double
    L.orc_ctm_estep_range.argtypes = [P, C.c_int, C.c_int]
    L.orc_alpha_objective.restype = C.c_double
    L.orc_alpha_objective.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_int]
    L.orc_ctm_update_alpha.argtypes = [P]
    L.orc_ctm_fit.argtypes = [P, C.c_int, C.c_double, C.c_int, C.c_int, f64p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                              C.POINTER(C.c_double)]
    L.orc_ctm_unsmoothed_update_theta.argtypes = [P, C.c_int]
    L.orc_ctm_infer.argtypes = [P, C.c_int, C.c_int, C.c_double, f64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_twin_topics.argtypes = [P, C.c_void_p]
    L.orc_twin_estep.argtypes = [P, f64p]
    L.orc_twin_estep_fused.argtypes = [P, f64p]
    L.orc_twin_moments.argtypes = [P, f64p]
    L.orc_twin_gauss.argtypes = [P, f64p, C.c_int]; L.orc_twin_gauss.restype = C.c_int
    L.orc_twin_pass.argtypes = [P, C.c_int]; L.orc_twin_pass.restype = C.c_int
    L.orc_twin_tables_from_Elnphi.argtypes = [P]
    L.orc_twin_infer_pass.argtypes = [P, C.c_int]; L.orc_twin_infer_pass.restype = C.c_int
    L.orc_twin_objectives.argtypes = [C.c_int, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p, f64p]
    L.orc_twin_arg_ranges.argtypes = [f64p, C.c_int]
    for name in ("exp", "log", "digamma", "exptab", "logtab", "digammatab"):
        getattr(L, "orc_ar_%s_vec" % name).argtypes = [C.c_int, f64p, f64p]
    _LIB = L
    return L


_LIB_OMP = None


def lib_omp():
    """The OpenMP variant of the LDA pass (oracle/mmm_oracle_omp.c): bench.py's all-cores CPU baseline only."""
    global _LIB_OMP
    if _LIB_OMP is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "build", "libmmm_oracle_omp.so"))
        L.orc_omp_threads.restype = C.c_int
        L.orc_omp_set_threads.argtypes = [C.c_int]
        L.orc_lda_pass_omp.restype = C.c_double
        L.orc_lda_pass_omp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i64p, i32p, i32p, f64p, f64p, f64p, f64p, f64p, f64p, f64p]
        _LIB_OMP = L
    return _LIB_OMP


# ------------------------------------------------------------------------------------------------------
# scalar helpers
# ------------------------------------------------------------------------------------------------------
def digamma(x):
    x = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)))
    out = np.empty_like(x)
    lib().orc_digamma_vec(x.size, x, out)
    return out


def inv_logdet(A):
    A = np.asfortranarray(np.asarray(A, dtype=np.float64))
    n = A.shape[0]
    Ai = np.empty(n * n)
    ld = C.c_double(); sg = C.c_int()
    rc = lib().orc_inv_logdet(n, np.ascontiguousarray(A.ravel(order="F")), Ai, C.byref(ld), C.byref(sg))
    return rc, Ai.reshape(n, n, order="F"), ld.value, sg.value


def mma_minimize(fun, x0, lb=None, ub=None, xtol_rel=1e-4, xtol_abs=1e-4, rule=0, max_eval=100000, trace=False):
    """fun(x) -> (value, grad); minimised. Returns (x, fmin, n_eval, n_outer); with trace=True a fifth item, the array of the solver's
    inner iterations, one row [rho, gval, wval, fcur, sigma[n], xcur[n]] each."""
    x = np.array(x0, dtype=np.float64)
    n = x.size
    tbuf = None
    if trace:
        tbuf = np.zeros((4096, 4 + 2 * n))
        lib().orc_mma_set_trace(tbuf.ctypes.data_as(C.c_void_p), tbuf.shape[0])

    def cb(nn, xp, gp, _):
        xv = np.ctypeslib.as_array(xp, shape=(nn,))
        v, g = fun(xv.copy())
        if gp:
            np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
        return float(v)

    lbp = np.ascontiguousarray(lb, dtype=np.float64) if lb is not None else None
    ubp = np.ascontiguousarray(ub, dtype=np.float64) if ub is not None else None
    minf = C.c_double(); no = C.c_int()
    nev = lib().orc_mma_minimize(n, OBJ_CB(cb), None,
                                 lbp.ctypes.data if lbp is not None else None,
                                 ubp.ctypes.data if ubp is not None else None,
                                 x, C.byref(minf), xtol_rel, xtol_abs, rule, max_eval, C.byref(no))
    if trace:
        rows = lib().orc_mma_trace_rows()
        lib().orc_mma_set_trace(None, 0)
        return x, minf.value, nev, no.value, tbuf[:rows].copy()
    return x, minf.value, nev, no.value


def alpha_objective(alpha, sum_Elnphi, K, V):
    g = C.c_double()
    v = lib().orc_alpha_objective(float(alpha), C.byref(g), float(sum_Elnphi), int(K), int(V))
    return v, g.value


def lambda_objective(lam, nu, Ndivzeta, sumtheta, mu, invSigma):
    n = len(lam); g = np.empty(n)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (lam, nu, Ndivzeta, sumtheta, mu)]
    iS = np.ascontiguousarray(np.asarray(invSigma, dtype=np.float64).ravel(order="F"))
    v = lib().orc_lambda_objective(n, a[0], g.ctypes.data, a[1], a[2], a[3], a[4], iS)
    return v, g


def nu_objective(nu, lam, Ndivzeta, mu, invSigma):
    n = len(nu); g = np.empty(n)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (nu, lam, Ndivzeta, mu)]
    iS = np.ascontiguousarray(np.asarray(invSigma, dtype=np.float64).ravel(order="F"))
    v = lib().orc_nu_objective(n, a[0], g.ctypes.data, a[1], a[2], a[3], iS)
    return v, g


# ------------------------------------------------------------------------------------------------------
# corpus flattening (reference layout X[d] = W_d x 2 Int matrices, 1-based terms: utils.jl:1-7)
# ------------------------------------------------------------------------------------------------------
def flatten_lda(X):
    """X: list of (W_d x 2) integer arrays with 1-based term ids -> (doc_ptr, term0, count)."""
    D = len(X)
    doc_ptr = np.zeros(D + 1, dtype=np.int64)
    for d in range(D):
        doc_ptr[d + 1] = doc_ptr[d] + np.asarray(X[d]).reshape(-1, 2).shape[0]
    term = np.empty(doc_ptr[-1], dtype=np.int32); count = np.empty(doc_ptr[-1], dtype=np.int32)
    for d in range(D):
        x = np.asarray(X[d]).reshape(-1, 2)
        term[doc_ptr[d]:doc_ptr[d + 1]] = x[:, 0] - 1
        count[doc_ptr[d]:doc_ptr[d + 1]] = x[:, 1]
    return doc_ptr, term, count


def flatten_mm(X, M):
    """X[d][m] -> modality-major concatenation; doc_ptr is M*(D+1) absolute."""
    D = len(X)
    doc_ptr = np.zeros(M * (D + 1), dtype=np.int64)
    terms, counts = [], []
    base = 0
    for m in range(M):
        dp, t, c = flatten_lda([X[d][m] for d in range(D)])
        doc_ptr[m * (D + 1):(m + 1) * (D + 1)] = dp + base
        base += dp[-1]
        terms.append(t); counts.append(c)
    return doc_ptr, np.concatenate(terms).astype(np.int32), np.concatenate(counts).astype(np.int32)


# ------------------------------------------------------------------------------------------------------
# LDA
# ------------------------------------------------------------------------------------------------------
class LdaOracle:
    """State container mirroring `mutable struct LDA` (LDA.jl:1-22) in the flat layouts of the header."""

    def __init__(self, K, alpha, eta, X, V=None, lambda0=None, seed=0):
        self.K, self.alpha, self.eta = int(K), float(alpha), float(eta)
        self.doc_ptr, self.term, self.count = flatten_lda(X)
        self.D = len(X)
        self.V = int(V) if V is not None else (int(self.term.max()) + 1 if self.term.size else 0)
        self.nnz = int(self.doc_ptr[-1])
        if lambda0 is None:
            lambda0 = np.random.default_rng(seed).integers(1, 101, size=(self.V, self.K)).astype(np.float64)
        self.lam = np.ascontiguousarray(np.asarray(lambda0, dtype=np.float64).ravel(order="F"))  # V x K col-major
        self.Elnbeta = np.empty(self.V * self.K); self.beta = np.empty(self.V * self.K)
        lib().orc_lda_update_Elnbeta(self.V, self.K, self.lam, self.Elnbeta)
        self.gamma = np.ones(self.K * self.D); self.theta = np.empty(self.K * self.D)
        self.Elntheta = np.empty(self.K * self.D)
        lib().orc_lda_update_Elntheta(self.K, self.D, self.gamma, self.Elntheta)
        self.phi = np.full(self.K * self.nnz, 1.0 / self.K)
        self.N = np.array([self.count[self.doc_ptr[d]:self.doc_ptr[d + 1]].sum() for d in range(self.D)])

    def update_gamma(self):
        lib().orc_lda_update_gamma(self.K, self.D, self.alpha, self.doc_ptr, self.count, self.phi, self.gamma, self.Elntheta)

    def update_phi(self):
        lib().orc_lda_update_phi(self.K, self.D, self.V, self.doc_ptr, self.term, self.Elntheta, self.Elnbeta, self.phi)

    def update_lambda(self):
        lib().orc_lda_update_lambda(self.K, self.D, self.V, self.eta, self.doc_ptr, self.term, self.count, self.phi, self.lam, self.Elnbeta)

    def update_beta(self):
        lib().orc_lda_update_beta(self.V, self.K, self.lam, self.beta)

    def update_theta(self):
        lib().orc_lda_update_theta(self.K, self.D, self.gamma, self.theta)

    def loglik(self):
        return lib().orc_lda_loglik(self.K, self.D, self.V, self.doc_ptr, self.term, self.count, self.theta, self.beta)

    def pass_omp(self):
        """One body of fit! with the document loops on all host cores (timing baseline); returns the ll."""
        return lib_omp().orc_lda_pass_omp(self.D, self.V, self.K, self.alpha, self.eta, self.doc_ptr, self.term, self.count, self.lam,
                                          self.Elnbeta, self.beta, self.gamma, self.Elntheta, self.theta, self.phi)

    def elbo(self):
        t = np.empty(7)
        e = lib().orc_lda_elbo(self.K, self.D, self.V, self.alpha, self.eta, self.doc_ptr, self.term, self.count,
                               self.lam, self.Elnbeta, self.gamma, self.Elntheta, self.phi, t)
        return e, t

    def fit(self, maxiter=1000, tol=1e-4):
        ll = np.zeros(maxiter); ni = C.c_int(); cv = C.c_int(); el = C.c_double()
        lib().orc_lda_fit(self.D, self.V, self.K, self.alpha, self.eta, self.doc_ptr, self.term, self.count,
                          maxiter, tol, self.lam, self.gamma, self.Elntheta, self.theta, self.Elnbeta, self.beta,
                          self.phi, ll, C.byref(ni), C.byref(cv), C.byref(el))
        self.converged = bool(cv.value); self.elbo_value = el.value
        self.ll_hist = ll[:ni.value].copy()
        return self.ll_hist

    # ---- frozen-topic inference (LDA.jl:226-295) ---------------------------------------------------------------------
    def unsmoothed_update_phi(self):
        lib().orc_lda_unsmoothed_update_phi(self.K, self.D, self.V, self.doc_ptr, self.term, self.Elntheta, self.beta, self.phi)

    def _infer(self, unsmoothed, maxiter, tol):
        ll = np.zeros(maxiter); ni = C.c_int(); cv = C.c_int()
        lib().orc_lda_infer(self.D, self.V, self.K, self.alpha, self.doc_ptr, self.term, self.count, int(unsmoothed), maxiter, tol,
                            self.Elnbeta, self.beta, self.gamma, self.Elntheta, self.theta, self.phi, ll, C.byref(ni), C.byref(cv))
        self.converged = bool(cv.value); self.ll_hist = ll[:ni.value].copy()
        return self.ll_hist

    def transform(self, X, maxiter=1000, tol=1e-4):
        """transform(model::LDA, X) LDA.jl:233-263 -> (theta [K x D_new] column-major flat, new oracle model)."""
        new = LdaOracle(self.K, self.alpha, self.eta, X, V=self.V, seed=1)
        new.beta[:] = self.beta                                        # :237
        new._infer(True, maxiter, tol)
        return new.theta, new

    def fit_heldout(self, X, maxiter=100):
        """fit_heldout(Xheldout, model::LDA) LDA.jl:265-295."""
        new = LdaOracle(self.K, self.alpha, self.eta, X, V=self.V, seed=1)
        new.lam[:] = self.lam; new.beta[:] = self.beta; new.Elnbeta[:] = self.Elnbeta          # :269-271
        new._infer(False, maxiter, 1e-4)
        new.elbo_value = new.elbo()[0]; new.ll = new.ll_hist[-1]
        return new

    def phi_doc(self, d):
        W = self.doc_ptr[d + 1] - self.doc_ptr[d]
        return self.phi[self.K * self.doc_ptr[d]:self.K * self.doc_ptr[d + 1]].reshape(W, self.K).T  # K x W


# ------------------------------------------------------------------------------------------------------
# ILDA
# ------------------------------------------------------------------------------------------------------
class IldaOracle:
    """State container mirroring `mutable struct ILDA` (ILDA.jl:1-23).  lam / Elnbeta / beta: the I factor matrices (J_i x K,
    column-major) concatenated; `mat(arr, i)` views one.  features: V x I matrix of 1-based values."""

    def __init__(self, K, alpha, eta, features, X, lambda0=None, seed=0):
        self.K, self.alpha = int(K), float(alpha)
        f = np.asarray(features, dtype=np.int64)
        self.V, self.I = int(f.shape[0]), int(f.shape[1])
        self.J = np.ascontiguousarray(f.max(axis=0), dtype=np.int32)
        self.eta = np.full(self.I, float(eta)) if np.ndim(eta) == 0 else np.asarray(eta, dtype=np.float64).copy()
        self.features = np.ascontiguousarray((f - 1).T.ravel(), dtype=np.int32)          # [i*V + v]
        self.doc_ptr, self.term, self.count = flatten_lda(X)
        self.D = len(X); self.nnz = int(self.doc_ptr[-1])
        self.off = np.concatenate([[0], np.cumsum(self.J.astype(np.int64) * self.K)])
        n = int(self.off[-1])
        if lambda0 is None:
            lambda0 = np.random.default_rng(seed).integers(1, 101, size=n).astype(np.float64)
        self.lam = np.ascontiguousarray(lambda0, dtype=np.float64).copy()
        assert self.lam.size == n
        self.Elnbeta = np.empty(n); self.beta = np.zeros(n)
        lib().orc_ilda_update_Elnbeta(self.K, self.I, self.J, self.lam, self.Elnbeta)
        self.gamma = np.ones(self.K * self.D); self.theta = np.zeros(self.K * self.D); self.Elntheta = np.empty(self.K * self.D)
        lib().orc_lda_update_Elntheta(self.K, self.D, self.gamma, self.Elntheta)
        self.phi = np.full(self.K * self.nnz, 1.0 / self.K)

    def mat(self, arr, i):
        return arr[self.off[i]:self.off[i + 1]].reshape(int(self.J[i]), self.K, order="F")

    def update_gamma(self):
        lib().orc_lda_update_gamma(self.K, self.D, self.alpha, self.doc_ptr, self.count, self.phi, self.gamma, self.Elntheta)

    def update_phi(self):
        lib().orc_ilda_update_phi(self.K, self.D, self.V, self.I, self.J, self.features, self.doc_ptr, self.term, self.Elntheta, self.Elnbeta, 0, self.phi)

    def update_lambda(self):
        lib().orc_ilda_update_lambda(self.K, self.D, self.V, self.I, self.J, self.eta, self.features, self.doc_ptr, self.term, self.count,
                                     self.phi, self.lam, self.Elnbeta)

    def update_beta(self):
        lib().orc_ilda_update_beta(self.K, self.I, self.J, self.lam, self.beta)

    def update_theta(self):
        lib().orc_lda_update_theta(self.K, self.D, self.gamma, self.theta)

    def loglik(self):
        return lib().orc_ilda_loglik(self.K, self.D, self.V, self.I, self.J, self.features, self.doc_ptr, self.term, self.count, self.theta, self.beta)

    def elbo(self):
        t = np.empty(7)
        e = lib().orc_ilda_elbo(self.K, self.D, self.V, self.I, self.J, self.alpha, self.eta, self.features, self.doc_ptr, self.term, self.count,
                                self.lam, self.Elnbeta, self.gamma, self.Elntheta, self.phi, t)
        return e, t

    def _run(self, frozen, maxiter, tol):
        ll = np.zeros(maxiter); ni = C.c_int(); cv = C.c_int(); el = C.c_double()
        lib().orc_ilda_fit(self.D, self.V, self.K, self.I, self.J, self.alpha, self.eta, self.features, self.doc_ptr, self.term, self.count,
                           int(frozen), maxiter, tol, self.lam, self.Elnbeta, self.beta, self.gamma, self.Elntheta, self.theta, self.phi, ll,
                           C.byref(ni), C.byref(cv), C.byref(el))
        self.converged = bool(cv.value); self.elbo_value = el.value
        self.ll_hist = ll[:ni.value].copy()
        return self.ll_hist

    def fit(self, maxiter=1000, tol=1e-4):
        return self._run(False, maxiter, tol)

    def fit_heldout(self, X, maxiter=100):
        """fit_heldout(Xheldout, model::ILDA) ILDA.jl:320-353"""
        f1 = self.features.reshape(self.I, self.V).T + 1
        new = IldaOracle(self.K, self.alpha, self.eta, f1, X, seed=1)
        new.lam[:] = self.lam; new.beta[:] = self.beta; new.Elnbeta[:] = self.Elnbeta
        new._run(True, maxiter, 1e-4)
        new.ll = new.ll_hist[-1]
        return new

    def phi_doc(self, d):
        W = self.doc_ptr[d + 1] - self.doc_ptr[d]
        return self.phi[self.K * self.doc_ptr[d]:self.K * self.doc_ptr[d + 1]].reshape(W, self.K).T


# ------------------------------------------------------------------------------------------------------
# MMCTM / IMMCTM
# ------------------------------------------------------------------------------------------------------
class CtmOracle:
    """State container mirroring `mutable struct MMCTM` / `IMMCTM` in flat layouts.

    gamma0: flat init array in the header's gamma layout (integers 1..100 in the reference ctor).
    features: None (MMCTM) or list of (V_m x I_m) 1-based integer matrices (IMMCTM).
    """

    def __init__(self, K, alpha, X, V=None, features=None, gamma0=None, seed=0, xtol_rule=0, max_eval=100000, geometry=None):
        """geometry: None -> every sum in index order (mmm_oracle.c).  A dict {L, grid_e, waves_e, grid_m} (the launch
        geometry of a device handle, `MMCTM.geometry()`) -> the order-matched variant (mmm_twin.c): same algorithm, sums
        associated as the gfx950 kernels associate them, exp/log/digamma from csrc/mmm_arith.h."""
        self.K = np.asarray(K, dtype=np.int32); self.M = len(K); self.MK = int(self.K.sum())
        self.D = len(X)
        self.doc_ptr, self.term, self.count = flatten_mm(X, self.M)
        D = self.D
        self.immctm = features is not None
        if self.immctm:
            self.nfeat = np.array([np.asarray(f).shape[1] for f in features], dtype=np.int32)
            self.J = np.concatenate([np.asarray(f).max(axis=0) for f in features]).astype(np.int32)
            self.V = np.array([np.asarray(f).shape[0] for f in features], dtype=np.int32)
            # [foff[m] + i*V + v], 0-based values
            self.features = np.concatenate([(np.asarray(f, dtype=np.int32) - 1).T.ravel() for f in features]).astype(np.int32)
            a = np.asarray(alpha, dtype=object)
            if np.ndim(alpha[0]) == 0:
                self.alpha = np.concatenate([np.full(self.nfeat[m], float(alpha[m])) for m in range(self.M)])
            else:
                self.alpha = np.concatenate([np.asarray(alpha[m], dtype=np.float64) for m in range(self.M)])
            SJ = []
            o = 0
            for m in range(self.M):
                SJ.append(int(self.J[o:o + self.nfeat[m]].sum())); o += self.nfeat[m]
            self.SJ = np.array(SJ)
            self.gsize = [int(self.K[m] * self.SJ[m]) for m in range(self.M)]
        else:
            if V is None:
                V = []
                for m in range(self.M):
                    s, e = self.doc_ptr[m * (D + 1)], self.doc_ptr[m * (D + 1) + D]
                    V.append(int(self.term[s:e].max()) + 1 if e > s else 0)
            self.V = np.asarray(V, dtype=np.int32)
            self.alpha = np.asarray(alpha, dtype=np.float64).copy()
            self.gsize = [int(self.K[m] * self.V[m]) for m in range(self.M)]
            self.nfeat = None; self.J = None; self.features = None
        self.goff = np.concatenate([[0], np.cumsum(self.gsize)]).astype(np.int64)
        G = int(self.goff[-1])
        if gamma0 is None:
            gamma0 = np.random.default_rng(seed).integers(1, 101, size=G).astype(np.float64)
        self.gamma = np.ascontiguousarray(gamma0, dtype=np.float64).copy()
        assert self.gamma.size == G
        self.Elnphi = np.empty(G); self.phi = np.empty(G)
        self.mu = np.zeros(self.MK); self.Sigma = np.zeros(self.MK * self.MK); self.invSigma = np.zeros(self.MK * self.MK)
        self.lam = np.zeros(self.MK * D); self.nu = np.ones(self.MK * D)
        self.zeta = np.zeros(self.M * D); self.props = np.zeros(self.MK * D)
        self.nnz = [int(self.doc_ptr[m * (D + 1) + D] - self.doc_ptr[m * (D + 1)]) for m in range(self.M)]
        self.toff = np.concatenate([[0], np.cumsum([self.nnz[m] * int(self.K[m]) for m in range(self.M)])]).astype(np.int64)
        self.theta = np.zeros(int(self.toff[-1]))
        self.N = np.array([[self.count[self.doc_ptr[m * (D + 1) + d]:self.doc_ptr[m * (D + 1) + d + 1]].sum()
                            for m in range(self.M)] for d in range(D)])
        s = OrcCtm()
        s.D, s.M, s.MK = D, self.M, self.MK
        self._Kc = np.ascontiguousarray(self.K, dtype=np.int32); self._Vc = np.ascontiguousarray(self.V, dtype=np.int32)
        s.K = self._Kc.ctypes.data; s.V = self._Vc.ctypes.data
        s.doc_ptr = self.doc_ptr.ctypes.data; s.term = self.term.ctypes.data; s.count = self.count.ctypes.data
        if self.immctm:
            s.n_feat = self.nfeat.ctypes.data; s.J = self.J.ctypes.data; s.features = self.features.ctypes.data
        s.alpha = self.alpha.ctypes.data
        s.mu = self.mu.ctypes.data; s.Sigma = self.Sigma.ctypes.data; s.invSigma = self.invSigma.ctypes.data
        s.gamma = self.gamma.ctypes.data; s.Elnphi = self.Elnphi.ctypes.data
        s.phi = None if self.immctm else self.phi.ctypes.data
        s.lambda_ = self.lam.ctypes.data; s.nu = self.nu.ctypes.data; s.zeta = self.zeta.ctypes.data
        s.props = self.props.ctypes.data; s.theta = self.theta.ctypes.data
        s.xtol_rel = 1e-4; s.xtol_abs = 1e-4; s.nu_lower = 1e-7; s.xtol_rule = xtol_rule; s.max_eval = max_eval
        self.nev_nu = np.zeros(max(D, 1), dtype=np.int32); self.nev_lambda = np.zeros(max(D, 1), dtype=np.int32)
        s.nev_nu = self.nev_nu.ctypes.data; s.nev_lambda = self.nev_lambda.ctypes.data
        self.expE = np.zeros(int(sum(int(self.K[m]) * int(self.V[m]) for m in range(self.M))))
        s.expE = self.expE.ctypes.data
        if geometry is not None:
            s.arith = 1
            s.L, s.grid_e, s.waves_e, s.grid_m = (int(geometry[k]) for k in ("L", "grid_e", "waves_e", "grid_m"))
            s.Ls = int(geometry.get("Ls", 0) or s.L)
            s.cpl = int(geometry.get("cpl", 1) or 1)
            s.tdense = int(geometry.get("tdense", 0) or 0)
            assert s.L in (16, 32, 64) and s.L >= self.MK and s.grid_e >= 1 and s.waves_e >= 1 and s.grid_m >= 1
            assert (s.cpl == 1 and self.MK <= s.Ls <= 64) or (s.cpl > 1 and self.MK % s.cpl == 0 and s.Ls * s.cpl >= self.MK and s.Ls in (2, 4, 8, 16))
        self.s = s
        lib().orc_ctm_init(C.byref(s))

    # thin pass-throughs (d is 0-based)
    def update_zeta(self, d): lib().orc_ctm_update_zeta(C.byref(self.s), d)
    def update_theta(self, d): lib().orc_ctm_update_theta(C.byref(self.s), d)
    def update_nu(self, d): lib().orc_ctm_update_nu(C.byref(self.s), d)
    def update_lambda(self, d): lib().orc_ctm_update_lambda(C.byref(self.s), d)
    def fitdoc(self, d): lib().orc_ctm_fitdoc(C.byref(self.s), d)
    def estep_range(self, d0, d1): lib().orc_ctm_estep_range(C.byref(self.s), d0, d1)
    def update_mu(self): lib().orc_ctm_update_mu(C.byref(self.s))
    def update_Sigma(self): return lib().orc_ctm_update_Sigma(C.byref(self.s))
    def update_Elnphi(self): lib().orc_ctm_update_Elnphi(C.byref(self.s))
    def update_gamma(self): lib().orc_ctm_update_gamma(C.byref(self.s))
    def update_props(self): lib().orc_ctm_update_props(C.byref(self.s))
    def update_phi(self): lib().orc_ctm_update_phi(C.byref(self.s))

    def sumtheta(self, d):
        o = np.empty(self.MK); lib().orc_ctm_calc_sumtheta(C.byref(self.s), d, o); return o

    def Ndivzeta(self, d):
        o = np.empty(self.MK); lib().orc_ctm_calc_Ndivzeta(C.byref(self.s), d, o); return o

    def loglik(self):
        o = np.empty(self.M); lib().orc_ctm_loglik(C.byref(self.s), o); return o

    def elbo(self):
        t = np.empty(7); e = lib().orc_ctm_elbo(C.byref(self.s), t); return e, t

    def update_alpha(self): lib().orc_ctm_update_alpha(C.byref(self.s))

    # ---- the pieces of one pass in device order (geometry given) ----
    def twin_pass(self, update_sigma=True):
        assert self.s.arith
        return lib().orc_twin_pass(C.byref(self.s), int(update_sigma or self.immctm))

    def twin_estep(self):
        assert self.s.arith
        sG = np.zeros(self.expE.size); lib().orc_twin_estep(C.byref(self.s), sG); return sG

    def twin_estep_fused(self):
        """the E-step as the device's FUSED pass runs it (theta phase over rows of counts when the geometry says tdense)"""
        sG = np.zeros(self.expE.size); lib().orc_twin_estep_fused(C.byref(self.s), sG); return sG

    def fit(self, maxiter=100, tol=1e-4, update_sigma=True, auto_alpha=False):
        ll = np.zeros(self.M * maxiter); ni = C.c_int(); cv = C.c_int(); el = C.c_double()
        lib().orc_ctm_fit(C.byref(self.s), maxiter, tol, int(update_sigma), int(auto_alpha), ll, C.byref(ni), C.byref(cv), C.byref(el))
        self.converged = bool(cv.value); self.elbo_value = el.value
        self.ll_hist = ll[:self.M * ni.value].reshape(ni.value, self.M).copy()
        return self.ll_hist

    # ---- frozen-topic inference (MMCTM.jl:496-634, IMMCTM.jl:468-545) ---------------------------------------------------
    def unsmoothed_update_theta(self, d): lib().orc_ctm_unsmoothed_update_theta(C.byref(self.s), d)

    def infer(self, flags, maxiter, tol):
        ll = np.zeros(self.M * maxiter); ni = C.c_int(); cv = C.c_int()
        lib().orc_ctm_infer(C.byref(self.s), int(flags), maxiter, tol, ll, C.byref(ni), C.byref(cv))
        self.converged = bool(cv.value)
        self.ll_hist = ll[:self.M * ni.value].reshape(ni.value, self.M).copy()
        return self.ll_hist

    # views
    def theta_dm(self, d, m):
        D = self.D
        e0 = self.doc_ptr[m * (D + 1) + d] - self.doc_ptr[m * (D + 1)]
        e1 = self.doc_ptr[m * (D + 1) + d + 1] - self.doc_ptr[m * (D + 1)]
        Km = int(self.K[m])
        return self.theta[self.toff[m] + e0 * Km:self.toff[m] + e1 * Km].reshape(e1 - e0, Km).T  # K_m x W

    def gamma_mk(self, m, k, arr=None):
        arr = self.gamma if arr is None else arr
        w = self.gsize[m] // int(self.K[m])
        return arr[self.goff[m] + k * w:self.goff[m] + (k + 1) * w]
