"""Cross-validate the C oracle with an independent numpy/scipy restatement (tests/np_ref.py) on seeded
synthetic corpora: closed-form updates, log-likelihood and every ELBO term to ~1e-11; MMA outputs by
first-order optimality and against a scipy optimiser (the reference pins none of these -- SURVEY §8c)."""
import numpy as np
import pytest
from scipy.optimize import minimize

import np_ref


def test_lda_two_iterations_and_elbo(oracle):
    X, lam0 = np_ref.synth_lda(40, 24, 5, seed=11, mean_n=400)
    K, alpha, eta = 5, 0.1, 0.1
    m = oracle.LdaOracle(K, alpha, eta, X, V=24, lambda0=lam0)
    lam = lam0.copy(); phi = [np.full((K, x.shape[0]), 1.0 / K) for x in X]
    for it in range(2):
        r = np_ref.lda_iteration(X, K, alpha, eta, lam, phi)
        m.update_gamma(); m.update_phi(); m.update_lambda(); m.update_beta(); m.update_theta()
        np.testing.assert_allclose(m.gamma.reshape(-1, K).T, r["gamma"], rtol=1e-12)
        np.testing.assert_allclose(m.Elntheta.reshape(-1, K).T, r["Elntheta"], rtol=1e-11, atol=1e-13)
        for d in range(len(X)):
            np.testing.assert_allclose(m.phi_doc(d), r["phi"][d], rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(m.lam.reshape(24, K, order="F"), r["lam"], rtol=1e-12)
        np.testing.assert_allclose(m.Elnbeta.reshape(24, K, order="F"), r["Elnbeta"], rtol=1e-11, atol=1e-13)
        assert m.loglik() == pytest.approx(r["ll"], rel=1e-12)
        lam, phi = r["lam"], r["phi"]
    e, t = m.elbo()
    e2, t2 = np_ref.lda_elbo(X, K, alpha, eta, r["lam"], r["Elnbeta"], r["gamma"], r["Elntheta"], r["phi"])
    np.testing.assert_allclose(t, t2, rtol=1e-11)
    assert e == pytest.approx(e2, rel=1e-11)


def test_lda_fit_matches_stepwise(oracle):
    X, lam0 = np_ref.synth_lda(30, 16, 3, seed=5, mean_n=300)
    a = oracle.LdaOracle(3, 0.1, 0.1, X, V=16, lambda0=lam0)
    ll = a.fit(maxiter=15, tol=0.0)
    assert len(ll) == 15 and not a.converged
    b = oracle.LdaOracle(3, 0.1, 0.1, X, V=16, lambda0=lam0)
    for _ in range(15):
        b.update_gamma(); b.update_phi(); b.update_lambda(); b.update_beta(); b.update_theta()
    np.testing.assert_array_equal(a.lam, b.lam); np.testing.assert_array_equal(a.phi, b.phi)
    assert a.elbo_value == b.elbo()[0]


@pytest.mark.parametrize("rule", [0, 1])
def test_mma_on_quadratic_and_bounds(oracle, rule):
    A = np.array([[3.0, 0.5], [0.5, 1.0]]); b = np.array([1.0, -2.0])
    f = lambda x: (0.5 * x @ A @ x - b @ x, A @ x - b)
    x, fmin, nev, nout = oracle.mma_minimize(f, [5.0, 5.0], rule=rule)
    np.testing.assert_allclose(x, np.linalg.solve(A, b), atol=2e-3)
    assert nev > 2 and nout >= 2
    # lower bound active: minimise (x-(-1))^2 subject to x >= 0.5
    g = lambda x: (((x + 1.0) ** 2).sum(), 2 * (x + 1.0))
    x, fmin, nev, nout = oracle.mma_minimize(g, [3.0], lb=[0.5], rule=rule)
    assert x[0] == pytest.approx(0.5, abs=1e-12)


def _mm_setup(oracle, D=25, seed=3, empty_frac=0.2):
    K = [3, 4]; V = [20, 12]; alpha = [0.1, 0.1]
    X, g0 = np_ref.synth_mm(D, V, K, seed=seed, means=[300, 60], empty_frac=empty_frac)
    m = oracle.CtmOracle(K, alpha, X, V=V, gamma0=np.concatenate([g.ravel() for g in g0]))
    return K, V, alpha, X, g0, m


def test_mmctm_estep_closed_forms_and_mma_optimality(oracle):
    K, V, alpha, X, g0, m = _mm_setup(oracle)
    from scipy.special import digamma as psi
    Elnphi = [psi(g) - psi(g.sum(axis=1, keepdims=True)) for g in g0]
    MK = sum(K)
    # give the Gaussian a non-trivial shape first
    rng = np.random.default_rng(0)
    A = rng.standard_normal((MK, MK)); S = A @ A.T / MK + np.eye(MK)
    m.mu[:] = rng.standard_normal(MK) * 0.3; m.Sigma[:] = S.ravel(); m.invSigma[:] = np.linalg.inv(S).ravel(order="F")
    invS = m.invSigma.reshape(MK, MK, order="F")
    for d in range(len(X)):
        lam_old = m.lam[MK * d:MK * (d + 1)].copy(); nu_old = m.nu[MK * d:MK * (d + 1)].copy()
        zeta, theta = np_ref.mmctm_zeta_theta(X, K, lam_old, nu_old, Elnphi, d)
        m.update_zeta(d); m.update_theta(d)
        np.testing.assert_allclose(m.zeta[2 * d:2 * d + 2], zeta, rtol=1e-13)
        for mm in range(2):
            np.testing.assert_allclose(m.theta_dm(d, mm), theta[mm], rtol=1e-11)
        sumth, Ndz, f_lam, f_nu = np_ref.mmctm_objs(X, K, d, zeta, theta, m.mu, invS)
        np.testing.assert_allclose(m.sumtheta(d), sumth, rtol=1e-12)
        np.testing.assert_allclose(m.Ndivzeta(d), Ndz, rtol=1e-13)
        # nu: each MMA call must not decrease the objective; CCSA's global rho can end a solve started far
        # away (nu=1) early under xtol_abs=1e-4, so the optimum is reached over restarts, as in the EM loop
        # (every outer iteration re-creates the Opt: MMCTM.jl:157) -- compare after three restarts.
        f_prev = f_nu(nu_old, lam_old)[0]
        for _ in range(3):
            m.update_nu(d)
            nu_new = m.nu[MK * d:MK * (d + 1)].copy()
            assert np.all(nu_new >= 1e-7)
            f_new = f_nu(nu_new, lam_old)[0]
            assert f_new >= f_prev - 1e-9
            f_prev = f_new
        ref = minimize(lambda x: tuple(-v for v in f_nu(x, lam_old)), nu_old, jac=True, method="L-BFGS-B",
                       bounds=[(1e-7, None)] * MK, options=dict(gtol=1e-12, ftol=1e-15))
        np.testing.assert_allclose(nu_new, ref.x, atol=1e-3, rtol=1e-3)
        # lambda
        f_prev = f_lam(lam_old, nu_new)[0]
        for _ in range(3):
            m.update_lambda(d)
            lam_new = m.lam[MK * d:MK * (d + 1)].copy()
            f_new = f_lam(lam_new, nu_new)[0]
            assert f_new >= f_prev - 1e-9
            f_prev = f_new
        ref = minimize(lambda x: tuple(-v for v in f_lam(x, nu_new)), lam_old, jac=True, method="L-BFGS-B",
                       options=dict(gtol=1e-12, ftol=1e-15))
        np.testing.assert_allclose(lam_new, ref.x, atol=2e-3)
    assert m.s.n_solver_cap == 0 and m.s.n_eval_lambda > 0 and m.s.n_eval_nu > 0


def test_mmctm_mstep_loglik_elbo(oracle):
    K, V, alpha, X, g0, m = _mm_setup(oracle)
    D = len(X); MK = sum(K)
    for it in range(2):
        m.estep_range(0, D)
        lam = m.lam.reshape(D, MK).copy(); nu = m.nu.reshape(D, MK).copy()
        theta = [[m.theta_dm(d, mm).copy() for mm in range(2)] for d in range(D)]
        zeta = m.zeta.reshape(D, 2).copy()
        mu, S, invS, gamma, Elnphi, phi, props, ll = np_ref.mmctm_mstep(X, K, V, alpha, lam, nu, theta)
        m.update_mu(); assert m.update_Sigma() == 0; m.update_gamma(); m.update_props(); m.update_phi()
        np.testing.assert_allclose(m.mu, mu, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(m.Sigma.reshape(MK, MK), S, rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(m.invSigma.reshape(MK, MK, order="F"), invS, rtol=1e-9, atol=1e-11)
        for mm in range(2):
            for k in range(K[mm]):
                np.testing.assert_allclose(m.gamma_mk(mm, k), gamma[mm][k], rtol=1e-12)
                np.testing.assert_allclose(m.gamma_mk(mm, k, m.Elnphi), Elnphi[mm][k], rtol=1e-11, atol=1e-13)
                np.testing.assert_allclose(m.gamma_mk(mm, k, m.phi), phi[mm][k], rtol=1e-12)
        np.testing.assert_allclose(m.props.reshape(D, MK), props, rtol=1e-12)
        np.testing.assert_allclose(m.loglik(), ll, rtol=1e-12)
    e, t = m.elbo()
    e2, t2 = np_ref.mmctm_elbo(X, K, V, alpha, mu, m.invSigma.reshape(MK, MK, order="F"), gamma, Elnphi, lam, nu, zeta, theta)
    np.testing.assert_allclose(t, t2, rtol=1e-10)
    assert e == pytest.approx(e2, rel=1e-10)


def test_mmctm_fit_runs_and_xtol_rules_agree_loosely(oracle):
    K, V, alpha, X, g0, _ = _mm_setup(oracle, D=20, seed=9, empty_frac=0.1)
    g = np.concatenate([x.ravel() for x in g0])
    a = oracle.CtmOracle(K, alpha, X, V=V, gamma0=g, xtol_rule=0)
    b = oracle.CtmOracle(K, alpha, X, V=V, gamma0=g, xtol_rule=1)
    la = a.fit(maxiter=12, tol=1e-4); lb = b.fit(maxiter=12, tol=1e-4)
    assert la.shape[1] == 2 and len(la) >= 11 and np.all(np.isfinite(la))
    np.testing.assert_allclose(la[-1], lb[-1], rtol=2e-3)   # the two NLopt stopping rules differ only at the 1e-4 level
    assert np.isfinite(a.elbo_value)


# ---- frozen-topic inference restatements (LDA.jl:226-295, MMCTM.jl:496-552) against plain numpy ----------------------
def test_lda_unsmoothed_phi_and_transform_loop(oracle):
    X, lam0 = np_ref.synth_lda(30, 20, 4, seed=8, mean_n=300)
    K, V = 4, 20
    m = oracle.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0)
    m.fit(maxiter=5, tol=0.0)
    Xn, _ = np_ref.synth_lda(12, V, K, seed=80, mean_n=200)
    theta, new = m.transform(Xn, maxiter=15, tol=0.0)
    beta = m.beta.reshape(V, K, order="F")
    # numpy replay of LDA.jl:241-247
    from scipy.special import digamma
    gamma = np.ones((K, len(Xn))); phi = [np.full((K, x.shape[0]), 1.0 / K) for x in Xn]
    lls = []
    for _ in range(15):
        for d, x in enumerate(Xn):
            gamma[:, d] = 0.1 + phi[d] @ x[:, 1]                                  # update_γ! LDA.jl:82-90
        Eln = digamma(gamma) - digamma(gamma.sum(axis=0))
        for d, x in enumerate(Xn):
            p = np.exp(Eln[:, d])[:, None] * beta[x[:, 0] - 1, :].T               # unsmoothed_update_ϕ! LDA.jl:226-231
            phi[d] = p / p.sum(axis=0)
        th = gamma / gamma.sum(axis=0)
        num = sum(float(x[:, 1] @ np.log(beta[x[:, 0] - 1, :] @ th[:, d])) for d, x in enumerate(Xn))
        lls.append(num / sum(int(x[:, 1].sum()) for x in Xn))
    np.testing.assert_allclose(theta.reshape(len(Xn), K).T, th, rtol=1e-11)
    np.testing.assert_allclose(new.ll_hist, lls, rtol=1e-11)
    for d in range(len(Xn)):
        np.testing.assert_allclose(new.phi_doc(d), phi[d], rtol=1e-10, atol=1e-300)
    # fit_heldout differs only in the smoothed phi (LDA.jl:274-276)
    h = m.fit_heldout(Xn, maxiter=15)
    assert len(h.ll_hist) >= 11 and np.isfinite(h.elbo_value)
    assert not np.allclose(h.theta, theta)


def test_mmctm_unsmoothed_theta_and_transform_flags(oracle):
    _, _, _, X, _, m = _mm_setup(oracle)
    m.fit(maxiter=2, tol=0.0)
    K = [int(k) for k in m.K]
    new = oracle.CtmOracle(K, m.alpha, X, V=[int(v) for v in m.V], seed=9)
    new.phi[:] = m.phi
    rng = np.random.default_rng(0)
    new.lam[:] = rng.standard_normal(new.lam.size)
    for d in range(new.D):
        new.unsmoothed_update_theta(d)
    off = 0
    for mod in range(new.M):
        ph = m.phi[m.goff[mod]:m.goff[mod + 1]].reshape(K[mod], -1)
        for d in (0, 3, new.D - 1):
            x = X[d][mod]
            if x.shape[0] == 0:
                continue
            t = np.exp(new.lam[new.MK * d + off:new.MK * d + off + K[mod]])[:, None] * ph[:, x[:, 0] - 1]     # MMCTM.jl:496-509
            np.testing.assert_allclose(new.theta_dm(d, mod), t / t.sum(axis=0), rtol=1e-12)
        off += K[mod]
    # transform without fit_gaussian leaves Sigma alone, with it refits it (test/mmctm.jl:390-406)
    a = oracle.CtmOracle(K, m.alpha, X, V=[int(v) for v in m.V], seed=9); a.phi[:] = m.phi; a.mu[:] = m.mu; a.Sigma[:] = m.Sigma
    a.infer(1, 2, 1e4)
    np.testing.assert_array_equal(a.Sigma, m.Sigma)
    b = oracle.CtmOracle(K, m.alpha, X, V=[int(v) for v in m.V], seed=9); b.phi[:] = m.phi
    ll = b.infer(3, 12, 1e4)
    assert ll.shape == (11, new.M) and b.converged and not np.allclose(b.Sigma, m.Sigma)


def test_openmp_lda_pass_equals_the_sequential_oracle(oracle):
    """oracle/mmm_oracle_omp.c (bench.py's all-cores CPU baseline) against the literal sequential restatement."""
    X, lam0 = np_ref.synth_lda(300, 96, 10, seed=21, mean_n=500)
    a = oracle.LdaOracle(10, 0.1, 0.1, X, V=96, lambda0=lam0)
    b = oracle.LdaOracle(10, 0.1, 0.1, X, V=96, lambda0=lam0)
    for _ in range(5):
        a.update_gamma(); a.update_phi(); a.update_lambda(); a.update_beta(); a.update_theta()
        lla = a.loglik()
        llb = b.pass_omp()
        assert llb == pytest.approx(lla, rel=1e-12)
    np.testing.assert_allclose(b.lam, a.lam, rtol=1e-12)
    np.testing.assert_allclose(b.phi, a.phi, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(b.theta, a.theta, rtol=1e-12)
