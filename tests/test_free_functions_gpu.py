"""The reference's free functions and per-document calls through the C ABI (csrc/free.hip, mmm_ctm_update_doc, mmm_ctm_doc_sums),
against the reference's own known-answer tests: test/common.jl:79-97 (λ_objective), test/mmctm.jl:59-90 (calculate_Ndivζ / sumθ),
:135-148 (ν_objective), :268-279 (α_objective), :349-388 (log-likelihood helpers); test/immctm.jl:350-386; test/lda.jl (ll of a fit)."""
import numpy as np
import pytest

import np_ref

pytestmark = pytest.mark.gpu


def arr(x):
    return np.asarray(x, dtype=np.float64)


def _toy(mmm, kats, imm=False):
    c = kats["corpora"]
    X = [[arr(xm).astype(np.int64) for xm in xd] for xd in c["X_mm"]]
    if imm:
        return mmm.IMMCTM(c["K_mm"], c["alpha_mm"], c["features"], X, seed=5)
    return mmm.MMCTM(c["K_mm"], c["alpha_mm"], X, seed=5)


def test_lambda_objective_free(mmm, kats):                 # test/common.jl:79-97
    k = kats["lambda_objective"]
    g = np.zeros(5)
    v = mmm.λ_objective(k["lambda"], g, k["nu"], k["Ndivzeta"], k["sumtheta"], k["mu"], np.eye(5))
    assert v == pytest.approx(k["value"], rel=1e-13)
    np.testing.assert_allclose(g, k["grad"], rtol=1e-13)
    assert mmm.λ_objective(k["lambda"], None, k["nu"], k["Ndivzeta"], k["sumtheta"], k["mu"], np.eye(5)) == v     # `length(∇λ) == 0`


def test_nu_objective_free(mmm, kats):                     # test/mmctm.jl:135-148
    k = kats["nu_objective"]
    g = np.zeros(5)
    v = mmm.ν_objective(k["nu"], g, k["lambda"], kats["lambda_objective"]["Ndivzeta"], k["mu"], np.eye(5))
    assert v == pytest.approx(k["value"], rel=1e-13)
    np.testing.assert_allclose(g, k["grad"], rtol=1e-13)


def test_objectives_free_match_the_model_bound_ones(mmm):
    """a non-trivial invΣ (the KATs use I): the free functions agree with mmm_ctm_objectives on a fitted model's document"""
    X, g0 = np_ref.synth_mm(40, [30, 20], [4, 3], seed=11, means=[300, 60], empty_frac=0.0)
    m = mmm.MMCTM([4, 3], [0.1, 0.1], [30, 20], X, γ0=g0)
    mmm.fit(m, maxiter=3, tol=0.0, verbose=False)
    d = 7
    lv, lg, nv, ng = m.objectives(d)
    s, c = mmm.calculate_sumθ(m, d), mmm.calculate_Ndivζ(m, d)
    g = np.zeros(7)
    assert mmm.λ_objective(m.λ[d], g, m.ν[d], c, s, m.μ, m.invΣ) == pytest.approx(lv, rel=1e-12)
    np.testing.assert_allclose(g, lg, rtol=1e-11, atol=1e-11)
    assert mmm.ν_objective(m.ν[d], g, m.λ[d], c, m.μ, m.invΣ) == pytest.approx(nv, rel=1e-12)
    np.testing.assert_allclose(g, ng, rtol=1e-11, atol=1e-11)


def test_alpha_objective_free(mmm, kats):                  # test/mmctm.jl:268-279; test/immctm.jl:273-284
    for c in kats["alpha_objective"]["cases"]:
        g = np.zeros(1)
        v = mmm.α_objective([c["alpha"]], g, c["sum_Elnphi"], c["K"], c["V"])
        assert v == pytest.approx(c["L"], rel=1e-12)
        assert g[0] == pytest.approx(c["grad"], rel=1e-12)


@pytest.mark.parametrize("imm", [False, True])
def test_calculate_Ndivzeta_sumtheta(mmm, kats, imm):      # test/mmctm.jl:59-90; test/immctm.jl:80-110
    model = _toy(mmm, kats, imm)
    model.ζ = kats["calc_Ndivzeta"]["zeta"]
    np.testing.assert_allclose(mmm.calculate_Ndivζ(model, 0), kats["calc_Ndivzeta"]["doc1"], rtol=1e-15)
    model = _toy(mmm, kats, imm)
    model.θ[0] = [arr(t) for t in kats["calc_sumtheta"]["theta_doc1"]]
    np.testing.assert_allclose(mmm.calculate_sumθ(model, 0), kats["calc_sumtheta"]["doc1"], rtol=1e-14)


def test_modality_loglikelihood_free(mmm, kats):           # test/mmctm.jl:349-380
    k = kats["loglik_mmctm"]
    X = [[arr(xm).astype(np.int64) for xm in xd] for xd in kats["corpora"]["X_mm"]]
    Xm1 = [X[d][0] for d in range(2)]
    phi = [arr(g) / arr(g).sum() for g in k["gamma_m1"]]
    assert mmm.calculate_docmodality_loglikelihood(Xm1[0], k["props"][0], phi) == pytest.approx(k["docmodality_ll_d1"], rel=1e-13)
    assert mmm.calculate_modality_loglikelihood(Xm1, k["props"], phi) == pytest.approx(k["modality_ll_m1"], rel=1e-13)
    # a document without counts does not enter (MMCTM.jl:409: `if doc_N > 0`)
    Xe = Xm1 + [np.zeros((0, 2), dtype=np.int64)]
    assert mmm.calculate_modality_loglikelihood(Xe, k["props"] + [[0.5, 0.5]], phi) == pytest.approx(k["modality_ll_m1"], rel=1e-13)


def test_modality_loglikelihood_features_free(mmm, kats):  # test/immctm.jl:350-386
    k = kats["loglik_immctm"]
    X = [[arr(xm).astype(np.int64) for xm in xd] for xd in kats["corpora"]["X_mm"]]
    Xm1 = [X[d][0] for d in range(2)]
    phi = [[arr(gi) / arr(gi).sum() for gi in gk] for gk in k["gamma_m1"]]
    res = mmm.calculate_modality_loglikelihood(Xm1, k["eta"], phi, features=kats["corpora"]["features"][0])
    assert res == pytest.approx(k["modality_ll_m1"], rel=1e-13)


def test_lda_free_loglikelihood_equals_the_fit_s(mmm):     # LDA.jl:174-196: calculate_loglikelihood(X, θ, β) on a fitted model's θ, β
    X, lam0 = np_ref.synth_lda(300, 96, 10, seed=3, mean_n=800)
    g = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    ll = mmm.fit(g, maxiter=5, tol=0.0, verbose=False)
    assert mmm.calculate_loglikelihood(X, g.θ, g.β) == pytest.approx(ll[-1], rel=1e-12)


def test_lda_update_Eln_alone(mmm, oracle):                # LDA.jl:78-80, 96-98
    from scipy.special import digamma
    X, lam0 = np_ref.synth_lda(50, 40, 6, seed=5, mean_n=300)
    g = mmm.LDA(6, 0.1, 0.1, 40, X, λ0=lam0)
    gam = np.random.default_rng(1).uniform(0.05, 30.0, size=(6, 50))
    g.γ = gam
    mmm.update_Elnθ(g)
    np.testing.assert_allclose(g.Elnθ, digamma(gam) - digamma(gam.sum(axis=0)), rtol=1e-12, atol=1e-13)
    lam = np.random.default_rng(2).uniform(0.05, 300.0, size=(40, 6))
    g.λ = lam
    mmm.update_Elnβ(g)
    np.testing.assert_allclose(g.Elnβ, digamma(lam) - digamma(lam.sum(axis=0)), rtol=1e-12, atol=1e-13)
    mmm.update_ϕ(g)                                         # ... and update_ϕ! reads the refreshed tables
    e = np.exp(g.Elnθ[:, 0][:, None] + g.Elnβ[X[0][:, 0] - 1, :].T)
    np.testing.assert_allclose(g.ϕ[0], e / e.sum(axis=0), rtol=1e-12)


@pytest.mark.parametrize("imm", [False, True])
def test_per_document_calls_equal_the_all_documents_launch(mmm, imm):
    """fitdoc!(model, d) for one d = that document's row of the all-documents stage sequence, every other document untouched"""
    feats = None
    if imm:
        f = np.array([[a + 1, b + 1] for a in range(4) for b in range(3)])
        feats = [f, f[:8]]
    X, g0 = np_ref.synth_mm(30, [12, 8], [3, 2], seed=9, means=[200, 50], empty_frac=0.1)
    def make():
        if imm:
            return mmm.IMMCTM([3, 2], [0.1, 0.1], feats, X, seed=3)
        return mmm.MMCTM([3, 2], [0.1, 0.1], [12, 8], X, γ0=g0)
    a, b = make(), make()
    mmm.fit(a, maxiter=2, tol=0.0, verbose=False); mmm.fit(b, maxiter=2, tol=0.0, verbose=False)
    lam0, nu0, z0 = a.lam_matrix().copy(), a._get("nu").copy(), a._get("zeta").copy()
    mmm.update_ζ(b); mmm.update_θ(b); mmm.update_ν(b); mmm.update_λ(b)          # every document
    d = 4
    mmm.update_ζ(a, d); mmm.update_θ(a, d); mmm.update_ν(a, d); mmm.update_λ(a, d)
    lamA, lamB = a.lam_matrix(), b.lam_matrix()
    assert np.array_equal(lamA[d], lamB[d]) and np.array_equal(a.ν[d], b.ν[d]) and np.array_equal(a.ζ[d], b.ζ[d])
    for m in range(2):
        assert np.array_equal(a.θ[d][m], b.θ[d][m])
    keep = np.arange(30) != d
    assert np.array_equal(lamA[keep], lam0[keep])
    assert np.array_equal(a._get("nu").reshape(30, 5)[keep], nu0.reshape(30, 5)[keep])
    assert np.array_equal(a._get("zeta").reshape(30, 2)[keep], z0.reshape(30, 2)[keep])
