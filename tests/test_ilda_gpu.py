"""ILDA (src/ILDA.jl) on the HIP backend: the reference's known-answer tests (test/ilda.jl) replayed
through the C ABI, and whole fits / held-out inference against the CPU oracle's restatement."""
import numpy as np
import pytest

import np_ref

pytestmark = pytest.mark.gpu
SNV3 = np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])


def arr(x):
    return np.asarray(x, dtype=np.float64)


def _toy(mmm, kats, eta=0.1, **kw):
    c = kats["corpora"]
    X = [arr(x).astype(np.int64).reshape(-1, 2) for x in c["X_ilda"]]
    return mmm.ILDA(2, 0.1, eta, c["features_ilda"], X, seed=3, **kw)


def test_constructor(mmm, kats):                              # test/ilda.jl:24-51
    model = _toy(mmm, kats)
    assert (model.K, model.D, model.I) == (2, 2, 2) and model.J == [2, 2] and model.η.tolist() == [0.1, 0.1]
    assert len(model.λ) == 2 and model.λ[0].shape == (2, 2) and np.all(model.λ[0] > 0) and np.all(model.λ[1] > 0)
    assert len(model.Elnβ) == 2 and model.Elnβ[1].shape == (2, 2)
    assert model.γ.shape == (2, 2) and np.all(model.γ > 0) and model.Elnθ.shape == (2, 2)
    np.testing.assert_allclose(model.ϕ[0].sum(axis=0), np.ones(2))
    assert _toy(mmm, kats, eta=[0.01, 0.5]).η.tolist() == [0.01, 0.5]


def test_update_phi(mmm, kats):                               # test/ilda.jl:53-93
    k = kats["ilda_update_phi"]
    model = _toy(mmm, kats)
    model.Elnθ = arr(k["Elntheta"])
    model.Elnβ = [arr(k["Elnbeta"][0]), arr(k["Elnbeta"][1])]
    mmm.update_ϕ(model)
    np.testing.assert_allclose(model.ϕ[0], arr(k["phi"][0]), rtol=1e-12)
    np.testing.assert_allclose(model.ϕ[1], arr(k["phi"][1]), rtol=1e-12)


def test_update_gamma(mmm, kats):                             # test/ilda.jl:95-112
    k = kats["ilda_update_gamma"]
    model = _toy(mmm, kats)
    model.ϕ[0] = arr(k["phi_doc1"])
    mmm.update_γ(model)
    np.testing.assert_allclose(model.γ[:, 0], k["gamma_doc1"], rtol=1e-14)
    np.testing.assert_allclose(model.Elnθ[:, 0], k["Elntheta_doc1"], rtol=1e-12)


def test_update_lambda(mmm, kats):                            # test/ilda.jl:114-160
    k = kats["ilda_update_lambda"]
    model = _toy(mmm, kats, eta=k["eta"])
    model.ϕ = [arr(p) for p in k["phi"]]
    mmm.update_λ(model)
    for i in range(2):
        np.testing.assert_allclose(model.λ[i], arr(k["lambda"][i]), rtol=1e-13)
        np.testing.assert_allclose(model.Elnβ[i], arr(k["Elnbeta"][i]), rtol=1e-12)


def test_elbo_of_constructor_state(mmm, kats, oracle):        # test/ilda.jl:162-175
    model = _toy(mmm, kats)
    e, t = mmm.calculate_elbo(model, terms=True)
    assert np.all(np.isfinite(t)) and e < 0.0
    c = kats["corpora"]
    o = oracle.IldaOracle(2, 0.1, 0.1, c["features_ilda"], [arr(x).astype(np.int64).reshape(-1, 2) for x in c["X_ilda"]],
                          lambda0=np.concatenate([model.λ[i].ravel(order="F") for i in range(2)]))
    eo, to = o.elbo()
    np.testing.assert_allclose(t, to, rtol=1e-12)
    assert e == pytest.approx(eo, rel=1e-12)


def _pair(mmm, oracle, D, K, seed, eta=(0.1, 0.3, 0.2)):
    X, _ = np_ref.synth_lda(D, 96, K, seed=seed, mean_n=1200)
    J = SNV3.max(axis=0)
    lam0 = [np.random.default_rng(seed + 1).integers(1, 101, size=(int(j), K)).astype(np.float64) for j in J]
    g = mmm.ILDA(K, 0.1, list(eta), SNV3, X, λ0=lam0)
    o = oracle.IldaOracle(K, 0.1, list(eta), SNV3, X, lambda0=np.concatenate([l.ravel(order="F") for l in lam0]))
    return X, g, o


def test_stage_sequence_against_oracle(mmm, oracle):
    X, g, o = _pair(mmm, oracle, 60, 5, seed=11)
    for it in range(3):
        mmm.update_γ(g); mmm.update_ϕ(g); mmm.update_λ(g); mmm.update_β(g); mmm.update_θ(g)
        o.update_gamma(); o.update_phi(); o.update_lambda(); o.update_beta(); o.update_theta()
        np.testing.assert_allclose(g.γ, o.gamma.reshape(60, 5).T, rtol=1e-11)
        np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, 5), rtol=1e-10, atol=1e-300)
        for i in range(3):
            np.testing.assert_allclose(g.λ[i], o.mat(o.lam, i), rtol=1e-11)
            np.testing.assert_allclose(g.Elnβ[i], o.mat(o.Elnbeta, i), rtol=1e-10, atol=1e-13)
            np.testing.assert_allclose(g.β[i], o.mat(o.beta, i), rtol=1e-11)
        assert mmm.calculate_loglikelihood(g) == pytest.approx(o.loglik(), rel=1e-11)
    e, t = mmm.calculate_elbo(g, terms=True)
    eo, to = o.elbo()
    np.testing.assert_allclose(t, to, rtol=1e-10)


@pytest.mark.parametrize("K", [5, 10, 40])
def test_fit_matches_oracle(mmm, oracle, K):
    X, g, o = _pair(mmm, oracle, 150, K, seed=23)
    ll_g = mmm.fit(g, maxiter=80, tol=1e-4, verbose=False)
    ll_o = o.fit(maxiter=80, tol=1e-4)
    assert len(ll_g) == len(ll_o) and g.converged == o.converged
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    for i in range(3):
        np.testing.assert_allclose(g.λ[i], o.mat(o.lam, i), rtol=1e-7)
    np.testing.assert_allclose(g.θ, o.theta.reshape(150, K).T, rtol=1e-7)
    np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, K), rtol=1e-5, atol=1e-12)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
    assert g.ll == pytest.approx(ll_o[-1], rel=1e-9)


def test_fit_on_the_dense_row_estep_build(mmm, oracle, tuning):
    """ILDA shares the LDA E-step: the dense-row build (forced here; default for dense corpora of >= 192 documents per CU) and the reduce
    blocks joining the ll sweep (residency lowered so that the ll blocks loop) against the oracle."""
    tuning(lda_build="dense", resident_cap=64)
    X, g, o = _pair(mmm, oracle, 400, 10, seed=29)
    assert g.geometry()["dense"] == 1
    ll_g = mmm.fit(g, maxiter=30, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=30, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    for i in range(3):
        np.testing.assert_allclose(g.λ[i], o.mat(o.lam, i), rtol=1e-7)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)


def test_fit_heldout_and_transform(mmm, oracle):
    X, g, o = _pair(mmm, oracle, 120, 5, seed=31)
    mmm.fit(g, maxiter=20, tol=0.0, verbose=False); o.fit(maxiter=20, tol=0.0)
    Xn, _ = np_ref.synth_lda(50, 96, 5, seed=77, mean_n=600)
    hg = mmm.fit_heldout(Xn, g, maxiter=40)
    ho = o.fit_heldout(Xn, maxiter=40)
    assert hg.converged == ho.converged and len(hg.ll_history) == len(ho.ll_hist)
    np.testing.assert_allclose(hg.ll_history, ho.ll_hist, rtol=1e-7)
    np.testing.assert_allclose(hg.θ, ho.theta.reshape(50, 5).T, rtol=1e-6)
    assert hg.elbo == pytest.approx(ho.elbo_value, rel=1e-7)
    with pytest.raises(TypeError):
        mmm.transform(g, Xn)
