"""Frozen-topic inference: transform / fit_heldout / predict_modality_η on the HIP backend against the CPU
oracle's restatement of LDA.jl:226-295, MMCTM.jl:496-634, IMMCTM.jl:468-545, and the reference's own test of `transform`
(test/mmctm.jl:390-406)."""
import warnings

import numpy as np
import pytest

import np_ref
from test_ctm_gpu import SNV3, _toy

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------------------------ LDA
def _lda_pair(mmm, oracle, D=120, V=96, K=6, seed=4):
    X, lam0 = np_ref.synth_lda(D, V, K, seed=seed, mean_n=800)
    g = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
    o = oracle.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0)
    mmm.fit(g, maxiter=25, tol=1e-4, verbose=False)
    o.fit(maxiter=25, tol=1e-4)
    Xn, _ = np_ref.synth_lda(70, V, K, seed=seed + 100, mean_n=500)
    return g, o, Xn


def test_lda_transform(mmm, oracle):
    g, o, Xn = _lda_pair(mmm, oracle)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        th_g = mmm.transform(g, Xn, maxiter=40, tol=1e-5)
    th_o, onew = o.transform(Xn, maxiter=40, tol=1e-5)
    K, Dn = g.K, len(Xn)
    assert th_g.shape == (K, Dn)
    np.testing.assert_allclose(th_g, th_o.reshape(Dn, K).T, rtol=1e-9)
    np.testing.assert_allclose(th_g.sum(axis=0), 1.0, rtol=1e-13)


def test_lda_transform_and_heldout_on_dense_row_handles(mmm, oracle, tuning):
    """Handles that took the dense-row E-step build (forced): the frozen-topic passes and the stage API keep to the CSR sweeps."""
    tuning(lda_build="dense")
    g, o, Xn = _lda_pair(mmm, oracle, D=200, K=10, seed=6)
    assert g.geometry()["dense"] == 1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        th_g = mmm.transform(g, Xn, maxiter=40, tol=1e-5)
    th_o, _ = o.transform(Xn, maxiter=40, tol=1e-5)
    np.testing.assert_allclose(th_g, th_o.reshape(len(Xn), g.K).T, rtol=1e-9)
    hg = mmm.fit_heldout(Xn, g, maxiter=30)
    ho = o.fit_heldout(Xn, maxiter=30)
    assert len(hg.ll_history) == len(ho.ll_hist)
    np.testing.assert_allclose(hg.ll_history, ho.ll_hist, rtol=1e-10)
    # stage API after fused passes on the same handle
    mmm.update_γ(g); o.update_gamma(); mmm.update_ϕ(g); o.update_phi(); mmm.update_λ(g); o.update_lambda()
    np.testing.assert_allclose(g.λ, o.lam.reshape(g.V, g.K, order="F"), rtol=1e-8)


def test_lda_transform_stops_like_the_reference(mmm, oracle):
    g, o, Xn = _lda_pair(mmm, oracle, seed=9)
    th_o, onew = o.transform(Xn, maxiter=200, tol=1e-4)
    assert onew.converged and 11 <= len(onew.ll_hist) < 200
    new = mmm.LDA(g.K, g.α, g.η, g.V, Xn, seed=3)
    new.β = g.β
    from multimodalmusig_jl_amd.inference import _lda_infer
    hist = _lda_infer(new, True, 200, 1e-4, False)
    assert new.converged and len(hist) == len(onew.ll_hist)
    np.testing.assert_allclose(hist, onew.ll_hist, rtol=1e-10)
    np.testing.assert_allclose(new.θ, th_o.reshape(len(Xn), g.K).T, rtol=1e-9)
    np.testing.assert_allclose(new.phi_flat(), onew.phi.reshape(-1, g.K), rtol=1e-9, atol=1e-300)     # unsmoothed ϕ of the last pass


def test_lda_fit_heldout(mmm, oracle):
    g, o, Xn = _lda_pair(mmm, oracle, seed=14)
    hg = mmm.fit_heldout(Xn, g, maxiter=60)
    ho = o.fit_heldout(Xn, maxiter=60)
    assert hg.converged == ho.converged and len(hg.ll_history) == len(ho.ll_hist)
    np.testing.assert_allclose(hg.ll_history, ho.ll_hist, rtol=1e-10)
    K, Dn = g.K, len(Xn)
    np.testing.assert_allclose(hg.γ, ho.gamma.reshape(Dn, K).T, rtol=1e-9)
    np.testing.assert_allclose(hg.θ, ho.theta.reshape(Dn, K).T, rtol=1e-9)
    np.testing.assert_allclose(hg.phi_flat(), ho.phi.reshape(-1, K), rtol=1e-9, atol=1e-300)
    assert hg.elbo == pytest.approx(ho.elbo_value, rel=1e-9)
    assert hg.ll == pytest.approx(ho.ll, rel=1e-10)
    # the trained model is untouched and can go on training
    np.testing.assert_allclose(g.β, o.beta.reshape(g.V, K, order="F"), rtol=1e-9)


# ------------------------------------------------------------------------------------------------------------------ CTM
def test_transform_reference_test(mmm, kats):                 # test/mmctm.jl:390-406
    model = _toy(mmm, kats)
    mmm.fit(model, maxiter=1, verbose=False)
    X = model.X
    new = mmm.transform(model, X, maxiter=1, fit_gaussian=False)
    assert len(new.ll) == 2
    assert np.all(new.Σ == model.Σ)
    new = mmm.transform(model, X, maxiter=1, fit_gaussian=True)
    assert np.any(new.Σ != model.Σ)


def _ctm_trained(mmm, oracle, feats=None, seed=21, shape=None):
    if shape is not None:
        K, V, means = shape
    elif feats is None:
        K, V, means = [5, 4], [40, 24], [600, 80]
    else:
        K, V, means = [6], [96], [1200]
    X, g0 = np_ref.synth_mm(90, V, K, seed=seed, means=means, empty_frac=0.1)
    alpha = [0.1] * len(K)
    if feats is None:
        g0f = np.concatenate([x.ravel() for x in g0])
        g = mmm.MMCTM(K, alpha, V, X, γ0=g0)
        o = oracle.CtmOracle(K, alpha, X, V=V, gamma0=g0f)
    else:
        GM = sum(K[m] * int(np.asarray(feats[m]).max(axis=0).sum()) for m in range(len(K)))
        g0f = np.random.default_rng(seed).integers(1, 101, size=GM).astype(np.float64)
        g = mmm.IMMCTM(K, alpha, feats, X, γ0=g0f)
        o = oracle.CtmOracle(K, alpha, X, features=feats, gamma0=g0f)
    mmm.fit(g, maxiter=8, tol=0.0, verbose=False)
    # inference parity is checked from IDENTICAL trained globals: the oracle model takes the GPU model's
    o.fit(maxiter=1, tol=0.0)
    o.mu[:] = g.μ; o.Sigma[:] = g.Σ.ravel(order="F"); o.invSigma[:] = g.invΣ.ravel(order="F")
    o.gamma[:] = g._get("gamma"); o.Elnphi[:] = g._get("Elnphi")
    if feats is None:
        o.phi[:] = g._get("phi")
    Xn, _ = np_ref.synth_mm(60, V, K, seed=seed + 50, means=means, empty_frac=0.1)
    return g, o, Xn, K, V, alpha


def _fresh_oracle(oracle, o, Xn, K, V, alpha, feats=None, mods=None, geometry=None):
    """geometry (of the device handle that ran the same inference): the order-matched variant, which the device must equal bit for bit"""
    mods = list(range(len(K))) if mods is None else mods
    Km = [K[m] for m in mods]
    if feats is None:
        return oracle.CtmOracle(Km, [alpha[m] for m in mods], Xn, V=[V[m] for m in mods], seed=5, geometry=geometry)
    return oracle.CtmOracle(Km, [alpha[m] for m in mods], Xn, features=[feats[m] for m in mods], seed=5, geometry=geometry)


def _same_bits(a, b, what):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel(); b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    n = int((a.view(np.int64) != b.view(np.int64)).sum())
    assert n == 0, "%s: %d of %d values differ in their bits" % (what, n, a.size)


def _cmp_docs_exact(gn, on):
    """against the order-matched oracle: the documents' state in every bit, evaluation counts of the last pass equal"""
    D = on.D
    _same_bits(gn.lam_matrix(), on.lam, "lambda"); _same_bits(gn.nu_matrix(), on.nu, "nu"); _same_bits(gn._get("zeta"), on.zeta, "zeta")
    np.testing.assert_allclose(gn._get("theta"), on.theta, rtol=1e-12, atol=1e-300)
    st = gn.solver_stats(per_doc=True)
    assert np.array_equal(st["per_doc_nu"], on.nev_nu[:D]) and np.array_equal(st["per_doc_lambda"], on.nev_lambda[:D])


def _cmp_docs(gn, on, first_pass):
    D, MK, M = on.D, on.MK, on.M
    if first_pass:
        # Held-out documents start from λ = 0, ν = 1 under a trained (ill-conditioned) Σ⁻¹: the objective is flat along some
        # directions, LD_MMA's successive-iterate test (xtol 1e-4) fires an iteration apart more often than in training and the
        # two stopping points are up to several 1e-2 apart (the λ solves of those documents run 140-200 evaluations, some to
        # the cap; the device writes the step as one quotient, ctm_estep.cuh, the index-order oracle as NLopt does).  Documents
        # that took the same number of evaluations agree to the x-tolerance level or better, and most do.
        st = gn.solver_stats(per_doc=True)
        same = (np.abs(st["per_doc_nu"]) == np.abs(on.nev_nu[:D])) & (np.abs(st["per_doc_lambda"]) == np.abs(on.nev_lambda[:D]))
        assert same.mean() >= 0.8, "only %.2f of the documents took the literal oracle's evaluation counts" % same.mean()
        for a, b in ((gn.lam_matrix(), on.lam.reshape(D, MK)), (gn.nu_matrix(), on.nu.reshape(D, MK))):
            rows = (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max(axis=1)
            assert rows[same].max() < 1e-5, "documents with equal evaluation counts off by %g" % rows[same].max()
            assert rows.max() < 0.2, "worst document off by %g" % rows.max()
            assert np.mean(rows < 1e-7) >= 0.8
        np.testing.assert_allclose(gn._get("zeta").reshape(D, M), on.zeta.reshape(D, M), rtol=1e-9)
        np.testing.assert_allclose(gn._get("theta"), on.theta, rtol=1e-9, atol=1e-300)
    else:
        lam_o = on.lam.reshape(D, MK)
        err = (np.abs(gn.lam_matrix() - lam_o) / np.maximum(1.0, np.abs(lam_o))).max(axis=1)
        assert np.median(err) < 1e-3 and err.max() < 5e-2
        th = np.abs(gn._get("theta") - on.theta)
        assert np.median(th) < 1e-4


@pytest.mark.parametrize("fit_gaussian", [False, True])
def test_mmctm_transform(mmm, oracle, fit_gaussian):
    g, o, Xn, K, V, alpha = _ctm_trained(mmm, oracle)
    flags = 1 | (2 if fit_gaussian else 0)
    for maxiter, first in ((1, True), (14, False)):
        gn = mmm.transform(g, Xn, maxiter=maxiter, tol=1e-9, fit_gaussian=fit_gaussian)
        on = _fresh_oracle(oracle, o, Xn, K, V, alpha)
        on.phi[:] = o.phi                                              # MMCTM.jl:516
        if not fit_gaussian:
            on.mu[:] = o.mu; on.Sigma[:] = o.Sigma                     # MMCTM.jl:518-521: invΣ stays the identity
        ll_o = on.infer(flags, maxiter, 1e-9)
        assert gn.ll_history.shape == ll_o.shape
        np.testing.assert_allclose(gn.ll_history, ll_o, rtol=1e-6 if first else 1e-5)   # after an MMA solve: x-tolerance level, not round-off
        np.testing.assert_allclose(gn.ll, ll_o[-1], rtol=1e-5)
        _cmp_docs(gn, on, first)
        if first:
            np.testing.assert_allclose(gn._get("props"), on.props, rtol=1e-6, atol=1e-9)
        # the same inference by the order-matched oracle: bit-identical
        ot = _fresh_oracle(oracle, o, Xn, K, V, alpha, geometry=gn.geometry())
        ot.phi[:] = o.phi; ot.Elnphi[:] = o.Elnphi
        if not fit_gaussian:
            ot.mu[:] = o.mu; ot.Sigma[:] = o.Sigma
        ll_t = ot.infer(flags, maxiter, 1e-9)
        np.testing.assert_allclose(gn.ll_history, ll_t, rtol=1e-10)
        _cmp_docs_exact(gn, ot)
        if fit_gaussian:
            _same_bits(gn.μ, ot.mu, "mu"); _same_bits(np.asarray(gn.invΣ).ravel(order="F"), ot.invSigma, "invSigma")
        MK = sum(K)
        if fit_gaussian:
            # 60 documents: one document whose MMA solve stopped an iteration apart (|Δλ| < 2e-3, test_ctm_gpu.py docstring)
            # moves an entry of Σ by ~2 λ Δλ / D ~ 1e-4
            np.testing.assert_allclose(gn.Σ, on.Sigma.reshape(MK, MK), rtol=2e-3, atol=5e-4)
            np.testing.assert_allclose(gn.μ, on.mu, rtol=2e-3, atol=5e-4)
        else:
            assert np.array_equal(gn.invΣ, np.eye(MK)) and np.array_equal(gn.Σ, g.Σ)
        gn.close()


def test_mmctm_transform_default_tol_stops_at_pass_11(mmm, oracle):
    g, o, Xn, K, V, alpha = _ctm_trained(mmm, oracle)
    gn = mmm.transform(g, Xn)                                          # tol = 1e4 (MMCTM.jl:512)
    assert gn.converged and len(gn.ll_history) == 11


@pytest.mark.parametrize("case", ["mm", "imm"])
def test_fit_heldout_ctm(mmm, oracle, case):
    feats = SNV3 if case == "imm" else None
    g, o, Xn, K, V, alpha = _ctm_trained(mmm, oracle, feats=feats)
    for maxiter, first in ((1, True), (30, False)):
        gn = mmm.fit_heldout(Xn, g, maxiter=maxiter)
        on = _fresh_oracle(oracle, o, Xn, K, V, alpha, feats=feats)
        on.mu[:] = o.mu; on.Sigma[:] = o.Sigma; on.invSigma[:] = o.invSigma       # MMCTM.jl:558-560
        on.gamma[:] = o.gamma; on.Elnphi[:] = o.Elnphi                             # :561-562
        if feats is None:
            on.phi[:] = o.phi                                                      # :563
        ll_o = on.infer(0, maxiter, 1e-4)
        assert gn.converged == on.converged and gn.ll_history.shape == ll_o.shape
        np.testing.assert_allclose(gn.ll_history, ll_o, rtol=1e-6 if first else 1e-5)   # after an MMA solve: x-tolerance level, not round-off
        _cmp_docs(gn, on, first)
        # the same inference by the order-matched oracle: bit-identical
        ot = _fresh_oracle(oracle, o, Xn, K, V, alpha, feats=feats, geometry=gn.geometry())
        ot.mu[:] = o.mu; ot.Sigma[:] = o.Sigma; ot.invSigma[:] = o.invSigma; ot.gamma[:] = o.gamma; ot.Elnphi[:] = o.Elnphi
        if feats is None:
            ot.phi[:] = o.phi
        ll_t = ot.infer(0, maxiter, 1e-4)
        assert gn.converged == ot.converged and gn.ll_history.shape == ll_t.shape
        np.testing.assert_allclose(gn.ll_history, ll_t, rtol=1e-10)
        _cmp_docs_exact(gn, ot)
        gn.close()


@pytest.mark.parametrize("case", ["mm", "imm"])
def test_predict_modality_eta(mmm, oracle, case):
    """Three modalities, predict the η block of the middle one from the other two."""
    seed = 33
    K, V, means = [4, 3, 3], [30, 20, 16], [300, 100, 80]
    feats = None
    if case == "imm":
        feats = [np.array([[v // 5 + 1, v % 5 + 1] for v in range(30)]), np.array([[v // 4 + 1, v % 4 + 1] for v in range(20)]),
                 np.array([[v // 4 + 1, v % 4 + 1] for v in range(16)])]
    X, g0 = np_ref.synth_mm(80, V, K, seed=seed, means=means)
    alpha = [0.1, 0.1, 0.1]
    if feats is None:
        g = mmm.MMCTM(K, alpha, V, X, γ0=g0)
    else:
        g = mmm.IMMCTM(K, alpha, feats, X, seed=2)
    mmm.fit(g, maxiter=10, tol=0.0, verbose=False)
    m = 1
    Xobs = [[d[0], d[2]] for d in X[:40]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eta = mmm.predict_modality_η(Xobs, m, g, maxiter=25)
    assert len(eta) == 40 and all(e.shape == (K[m],) for e in eta)
    # oracle: same construction (MMCTM.jl:588-634) from the GPU model's globals
    μ, Σ, invΣ = g.μ, g.Σ, g.invΣ
    koff = np.concatenate([[0], np.cumsum(K)])
    un = np.arange(koff[m], koff[m + 1]); ob = np.setdiff1d(np.arange(koff[-1]), un)
    if feats is None:
        on = oracle.CtmOracle([K[0], K[2]], [0.1, 0.1], Xobs, V=[V[0], V[2]], seed=5)
        goff = g._goff
    else:
        on = oracle.CtmOracle([K[0], K[2]], [0.1, 0.1], Xobs, features=[feats[0], feats[2]], seed=5)
        goff = g._mgoff
    on.mu[:] = μ[ob]; on.Sigma[:] = Σ[np.ix_(ob, ob)].ravel(order="F"); on.invSigma[:] = invΣ[np.ix_(ob, ob)].ravel(order="F")
    for name, dst in (("gamma", on.gamma), ("Elnphi", on.Elnphi)):
        flat = g._get(name)
        dst[:] = np.concatenate([flat[goff[i]:goff[i + 1]] for i in (0, 2)])
    if feats is None:
        flat = g._get("phi")
        on.phi[:] = np.concatenate([flat[goff[i]:goff[i + 1]] for i in (0, 2)])
    on.infer(0, 25, 1e-4)
    A = Σ[np.ix_(un, ob)] @ invΣ[np.ix_(ob, ob)]
    lam_o = on.lam.reshape(len(Xobs), on.MK)
    eta_o = np.stack([μ[un] + A @ (lam_o[d] - μ[ob]) for d in range(len(Xobs))])
    err = np.abs(np.stack(eta) - eta_o) / np.maximum(1.0, np.abs(eta_o))
    assert np.median(err) < 1e-3 and err.max() < 5e-2


def test_fit_heldout_with_more_than_64_coordinates(mmm, oracle):
    """Frozen-topic inference through the generic kernels of csrc/ctm_big.cuh (sum K = 80): the first pass of fit_heldout from the trained
    globals against the index-order oracle -- zeta / theta exactly (they precede the solves), ll at the solver's x-tolerance level."""
    g, o, Xn, K, V, alpha = _ctm_trained(mmm, oracle, shape=([40, 40], [96, 48], [2500, 400]))
    assert g.geometry()["cpl"] == 4
    gn = mmm.fit_heldout(Xn, g, maxiter=1)
    on = _fresh_oracle(oracle, o, Xn, K, V, alpha)
    on.mu[:] = o.mu; on.Sigma[:] = o.Sigma; on.invSigma[:] = o.invSigma; on.gamma[:] = o.gamma; on.Elnphi[:] = o.Elnphi; on.phi[:] = o.phi
    ll_o = on.infer(0, 1, 1e-4)
    assert gn.ll_history.shape == ll_o.shape
    np.testing.assert_allclose(gn.ll_history, ll_o, rtol=1e-5)
    D, M = on.D, on.M
    np.testing.assert_allclose(gn._get("zeta").reshape(D, M), on.zeta.reshape(D, M), rtol=1e-9)
    np.testing.assert_allclose(gn._get("theta"), on.theta, rtol=1e-9, atol=1e-300)
    gn.close()
