"""The wide-vocabulary LDA path (K·V tables larger than LDS: one wave per document, tables through L2, ϕ written out, topic
statistics by a term-major posting sweep) against the CPU oracle and against the LDS path on the same inputs.
`lda_build = MMM_BUILD_WIDE` (mmm_ctx_set_tuning; read at create) forces the path for shapes the LDS path also handles, so the two can be compared directly."""
import warnings

import numpy as np
import pytest

import np_ref
from test_lda_gpu import _cmp_state, _pair

pytestmark = pytest.mark.gpu


def _wide_pair(mmm, oracle, tuning, *a, **kw):
    tuning(lda_build="wide")
    out = _pair(mmm, oracle, *a, **kw)
    tuning()
    return out


@pytest.mark.parametrize("D,V,K", [(101, 96, 10), (37, 24, 5), (40, 30, 32), (45, 96, 13)])
def test_forced_wide_stage_api_and_fit(mmm, oracle, tuning, D, V, K):
    X, g, o = _wide_pair(mmm, oracle, tuning, D, V, K, seed=100 + D, empty=(1, D - 1))
    mmm.update_γ(g); o.update_gamma()
    mmm.update_ϕ(g); o.update_phi()
    mmm.update_λ(g); o.update_lambda()
    mmm.update_β(g); o.update_beta()
    mmm.update_θ(g); o.update_theta()
    _cmp_state(g, o, 1e-11)
    assert mmm.calculate_loglikelihood(g) == pytest.approx(o.loglik(), rel=1e-11)
    X, g, o = _wide_pair(mmm, oracle, tuning, D, V, K, seed=100 + D, empty=(1, D - 1))     # (the oracle's fit starts from the constructor state)
    ll_g = mmm.fit(g, maxiter=13, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=13, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    _cmp_state(g, o, 1e-8)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-8)


def test_wide_equals_lds_path(mmm, oracle, tuning):
    """Same corpus, same λ0 through both data flows: same stopping pass, ll history and state to summation-order accuracy;
    and the wide path is deterministic (posting-order sums, no atomics)."""
    X, lam0 = np_ref.synth_lda(300, 96, 10, seed=5, mean_n=900)
    a = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    tuning(lda_build="wide")
    b = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    c = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    tuning()
    la, lb, lc = (mmm.fit(m, maxiter=300, tol=1e-5, verbose=False) for m in (a, b, c))
    assert len(la) == len(lb) and a.converged and b.converged
    np.testing.assert_allclose(la, lb, rtol=1e-11)
    np.testing.assert_allclose(a.λ, b.λ, rtol=1e-9)
    np.testing.assert_allclose(a.θ, b.θ, rtol=1e-9)
    assert a.elbo == pytest.approx(b.elbo, rel=1e-10)
    np.testing.assert_array_equal(lb, lc)
    np.testing.assert_array_equal(b.λ, c.λ)


@pytest.mark.parametrize("D,V,K,mean_n", [(150, 1536, 10, 3000), (60, 6000, 24, 2500)])
def test_vocabularies_beyond_lds(mmm, oracle, D, V, K, mean_n):
    """1536 pentanucleotide contexts × K = 10 (123 KB of table per copy) and a 6000-term vocabulary × K = 24 (1.15 MB): the shapes
    the LDS path cannot hold.  Early stop included: the reference's rule, evaluated on the device."""
    X, g, o = _pair(mmm, oracle, D, V, K, seed=40 + K, mean_n=mean_n, empty=(3,))
    ll_g = mmm.fit(g, maxiter=60, tol=1e-4, verbose=False)
    ll_o = o.fit(maxiter=60, tol=1e-4)
    assert len(ll_g) == len(ll_o) and g.converged == o.converged
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, K), rtol=1e-5, atol=1e-12)       # north-star bar
    np.testing.assert_allclose(g.θ, o.theta.reshape(D, K).T, rtol=1e-5)
    np.testing.assert_allclose(g.λ, o.lam.reshape(V, K, order="F"), rtol=1e-7)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-7)
    N = np.array([x[:, 1].sum() for x in X], dtype=np.float64)
    np.testing.assert_allclose(g.γ.sum(axis=0), K * 0.1 + N, rtol=1e-12)                         # mass conservation
    assert g.λ.sum() == pytest.approx(V * K * 0.1 + N.sum(), rel=1e-12)


def test_wide_inference(mmm, oracle):
    """transform / fit_heldout of a wide model: the frozen-topic passes run the same wide E-step without the statistics sweep."""
    D, V, K = 90, 2000, 8
    X, g, o = _pair(mmm, oracle, D, V, K, seed=8, mean_n=1500)
    mmm.fit(g, maxiter=20, tol=1e-4, verbose=False)
    o.fit(maxiter=20, tol=1e-4)
    Xn, _ = np_ref.synth_lda(40, V, K, seed=108, mean_n=700)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        th_g = mmm.transform(g, Xn, maxiter=30, tol=1e-5)
    th_o, _ = o.transform(Xn, maxiter=30, tol=1e-5)
    np.testing.assert_allclose(th_g, th_o.reshape(len(Xn), K).T, rtol=1e-9)
    hg = mmm.fit_heldout(Xn, g, maxiter=40)
    ho = o.fit_heldout(Xn, maxiter=40)
    assert len(hg.ll_history) == len(ho.ll_hist)
    np.testing.assert_allclose(hg.ll_history, ho.ll_hist, rtol=1e-10)
    np.testing.assert_allclose(hg.θ, ho.theta.reshape(len(Xn), K).T, rtol=1e-9)
    assert hg.elbo == pytest.approx(ho.elbo_value, rel=1e-9)


def test_wide_ilda(mmm, oracle, tuning):
    """ILDA on the wide data flow (effective V×K tables, statistics folded onto feature values by the same M-step kernel)."""
    from test_ilda_gpu import _pair as ilda_pair
    tuning(lda_build="wide")
    X, g, o = ilda_pair(mmm, oracle, 80, 6, seed=31)
    tuning()
    ll_g = mmm.fit(g, maxiter=14, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=14, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-8)


def test_wide_degenerate_shapes(mmm, oracle, tuning):
    """K = 1, a single document, all documents empty but one, unused vocabulary tail, and a term listed twice in a document."""
    for D, V, K, empty in [(5, 7, 1, ()), (1, 96, 10, ()), (6, 30, 4, (0, 1, 2, 4, 5)), (3, 200, 2, ())]:
        X, g, o = _wide_pair(mmm, oracle, tuning, D, V, K, seed=900 + D + K, mean_n=50, empty=empty)
        ll_g = mmm.fit(g, maxiter=4, tol=0.0, verbose=False)
        ll_o = o.fit(maxiter=4, tol=0.0)
        np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
        _cmp_state(g, o, 1e-9)
        assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
        g.close()
    X = [np.array([[1, 3], [2, 5], [1, 2]]), np.array([[2, 4], [3, 1]])]              # term 1 twice in document 0
    lam0 = np.random.default_rng(1).integers(1, 101, size=(3, 2)).astype(np.float64)
    tuning(lda_build="wide")
    g = mmm.LDA(2, 0.1, 0.1, 3, X, λ0=lam0)
    tuning()
    o = oracle.LdaOracle(2, 0.1, 0.1, X, V=3, lambda0=lam0)
    np.testing.assert_allclose(mmm.fit(g, maxiter=5, tol=0.0, verbose=False), o.fit(maxiter=5, tol=0.0), rtol=1e-11)
    np.testing.assert_allclose(g.λ, o.lam.reshape(3, 2, order="F"), rtol=1e-11)


@pytest.mark.parametrize("D,V,K,mean_n", [(300, 96, 48, 3000), (150, 50, 33, 400), (120, 200, 64, 2500), (90, 1536, 40, 3000), (100, 96, 65, 3000), (80, 120, 100, 2000),
                                           (60, 300, 129, 4000), (40, 96, 256, 3000)])
def test_more_than_32_topics(mmm, oracle, D, V, K, mean_n):
    """The reference has no limit on K (LDA.jl:24-54).  33..256 topics run the two sweeps of the wide path with rolled topic loops
    (k_lda_estep_big, k_lda_stats_big; beyond 64 topics a lane holds topics l, l + 64, ... in the per-document kernels): stage sequence and
    whole fits against the oracle, early stop included."""
    X, g, o = _pair(mmm, oracle, D, V, K, seed=70 + K, mean_n=mean_n, empty=(2, D - 1))
    assert g.geometry()["wide"] == 1
    mmm.update_γ(g); o.update_gamma()
    mmm.update_ϕ(g); o.update_phi()
    mmm.update_λ(g); o.update_lambda()
    mmm.update_β(g); o.update_beta()
    mmm.update_θ(g); o.update_theta()
    _cmp_state(g, o, 1e-11)
    assert mmm.calculate_loglikelihood(g) == pytest.approx(o.loglik(), rel=1e-11)
    X, g, o = _pair(mmm, oracle, D, V, K, seed=70 + K, mean_n=mean_n, empty=(2, D - 1))
    ll_g = mmm.fit(g, maxiter=40, tol=1e-4, verbose=False)
    ll_o = o.fit(maxiter=40, tol=1e-4)
    assert len(ll_g) == len(ll_o) and g.converged == o.converged
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, K), rtol=1e-5, atol=1e-12)       # north-star bar
    np.testing.assert_allclose(g.θ, o.theta.reshape(D, K).T, rtol=1e-5)
    np.testing.assert_allclose(g.λ, o.lam.reshape(V, K, order="F"), rtol=1e-7)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-7)
    N = np.array([x[:, 1].sum() for x in X], dtype=np.float64)
    np.testing.assert_allclose(g.γ.sum(axis=0), K * 0.1 + N, rtol=1e-12)                         # mass conservation


def test_more_than_256_topics_is_refused(mmm):
    X, lam0 = np_ref.synth_lda(20, 30, 8, seed=1, mean_n=100)
    with pytest.raises(mmm.MmmError, match="max 256"):
        mmm.LDA(257, 0.1, 0.1, 30, X, λ0=np.ones((30, 257)))
