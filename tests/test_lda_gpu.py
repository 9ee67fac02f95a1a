"""GPU parity tests for the LDA path: the HIP backend (through the C ABI / host mirror) against (i) the
reference's own known-answer tests, (ii) the CPU oracle on seeded synthetic corpora, (iii) size-independent
invariants at the BASELINE size.  Tolerances: 1e-11 relative for a single sweep (summation order differs),
1e-5 relative (north-star bar) on phi/theta/ELBO after a fit -- both written where they are used."""
import numpy as np
import pytest

import np_ref

pytestmark = pytest.mark.gpu


def arr(x):
    return np.asarray(x, dtype=np.float64)


# ---------------------------------------------------------------------------------- reference KATs (test/lda.jl)
def test_constructor(mmm, kats):
    c = kats["corpora"]
    X = [arr(x).astype(np.int64) for x in c["X_lda"]]
    model = mmm.LDA(c["K_lda"], c["alpha_lda"], c["eta_lda"], X, seed=0)
    assert (model.K, model.D, model.V) == (2, 2, 2)
    assert list(model.N) == kats["lda_ctor"]["N"]
    assert model.λ.shape == (2, 2) and np.all(model.λ > 0)
    assert model.γ.shape == (2, 2) and np.all(model.γ > 0)
    np.testing.assert_allclose(model.ϕ[0].sum(axis=0), np.ones(2))
    model = mmm.LDA(c["K_lda"], c["alpha_lda"], c["eta_lda"], 3, X, seed=0)
    assert model.V == 3 and model.λ.shape == (3, 2)


def test_update_phi(mmm, kats):
    c, k = kats["corpora"], kats["lda_update_phi"]
    model = mmm.LDA(c["K_lda"], c["alpha_lda"], c["eta_lda"], c["X_lda"], seed=0)
    model.Elnθ = arr(k["Elntheta"])
    model.Elnβ = arr(k["Elnbeta"])
    mmm.update_ϕ(model)
    np.testing.assert_allclose(model.ϕ[0], arr(k["phi_doc1"]), rtol=1e-13)


def test_update_gamma(mmm, kats):
    c, k = kats["corpora"], kats["lda_update_gamma"]
    model = mmm.LDA(c["K_lda"], c["alpha_lda"], c["eta_lda"], c["X_lda"], seed=0)
    model.ϕ[0] = arr(k["phi_doc1"])
    mmm.update_γ(model)
    np.testing.assert_allclose(model.γ[:, 0], k["gamma_doc1"], rtol=1e-13)
    np.testing.assert_allclose(model.Elnθ[:, 0], k["Elntheta_doc1"], rtol=1e-12)


def test_update_lambda(mmm, kats):
    c, k = kats["corpora"], kats["lda_update_lambda"]
    model = mmm.LDA(c["K_lda"], c["alpha_lda"], c["eta_lda"], c["X_lda"], seed=0)
    model.ϕ = [arr(p) for p in k["phi"]]
    mmm.update_λ(model)
    np.testing.assert_allclose(model.λ, arr(k["lambda"]), rtol=1e-13)
    np.testing.assert_allclose(model.Elnβ, arr(k["Elnbeta"]), rtol=1e-12)


def test_calculate_elbo_negative_at_construction(mmm, kats):
    c = kats["corpora"]
    model = mmm.LDA(c["K_lda"], c["alpha_lda"], c["eta_lda"], c["X_lda"], seed=3)
    e, t = mmm.calculate_elbo(model, terms=True)
    assert e < 0.0 and np.all(np.isfinite(t))          # test/lda.jl:105-118


# ---------------------------------------------------------------------------------- differential vs the oracle
def _pair(mmm, oracle, D, V, K, seed, mean_n=400, alpha=0.1, eta=0.1, empty=()):
    X, lam0 = np_ref.synth_lda(D, V, K, seed=seed, mean_n=mean_n)
    for d in empty:
        X[d] = np.zeros((0, 2), dtype=np.int64)
    g = mmm.LDA(K, alpha, eta, V, X, λ0=lam0)
    o = oracle.LdaOracle(K, alpha, eta, X, V=V, lambda0=lam0)
    return X, g, o


def _cmp_state(g, o, rtol, phi=True):
    K, D, V = g.K, g.D, g.V
    np.testing.assert_allclose(g.γ, o.gamma.reshape(D, K).T, rtol=rtol)
    np.testing.assert_allclose(g.Elnθ, o.Elntheta.reshape(D, K).T, rtol=rtol, atol=1e-13)
    np.testing.assert_allclose(g.λ, o.lam.reshape(V, K, order="F"), rtol=rtol)
    np.testing.assert_allclose(g.Elnβ, o.Elnbeta.reshape(V, K, order="F"), rtol=rtol, atol=1e-13)
    if phi:
        np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, K), rtol=rtol, atol=1e-300)


@pytest.mark.parametrize("D,V,K", [(37, 24, 5), (64, 96, 7), (101, 96, 10), (9, 130, 3)])
def test_single_sweep_matches_oracle(mmm, oracle, D, V, K):
    X, g, o = _pair(mmm, oracle, D, V, K, seed=100 + D, empty=(1, D - 1))
    # stage API, one reference function at a time
    mmm.update_γ(g); o.update_gamma()
    mmm.update_ϕ(g); o.update_phi()
    mmm.update_λ(g); o.update_lambda()
    mmm.update_β(g); o.update_beta()
    mmm.update_θ(g); o.update_theta()
    _cmp_state(g, o, 1e-11)
    np.testing.assert_allclose(g.β, o.beta.reshape(V, K, order="F"), rtol=1e-11)
    np.testing.assert_allclose(g.θ, o.theta.reshape(D, K).T, rtol=1e-11)
    assert mmm.calculate_loglikelihood(g) == pytest.approx(o.loglik(), rel=1e-11)
    e, t = mmm.calculate_elbo(g, terms=True)
    eo, to = o.elbo()
    np.testing.assert_allclose(t, to, rtol=1e-10)
    assert e == pytest.approx(eo, rel=1e-10)


@pytest.mark.parametrize("D,V,K", [(50, 96, 10), (33, 40, 7)])
def test_fused_iterations_equal_stage_sequence_and_oracle(mmm, oracle, D, V, K):
    X, g, o = _pair(mmm, oracle, D, V, K, seed=7, empty=(0,))
    n = 11
    check = mmm._lib.check
    check(mmm.lib().mmm_lda_iterate(g._h, n), g.ctx.h, "iterate")
    for _ in range(n):
        o.update_gamma(); o.update_phi(); o.update_lambda(); o.update_beta(); o.update_theta()
    _cmp_state(g, o, 1e-9)
    assert mmm.calculate_elbo(g) == pytest.approx(o.elbo()[0], rel=1e-9)
    # continue with the stage API from the fused state: must stay on the oracle's trajectory
    mmm.update_γ(g); o.update_gamma()
    mmm.update_ϕ(g); o.update_phi()
    mmm.update_λ(g); o.update_lambda()
    _cmp_state(g, o, 1e-9)


def test_fit_matches_oracle(mmm, oracle):
    """fit! with the reference's stopping rule (>10 passes, |dll|/|ll| < tol): same number of passes, same
    ll history, phi/theta/ELBO within the north-star 1e-5 relative bar."""
    X, g, o = _pair(mmm, oracle, 120, 96, 10, seed=2026, mean_n=3000)
    ll_g = mmm.fit(g, maxiter=200, tol=1e-4, verbose=False)
    ll_o = o.fit(maxiter=200, tol=1e-4)
    assert len(ll_g) == len(ll_o) and g.converged == o.converged
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, 10), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(g.θ, o.theta.reshape(120, 10).T, rtol=1e-5)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-5)
    assert g.ll == pytest.approx(ll_o[-1], rel=1e-9)


def test_brca_like_counts_k7(mmm, oracle):
    """BASELINE config 1 shape (K=7, alpha=eta=0.1, 96 terms, heavy-tailed N) on synthetic counts."""
    X, g, o = _pair(mmm, oracle, 80, 96, 7, seed=1, mean_n=20000)
    ll_g = mmm.fit(g, maxiter=15, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=15, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-7)


# ---------------------------------------------------------------------------------- invariants at BASELINE size
def test_invariants_at_full_size(mmm):
    D, V, K = 10000, 96, 10
    X, lam0 = np_ref.synth_lda(D, V, K, seed=20261004)
    g = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
    ll = mmm.fit(g, maxiter=12, tol=0.0, verbose=False)
    assert len(ll) == 12 and np.all(np.isfinite(ll)) and ll[-1] > ll[0]
    N = np.array([x[:, 1].sum() for x in X], dtype=np.float64)
    # mass conservation: sum_k gamma[k,d] = K*alpha + N_d ; sum_{v,k} lambda = V*K*eta + sum N
    np.testing.assert_allclose(g.γ.sum(axis=0), K * 0.1 + N, rtol=1e-12)
    assert g.λ.sum() == pytest.approx(V * K * 0.1 + N.sum(), rel=1e-12)
    phi = g.phi_flat()
    np.testing.assert_allclose(phi.sum(axis=1), 1.0, rtol=1e-13)
    np.testing.assert_allclose(g.β.sum(axis=0), 1.0, rtol=1e-13)
    np.testing.assert_allclose(g.θ.sum(axis=0), 1.0, rtol=1e-13)
    # idempotence of materialisation and determinism of the fused path (fixed-order reductions)
    g2 = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
    ll2 = mmm.fit(g2, maxiter=12, tol=0.0, verbose=False)
    np.testing.assert_array_equal(ll, ll2)
    np.testing.assert_array_equal(g.λ, g2.λ)


def test_config_2_at_its_own_size_against_the_oracle(mmm, oracle):
    """BASELINE configs[1] exactly as bench.py runs it (10,000 x 96, K = 10, corpus seed 20261004: the single-step build
    k_lda_estep<10,16,false,96,true>) held against the oracle, not only through invariants: 12 passes -- ll history, γ, λ, Elnβ, ELBO at
    1e-9, ϕ / θ at the reference's 1e-5 -- then fit!(tol = 1e-4): same pass count, same `converged` (LDA.jl:198-224)."""
    D, V, K = 10000, 96, 10
    X, lam0 = np_ref.synth_lda(D, V, K, seed=20261004)
    g = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
    geo = g.geometry()
    assert geo["single_step"] == 1 and geo["KP"] == 10 and geo["L"] == 16 and not geo["dense"]      # the build the benchmark times
    o = oracle.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0)
    ll_g = mmm.fit(g, maxiter=12, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=12, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    np.testing.assert_allclose(g.γ, o.gamma.reshape(D, K).T, rtol=1e-9)
    np.testing.assert_allclose(g.λ, o.lam.reshape(K, V).T, rtol=1e-9)
    np.testing.assert_allclose(g.Elnβ, o.Elnbeta.reshape(K, V).T, rtol=1e-9, atol=1e-12)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
    np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, K), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(g.θ, o.theta.reshape(D, K).T, rtol=1e-5)
    g2 = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
    o2 = oracle.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0)
    ll_g = mmm.fit(g2, maxiter=100, tol=1e-4, verbose=False)
    ll_o = o2.fit(maxiter=100, tol=1e-4)
    assert len(ll_g) == len(ll_o) and g2.converged == o2.converged
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    assert g2.elbo == pytest.approx(o2.elbo_value, rel=1e-7)


@pytest.mark.parametrize("D,K", [(10000, 10), (4000, 7), (12000, 12), (300, 4)])
def test_prologue_beside_the_previous_reduction_changes_no_bit(mmm, tuning, D, K):
    """Single-step build: Elntheta_{t+1} and exp(Elntheta_{t+1}) are formed by extra blocks of pass t's merged launch (same functions, same
    lanes) and the E-step kernel of pass t+1 starts at its term phase.  Against the build in which every pass forms its own
    (MMM_OFF_LDA_EARLY_PROLOGUE): every bit of ll, gamma, lambda, Elntheta, phi equal -- over passes enqueued in one call, over calls
    (the first pass of a call always forms its own), and when the stopping rule ends the fit in mid-chunk."""
    V = 96
    X, lam0 = np_ref.synth_lda(D, V, K, seed=77 + K)
    res = []
    for off in (False, True):
        tuning(disable=("lda_early_prologue",) if off else ())
        g = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
        assert g.geometry()["single_step"] == 1
        mmm.lib().mmm_lda_iterate(g._h, 5)
        g1 = g.γ.copy()
        mmm.lib().mmm_lda_iterate(g._h, 1); mmm.lib().mmm_lda_iterate(g._h, 7)
        st = (g1, g.γ.copy(), g.λ.copy(), g.Elnθ.copy(), g.phi_flat().copy())
        g2 = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
        ll = mmm.fit(g2, maxiter=200, tol=1e-4, verbose=False)
        g3 = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)          # a fit that ends by maxiter, its ELBO, then more passes on the same handle
        mmm.fit(g3, maxiter=6, tol=0.0, verbose=False)
        e3 = np.float64(g3.elbo)
        mmm.lib().mmm_lda_iterate(g3._h, 3)
        res.append(st + (np.asarray(ll), g2.γ.copy(), g2.λ.copy(), g2.Elnθ.copy(), np.float64(g2.elbo), e3, g3.γ.copy(), g3.Elnθ.copy(), g3.λ.copy()))
        tuning()
    assert len(res[0][5]) < 200          # the rule fired
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("D,V,K", [(70, 50, 16), (50, 96, 20), (40, 30, 32), (45, 96, 13)])
def test_wide_topic_counts_use_wider_lane_groups(mmm, oracle, tuning, D, V, K):
    """K >= 16 runs 32- or 64-lane document groups (K + 1 lanes are needed for the digamma step); K = 13 pads to KP = 16.
    (K > 24 goes to the wide-vocabulary kernels by default -- tests/test_lda_wide_gpu.py; here the LDS kernels are kept in play.)"""
    tuning(lda_build="sparse")
    X, g, o = _pair(mmm, oracle, D, V, K, seed=300 + K, empty=(2,))
    ll_g = mmm.fit(g, maxiter=13, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=13, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    _cmp_state(g, o, 1e-8)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-8)


def test_many_documents_per_wave_loop(mmm, oracle, tuning):
    """A tiny grid forces every wave through several document steps (the grid-stride path used for large corpora)."""
    tuning(grid_blocks=3)
    X, g, o = _pair(mmm, oracle, 500, 96, 10, seed=12, mean_n=500, empty=(7, 499))
    ll_g = mmm.fit(g, maxiter=12, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=12, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    _cmp_state(g, o, 1e-8)


def test_degenerate_shapes(mmm, oracle):
    """K = 1, a single document, a corpus whose documents are all empty but one, V larger than any used term."""
    for D, V, K, empty in [(5, 7, 1, ()), (1, 96, 10, ()), (6, 30, 4, (0, 1, 2, 4, 5)), (3, 200, 2, ())]:
        X, g, o = _pair(mmm, oracle, D, V, K, seed=900 + D + K, mean_n=50, empty=empty)
        ll_g = mmm.fit(g, maxiter=4, tol=0.0, verbose=False)
        ll_o = o.fit(maxiter=4, tol=0.0)
        np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
        _cmp_state(g, o, 1e-9)
        assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
        g.close()


@pytest.mark.parametrize("D,V,K,mean_n", [(500, 96, 10, 500), (300, 40, 7, 200), (260, 128, 4, 900), (70, 20, 12, 100), (333, 90, 5, 60),
                                           (200, 33, 15, 300)])
def test_dense_row_estep_matches_oracle(mmm, oracle, tuning, D, V, K, mean_n):
    """The dense-row E-step build (rows of counts, statistics in registers; taken by default for dense corpora too large for the single-step
    build) forced on small corpora of every slot count it has builds for, sparse rows and empty documents included."""
    tuning(lda_build="dense")
    X, g, o = _pair(mmm, oracle, D, V, K, seed=31 + D, mean_n=mean_n, empty=(3, D - 1))
    geo = g.geometry()
    assert geo["dense"] == 1 and geo["SL"] * 16 >= V and geo["single_step"] == 0
    ll_g = mmm.fit(g, maxiter=12, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=12, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    _cmp_state(g, o, 1e-8)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
    # several steps per wave and bitwise run-to-run
    tuning(lda_build="dense", grid_blocks=2)
    _, g2, _ = _pair(mmm, oracle, D, V, K, seed=31 + D, mean_n=mean_n, empty=(3, D - 1))
    _, g3, _ = _pair(mmm, oracle, D, V, K, seed=31 + D, mean_n=mean_n, empty=(3, D - 1))
    ll_2 = mmm.fit(g2, maxiter=12, tol=0.0, verbose=False)
    ll_3 = mmm.fit(g3, maxiter=12, tol=0.0, verbose=False)
    np.testing.assert_allclose(ll_2, ll_o, rtol=1e-9)
    assert np.array_equal(np.asarray(ll_2), np.asarray(ll_3)) and np.array_equal(g2.λ, g3.λ) and np.array_equal(g2.γ, g3.γ)
    _cmp_state(g2, o, 1e-8)


def _random_lda_shapes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        K = int(rng.integers(1, 25))
        V = int(rng.integers(5, 200))
        D = int(rng.integers(3, 700))
        mean_n = int(rng.integers(V // 2 + 1, 12 * V))
        out.append((D, V, K, mean_n))
    return out


@pytest.mark.parametrize("idx,shape", list(enumerate(_random_lda_shapes(16, 20261004))))
def test_random_shapes_every_estep_build_against_oracle(mmm, oracle, tuning, idx, shape):
    """Randomly drawn (D, V, K, document length): six fused passes through the default build of the shape, the grid-stride loop (a 3-block
    grid), the dense-row build (where the shape has one) and the wide-table path, each against the oracle."""
    D, V, K, mean_n = shape
    envs = [{}, {"grid_blocks": 3}, {"lda_build": "dense"}, {"lda_build": "wide"}]
    ll_o = None
    for env in envs:
        tuning(**env)
        X, g, o = _pair(mmm, oracle, D, V, K, seed=400 + idx, mean_n=mean_n, empty=(0,) if D > 4 else ())
        tuning()
        geo = g.geometry()
        if env.get("lda_build") == "wide":
            assert geo["wide"] == 1
        ll_g = mmm.fit(g, maxiter=6, tol=0.0, verbose=False)
        if ll_o is None:
            ll_o = o.fit(maxiter=6, tol=0.0); ref = o
        np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9, err_msg=str((env, geo)))
        np.testing.assert_allclose(g.λ, ref.lam.reshape(V, K, order="F"), rtol=1e-9, err_msg=str((env, geo)))
        np.testing.assert_allclose(g.γ, ref.gamma.reshape(D, K).T, rtol=1e-9, err_msg=str((env, geo)))
        assert g.elbo == pytest.approx(ref.elbo_value, rel=1e-9)
        g.close()


@pytest.mark.parametrize("env", [{}, {"lda_build": "dense"}, {"disable": ("lda_count_rows",)}, {"disable": ("lda_rows16",)}, {"disable": ("lda_padded_rows",)}])
def test_rows_of_counts_with_a_count_beyond_16_bits(mmm, oracle, tuning, env):
    """Dense corpora are also kept as rows of counts (16-bit where every count fits, else 32-bit) for the single-step E-step build, the ll
    blocks and the dense-row build; one count of 70,000 forces the 32-bit rows; every switch that selects another data path gives the
    oracle's fit."""
    tuning(**env)
    for big in (False, True):
        X, lam0 = np_ref.synth_lda(300, 96, 10, seed=21, mean_n=900)
        if big:
            X[5][0, 1] = 70000
        g = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
        o = oracle.LdaOracle(10, 0.1, 0.1, X, V=96, lambda0=lam0)
        np.testing.assert_allclose(mmm.fit(g, maxiter=10, tol=0.0, verbose=False), o.fit(maxiter=10, tol=0.0), rtol=1e-9)
        np.testing.assert_allclose(g.λ, o.lam.reshape(96, 10, order="F"), rtol=1e-9)
        np.testing.assert_allclose(g.γ, o.gamma.reshape(300, 10).T, rtol=1e-9)
        g.close()


def test_dense_row_build_is_not_taken_for_sparse_or_duplicated_rows(mmm, oracle, tuning):
    """Documents that list a term twice (the reference treats the rows separately) or a sparse corpus keep the CSR sweep."""
    tuning(lda_build="dense")
    X, lam0 = np_ref.synth_lda(40, 24, 5, seed=5, mean_n=100)
    X[7] = np.vstack([X[7], X[7][:1]])
    g = mmm.LDA(5, 0.1, 0.1, 24, X, λ0=lam0)
    assert g.geometry()["dense"] == 0
    o = oracle.LdaOracle(5, 0.1, 0.1, X, V=24, lambda0=lam0)
    np.testing.assert_allclose(mmm.fit(g, maxiter=5, tol=0.0, verbose=False), o.fit(maxiter=5, tol=0.0), rtol=1e-9)
    tuning()
    X, g, o = _pair(mmm, oracle, 200, 96, 10, seed=3, mean_n=3000)
    assert g.geometry()["dense"] == 0 and g.geometry()["single_step"] == 1       # small corpora: the single-step build


@pytest.mark.parametrize("dense", ["0", None])
def test_large_corpus_grid_stride_paths(mmm, oracle, tuning, dense):
    """50,000 documents: the grid-stride E-step builds (several steps per wave; the CSR sweep and, by default for this dense corpus, the
    dense-row build) and the ll blocks' loop (more document groups than ll blocks) against the oracle."""
    if dense is not None:
        tuning(lda_build="sparse")
    X, g, o = _pair(mmm, oracle, 50000, 96, 10, seed=77, mean_n=150, empty=(0, 49999))
    assert g.geometry()["dense"] == (0 if dense == "0" else 1)
    ll_g = mmm.fit(g, maxiter=4, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=4, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    np.testing.assert_allclose(g.λ, o.lam.reshape(96, 10, order="F"), rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)


@pytest.mark.parametrize("cap", [61, 64, 100, 30])
def test_merged_launch_never_exceeds_residency(mmm, oracle, cap, tuning):
    """The reduce + ll + M-step launch has blocks that wait for each other (16-byte cells).  The library launches it only with as
    many blocks as can be resident at once (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs; here lowered through
    mmm_tuning_opts.resident_cap): the ll blocks are cut to what fits (cap 61 / 64 / 100: 60 reduce blocks + 1 / 4 / 40 ll blocks for
    20,000 documents, which would take 313), or the split kernels run (cap 30 < 60 reduce blocks).  Either way: the oracle's
    results, no wait_timeout."""
    tuning(resident_cap=cap)
    X, g, o = _pair(mmm, oracle, 20000, 96, 10, seed=78, mean_n=120)
    ll_g = mmm.fit(g, maxiter=14, tol=0.0, verbose=False)        # raises MmmError on a device-side wait time-out
    ll_o = o.fit(maxiter=14, tol=0.0)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    np.testing.assert_allclose(g.λ, o.lam.reshape(96, 10, order="F"), rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)


def test_context_may_be_destroyed_before_its_models(mmm):
    """A garbage-collected host runs finalizers in no particular order (the Julia shim registers one per object): destroying the
    context first only marks it; the last model's destroy releases it."""
    X, lam0 = np_ref.synth_lda(50, 96, 10, seed=3, mean_n=300)
    ctx = mmm.Context(0)
    a = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0, ctx=ctx)
    b = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0, ctx=ctx)
    L = mmm.lib()
    assert L.mmm_ctx_destroy(ctx.h) == 1          # MMM_DEFERRED: two models alive (communicator / mailboxes are given up at once)
    ll = mmm.fit(a, maxiter=3, tol=0.0, verbose=False)   # stream and memory are still there
    assert np.all(np.isfinite(ll))
    ha, hb = a._h, b._h
    a._h = mmm._lib.C.c_void_p(); b._h = mmm._lib.C.c_void_p(); ctx.h = mmm._lib.C.c_void_p()     # the Python objects must not destroy again
    assert L.mmm_lda_destroy(ha) == 0
    assert L.mmm_lda_destroy(hb) == 0             # releases the context


def test_dense_row_build_with_a_never_observed_term_and_tiny_priors(mmm, tuning):
    """with eta = alpha = 1e-3 the normaliser sum_k a_k exp(Elnbeta_kv) of a term that no document contains underflows to 0;
    the dense-row build visits every slot of a row, so a slot without mass must contribute 0, not 0 x rcp(0) = NaN.  Checked against the
    CSR sweep (which never sees such a slot) on the same corpus."""
    X, lam0 = np_ref.synth_lda(400, 96, 10, seed=5, mean_n=600)
    dead = 37
    X = [x[x[:, 0] != dead + 1] for x in X]           # term ids are 1-based in X
    lam0 = np.ones_like(lam0)
    fits = {}
    for mode in ("1", "0"):
        tuning(lda_build="dense" if mode == "1" else "sparse")
        g = mmm.LDA(10, 1e-3, 1e-3, 96, X, λ0=lam0)
        assert g.geometry()["dense"] == int(mode)
        ll = mmm.fit(g, maxiter=60, tol=0.0, verbose=False)
        fits[mode] = (ll, np.array(g.γ), np.array(g.λ))
        g.close()
    for a, b in zip(fits["1"], fits["0"]):
        assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
        np.testing.assert_allclose(a, b, rtol=1e-9)
    assert np.all(np.abs(fits["1"][2][dead] - 1e-3) < 1e-12)      # lambda of the dead term = eta
