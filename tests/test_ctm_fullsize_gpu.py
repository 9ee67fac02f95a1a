"""BASELINE configs 4 and 5 at their FULL sizes on one GPU (50,000 documents x (96,38,32) terms, K = [10,10,8]; 100,000 documents x 96
terms, IMMCTM K = [10] with the SNV feature factorisation): the multi-tile moment sums, the reduction of hundreds of per-block
partials and the grid-stride solve phase, which the small parity cases never reach.

* against the order-matched oracle (oracle/mmm_twin.c) at full size: two whole passes, the state identical in every bit
  (lambda, nu, zeta of all documents; mu, Sigma^-1, gamma) and every document's LD_MMA evaluation counts equal;
* size-independent properties after six passes: props sum to 1 per modality, topic mass conservation
  sum(gamma[m]) = K_m V_m alpha + sum_d N_dm, Sigma symmetric positive definite with Sigma Sigma^-1 = I, ll finite and increasing,
  a second model reproduces the ll history bit for bit;
* a 200-document slice continued from the device's state: one more pass of those documents alone by the oracle gives the device's
  lambda / nu / zeta for them (the E-step of a document depends on the other documents only through the globals).
Reference: fitdoc! MMCTM.jl:450-455, IMMCTM.jl:430-435; fit! MMCTM.jl:457-494, IMMCTM.jl:437-466."""
import time

import numpy as np
import pytest

import np_ref

pytestmark = pytest.mark.gpu
SNV3 = [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]


def _bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel(); b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    return int((a.view(np.int64) != b.view(np.int64)).sum())


def _make(mmm, oracle, cfg, D, with_oracle=True):
    if cfg == 4:
        K, V, feats = [10, 10, 8], [96, 38, 32], None
    else:
        K, V, feats = [10], [96], SNV3
    X, g0 = np_ref.synth_mm(D, V, K, seed=20261003 + cfg)
    alpha = [0.1] * len(K)
    if feats is None:
        g = mmm.MMCTM(K, alpha, V, X, γ0=g0)
        init = np.concatenate([x.ravel() for x in g0])
    else:
        GM = sum(K[i] * int(f.max(axis=0).sum()) for i, f in enumerate(feats))
        init = np.random.default_rng(1).integers(1, 101, size=GM).astype(np.float64)
        g = mmm.IMMCTM(K, alpha, feats, X, γ0=init)
    o = None
    if with_oracle:
        o = oracle.CtmOracle(K, alpha, X, V=V if feats is None else None, features=feats, gamma0=init, geometry=g.geometry())
    return X, K, V, feats, init, g, o


@pytest.mark.parametrize("cfg,D", [(4, 50000), (5, 100000)])
def test_full_size_two_passes_bit_identical_to_oracle(mmm, oracle, cfg, D):
    t0 = time.time()
    X, K, V, feats, init, g, o = _make(mmm, oracle, cfg, D)
    MK = sum(K)
    geo = g.geometry()
    assert geo["grid_m"] > 64 and geo["grid_e"] > 64          # hundreds of partials: the strided 64-8-8 folds are exercised
    for it in range(2):
        mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
        assert o.twin_pass(True) == 0
        st = g.solver_stats(per_doc=True)
        assert st["n_capped"] == 0
        assert np.array_equal(st["per_doc_nu"], o.nev_nu[:D]) and np.array_equal(st["per_doc_lambda"], o.nev_lambda[:D])
        for name, a, b in (("lambda", g.lam_matrix(), o.lam), ("nu", g.nu_matrix(), o.nu), ("zeta", g._get("zeta"), o.zeta), ("mu", g.μ, o.mu),
                           ("Sigma", np.asarray(g.Σ).ravel(order="F"), o.Sigma), ("invSigma", np.asarray(g.invΣ).ravel(order="F"), o.invSigma),
                           ("gamma", g._get("gamma"), o.gamma), ("Elnphi", g._get("Elnphi"), o.Elnphi)):
            assert _bits(a, b) == 0, "cfg %d pass %d: %s differs in %d values" % (cfg, it + 1, name, _bits(a, b))
    np.testing.assert_allclose(mmm.calculate_loglikelihoods(g), o.loglik() if feats is not None else (o.update_props(), o.update_phi(), o.loglik())[2], rtol=1e-10)
    print("cfg %d, %d documents: 2 passes bit-identical to the oracle, geometry %s (%.0f s)" % (cfg, D, geo, time.time() - t0))


@pytest.mark.parametrize("cfg,D", [(4, 50000), (5, 100000)])
def test_full_size_properties_and_slice(mmm, oracle, cfg, D):
    X, K, V, feats, init, g, _ = _make(mmm, oracle, cfg, D, with_oracle=False)
    MK, M = sum(K), len(K)
    check = mmm._lib.check
    ll = mmm.fit(g, maxiter=5, tol=0.0, verbose=False)
    assert ll.shape == (5, M) and np.all(np.isfinite(ll)) and np.all(np.diff(ll, axis=0) > 0)
    # state S after 5 passes, then one more pass
    lam_S, nu_S = g.lam_matrix().copy(), g.nu_matrix().copy()
    mu_S, iS_S, gam_S = np.asarray(g.μ).copy(), np.asarray(g.invΣ).ravel(order="F").copy(), g._get("gamma").copy()
    check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
    assert g.solver_stats()["n_capped"] == 0
    # ---- properties
    N = np.array([[x[m][:, 1].sum() if len(x[m]) else 0 for m in range(M)] for x in X], dtype=np.float64)
    gam = g._get("gamma")
    if feats is None:
        props = g._get("props").reshape(D, MK)
        off = 0
        for m in range(M):
            np.testing.assert_allclose(props[:, off:off + K[m]].sum(axis=1), 1.0, rtol=1e-13)
            blk = gam[sum(K[i] * V[i] for i in range(m)):sum(K[i] * V[i] for i in range(m + 1))]
            assert blk.sum() == pytest.approx(K[m] * V[m] * 0.1 + N[:, m].sum(), rel=1e-11)          # sum_k theta = 1: every count lands in gamma
            off += K[m]
    else:
        # IMMCTM: every feature i of a topic receives every count once (IMMCTM.jl:209-221)
        SJ, I = 14, 3
        assert gam.sum() == pytest.approx(K[0] * SJ * 0.1 + I * N[:, 0].sum(), rel=1e-11)
    S = np.asarray(g.Σ); iS = np.asarray(g.invΣ)
    np.testing.assert_allclose(S, S.T, rtol=1e-12, atol=1e-14)
    assert np.linalg.eigvalsh(0.5 * (S + S.T)).min() > 0
    np.testing.assert_allclose(S @ iS, np.eye(MK), atol=1e-9)
    # ---- run-to-run: a second model gives the same bits
    X2, _, _, _, _, g2, _ = _make(mmm, oracle, cfg, D, with_oracle=False)
    ll2 = mmm.fit(g2, maxiter=5, tol=0.0, verbose=False)
    assert np.array_equal(ll, ll2)
    # ---- a 200-document slice continued by the oracle from the device's state
    n = 200
    d0 = D // 3
    o = oracle.CtmOracle(K, [0.1] * M, X[d0:d0 + n], V=V if feats is None else None, features=feats, gamma0=gam_S,
                         geometry=dict(g.geometry(), grid_e=1, waves_e=1, grid_m=1))
    oracle.lib().orc_twin_topics(oracle.C.byref(o.s), None)          # Elnphi / exp table from the device's gamma
    o.mu[:] = mu_S; o.invSigma[:] = iS_S
    o.lam[:] = lam_S[d0:d0 + n].ravel(); o.nu[:] = nu_S[d0:d0 + n].ravel()
    o.twin_estep_fused()           # (the device's fused pass: theta phase over rows of counts at these sizes; per-document results do not depend on the grid)
    assert _bits(g.lam_matrix()[d0:d0 + n], o.lam) == 0 and _bits(g.nu_matrix()[d0:d0 + n], o.nu) == 0
    assert _bits(g._get("zeta").reshape(D, M)[d0:d0 + n], o.zeta) == 0
    th = g._get("theta")
    assert np.all(np.isfinite(th)) and th.min() >= 0.0


@pytest.mark.parametrize("cfg,D", [(4, 50000), (5, 100000)])
def test_full_size_two_passes_against_the_literal_oracle_from_the_device_state(mmm, oracle, cfg, D):
    """The bit identity above is held against the order-matched oracle, which compiles the product's own `mmm_arith.h` and function tables: a
    table or association error that both sides share could hide behind it.  So the LITERAL restatement (oracle/mmm_oracle.c: index-order
    sums, libm exp / log, NLopt's formulas as written -- nothing shared with the kernels) takes two consecutive passes of configs 4 and 5 at
    their full sizes too, each from the device's state (LD_MMA trajectories fork between any two faithful evaluations, see
    test_brca_gpu.py::test_config3_every_pass_...): zeta / theta / gamma to 1e-9, the pass's log-likelihoods to 1e-8, mu / Sigma inside the
    north star's 1e-5, lambda and nu within 1e-7 for >= 95 % of the documents (the others stopped one evaluation apart: xtol = 1e-4)."""
    t0 = time.time()
    X, K, V, feats, init, g, _ = _make(mmm, oracle, cfg, D, with_oracle=False)
    MK, M = sum(K), len(K)
    o = oracle.CtmOracle(K, [0.1] * M, X, V=V if feats is None else None, features=feats, gamma0=init)      # geometry=None: the index-order variant
    check = mmm._lib.check
    check(mmm.lib().mmm_ctm_iterate(g._h, 3, 1), g.ctx.h, "iterate")          # away from the constructor state
    worst = {}
    for it in range(2):
        o.lam[:] = g.lam_matrix().ravel(); o.nu[:] = g.nu_matrix().ravel()
        o.mu[:] = g.μ; o.Sigma[:] = np.asarray(g.Σ).ravel(order="F"); o.invSigma[:] = np.asarray(g.invΣ).ravel(order="F")
        o.gamma[:] = g._get("gamma"); o.Elnphi[:] = g._get("Elnphi")
        check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
        o.estep_range(0, D); o.update_mu(); assert o.update_Sigma() == 0; o.update_gamma()
        if feats is None:
            o.update_props(); o.update_phi()
        np.testing.assert_allclose(g._get("zeta"), o.zeta, rtol=1e-9)
        np.testing.assert_allclose(g._get("theta"), o.theta, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(g._get("gamma"), o.gamma, rtol=1e-9)
        rows = np.maximum(np.abs(g.lam_matrix() - o.lam.reshape(D, MK)) / np.maximum(1.0, np.abs(o.lam.reshape(D, MK))),
                          np.abs(g.nu_matrix() - o.nu.reshape(D, MK)) / np.maximum(1.0, np.abs(o.nu.reshape(D, MK)))).max(axis=1)
        assert np.mean(rows < 1e-7) >= 0.95 and rows.max() < 2e-3, "pass %d: %.4f of the documents within 1e-7, worst %.2g" % (it + 1, np.mean(rows < 1e-7), rows.max())
        n = mmm._lib.C.c_int(); hist = np.zeros(M)
        check(mmm.lib().mmm_ctm_ll_history(g._h, hist.ctypes.data, 1, mmm._lib.C.byref(n)), g.ctx.h)
        e_ll = float(np.abs(hist / o.loglik() - 1).max())
        e_mu = float(np.abs(np.asarray(g.μ) - o.mu).max() / np.abs(o.mu).max())
        e_S = float(np.abs(np.asarray(g.Σ).ravel(order="F") - o.Sigma).max() / np.abs(o.Sigma).max())
        assert e_ll < 1e-8 and e_mu < 1e-5 and e_S < 1e-5, "pass %d: ll %.2g mu %.2g Sigma %.2g" % (it + 1, e_ll, e_mu, e_S)
        for k_, v_ in (("ll", e_ll), ("mu", e_mu), ("Sigma", e_S), ("docs_not_within_1e-9", float(np.mean(rows >= 1e-9))), ("worst_doc", float(rows.max()))):
            worst[k_] = max(worst.get(k_, 0.0), v_)
    print("cfg %d, %d documents, two passes against the index-order oracle from the device's state: %s (%.0f s)" % (cfg, D, worst, time.time() - t0))
