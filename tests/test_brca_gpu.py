"""BASELINE configs 1 and 3 on the reference's shipped BRCA-EU count tables (tests/golden/*.tsv, data only):
config 1 = LDA K=7, alpha=eta=0.1 on the SNV table; config 3 = MMCTM K=[7,7], alpha=[0.1,0.1] on SNV+SV (16 documents
have an empty SV modality).  GPU backend vs the CPU oracle."""
import os

import numpy as np
import pytest

import np_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tables(mmm):
    terms1, samples, snv = mmm.read_counts_tsv(os.path.join(GOLD, "brca-eu_snv_counts.tsv"))
    terms2, samples2, sv = mmm.read_counts_tsv(os.path.join(GOLD, "brca-eu_sv_counts.tsv"))
    assert samples == samples2 and snv.shape == (96, 560) and sv.shape == (48, 560)
    return samples, {s: snv[:, i] for i, s in enumerate(samples)}, {s: sv[:, i] for i, s in enumerate(samples)}


def test_config1_lda_k7_brca_snv(mmm, oracle):
    samples, snv, _ = _tables(mmm)
    X = mmm.format_counts_lda(snv, samples)                      # utils.jl:9-18
    assert len(X) == 560 and sum(x.shape[0] for x in X) == 53559   # nnz = 53,559
    lam0 = np.random.default_rng(1).integers(1, 101, size=(96, 7)).astype(np.float64)
    g = mmm.LDA(7, 0.1, 0.1, X, λ0=lam0)
    o = oracle.LdaOracle(7, 0.1, 0.1, X, V=96, lambda0=lam0)
    ll_g = mmm.fit(g, maxiter=60, tol=1e-4, verbose=False)
    ll_o = o.fit(maxiter=60, tol=1e-4)
    assert len(ll_g) == len(ll_o) and g.converged == o.converged
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    np.testing.assert_allclose(g.phi_flat(), o.phi.reshape(-1, 7), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(g.θ, o.theta.reshape(560, 7).T, rtol=1e-5)
    np.testing.assert_allclose(g.β, o.beta.reshape(96, 7, order="F"), rtol=1e-5)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-5)


def _config3(mmm):
    samples, snv, sv = _tables(mmm)
    X = mmm.format_counts_mmctm([snv, sv], samples)              # utils.jl:24-36
    assert sum(1 for d in X if d[1].shape[0] == 0) == 16         # 16 documents without SVs
    rng = np.random.default_rng(2)
    g0 = [rng.integers(1, 101, size=(7, 96)).astype(np.float64), rng.integers(1, 101, size=(7, 48)).astype(np.float64)]
    return X, g0, mmm.MMCTM([7, 7], [0.1, 0.1], [96, 48], X, γ0=g0)


def _bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel(); b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    return int((a.view(np.int64) != b.view(np.int64)).sum())


def test_config3_mmctm_77_brca_snv_sv(mmm, oracle):
    """BASELINE config 3 against the order-matched oracle (oracle/mmm_twin.c, the launch geometry of this handle): twelve
    passes compared after every pass -- every document's two LD_MMA solves take the same number of objective evaluations, the
    state is identical in every bit -- then the whole fit at the north star's 1e-5 on ll, ELBO, theta, props, gamma."""
    X, g0, g = _config3(mmm)
    geo = g.geometry()
    o = oracle.CtmOracle([7, 7], [0.1, 0.1], X, V=[96, 48], gamma0=np.concatenate([x.ravel() for x in g0]), geometry=geo)
    n = 12
    for it in range(n):
        mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
        assert o.twin_pass(True) == 0
        st = g.solver_stats(per_doc=True)
        assert st["n_capped"] == 0
        assert np.array_equal(st["per_doc_nu"], o.nev_nu[:560]) and np.array_equal(st["per_doc_lambda"], o.nev_lambda[:560]), "pass %d" % (it + 1)
        for name, a, b in (("lambda", g.lam_matrix(), o.lam), ("nu", g.nu_matrix(), o.nu), ("mu", g.μ, o.mu),
                           ("invSigma", np.asarray(g.invΣ).ravel(order="F"), o.invSigma), ("gamma", g._get("gamma"), o.gamma)):
            assert _bits(a, b) == 0, "pass %d: %s differs in %d values" % (it + 1, name, _bits(a, b))
    # the whole fit through fit!
    X, g0, g = _config3(mmm)
    o = oracle.CtmOracle([7, 7], [0.1, 0.1], X, V=[96, 48], gamma0=np.concatenate([x.ravel() for x in g0]), geometry=geo)
    ll_g = mmm.fit(g, maxiter=n, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=n, tol=0.0)
    assert ll_g.shape == (n, 2)
    rtol = 1e-5      # north star
    np.testing.assert_allclose(ll_g, ll_o, rtol=rtol)
    assert g.elbo == pytest.approx(o.elbo_value, rel=rtol)
    np.testing.assert_allclose(g._get("theta"), o.theta, rtol=rtol, atol=1e-300)
    o.update_props()
    np.testing.assert_allclose(g._get("props"), o.props, rtol=rtol)
    np.testing.assert_allclose(g._get("gamma"), o.gamma, rtol=rtol)
    np.testing.assert_allclose(g._get("phi"), o.phi, rtol=rtol)
    obs = (np.abs(ll_g / ll_o - 1).max(), abs(g.elbo / o.elbo_value - 1))
    print("config 3, %d passes: ll rel err %.1e, ELBO rel err %.1e (bar 1e-5); lambda, nu, mu, invSigma, gamma bit-identical" % ((n,) + obs))
    assert max(obs) < 1e-9


def test_config3_against_index_order_oracle(mmm, oracle):
    """Config 3 against the index-order variant.  The first passes agree to ~1e-12; later ones drift at the 1e-5 level because
    LD_MMA's stopping tests are discontinuous -- exactly the drift the two CPU variants show against each other on this corpus
    (tests/golden/oracle_trajectories.json: fork_vs_index_order_ll_rel = 2e-13, 1e-13, 2e-11, 8e-11, 5e-8, 2e-6, 1e-5, 2e-5, ...)."""
    X, g0, g = _config3(mmm)
    o = oracle.CtmOracle([7, 7], [0.1, 0.1], X, V=[96, 48], gamma0=np.concatenate([x.ravel() for x in g0]))
    n = 12
    ll_g = mmm.fit(g, maxiter=n, tol=0.0, verbose=False)
    ll_o = o.fit(maxiter=n, tol=0.0)
    np.testing.assert_allclose(ll_g[:3], ll_o[:3], rtol=1e-9)
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-4)
    print("config 3 ll rel err per pass against the index-order oracle:", np.abs(ll_g / ll_o - 1).max(axis=1))
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-4)
    pe = np.abs(g._get("props") - o.props)
    assert np.median(pe) < 2e-4 and pe.max() < 5e-2


def test_config3_every_pass_against_index_order_oracle_from_the_same_state(mmm, oracle):
    """What the free-running comparison above cannot show because LD_MMA trajectories fork: the distance between the device and
    the LITERAL oracle (index-order sums, libm, NLopt's formulas as written) does not come from the kernels and does not grow with
    the pass number.  Thirty passes of config 3; before every pass the oracle is given the device's state, both take the pass, and the
    results are compared: zeta / theta / gamma to 1e-9, ll and the Gaussian parameters inside the north star's 1e-5, lambda and nu
    equal to 1e-7 for >= 95 % of the documents (the others stopped one evaluation apart: the solver's tolerance, 1e-4)."""
    X, g0, g = _config3(mmm)
    o = oracle.CtmOracle([7, 7], [0.1, 0.1], X, V=[96, 48], gamma0=np.concatenate([x.ravel() for x in g0]))
    D, MK, M = 560, 14, 2
    worst = dict(ll=0.0, mu=0.0, Sigma=0.0, gamma=0.0, theta=0.0, docs_1e9=1.0, doc_max=0.0)
    for it in range(30):
        if it:      # the oracle continues from the device's state
            o.lam[:] = g.lam_matrix().ravel(); o.nu[:] = g.nu_matrix().ravel()
            o.mu[:] = g.μ; o.Sigma[:] = np.asarray(g.Σ).ravel(order="F"); o.invSigma[:] = np.asarray(g.invΣ).ravel(order="F")
            o.gamma[:] = g._get("gamma"); o.Elnphi[:] = g._get("Elnphi")
        mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
        o.estep_range(0, D); o.update_mu(); assert o.update_Sigma() == 0; o.update_gamma(); o.update_props(); o.update_phi()
        np.testing.assert_allclose(g._get("zeta"), o.zeta, rtol=1e-9)
        np.testing.assert_allclose(g._get("theta"), o.theta, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(g._get("gamma"), o.gamma, rtol=1e-9)
        rows = np.maximum(np.abs(g.lam_matrix() - o.lam.reshape(D, MK)) / np.maximum(1.0, np.abs(o.lam.reshape(D, MK))),
                          np.abs(g.nu_matrix() - o.nu.reshape(D, MK)) / np.maximum(1.0, np.abs(o.nu.reshape(D, MK)))).max(axis=1)
        assert np.mean(rows < 1e-7) >= 0.95 and rows.max() < 2e-3, "pass %d: %.3f of the documents within 1e-7, worst %.2g" % (it + 1, np.mean(rows < 1e-7), rows.max())
        n = mmm._lib.C.c_int(); hist = np.zeros((it + 1) * M)
        mmm._lib.check(mmm.lib().mmm_ctm_ll_history(g._h, hist.ctypes.data, it + 1, mmm._lib.C.byref(n)), g.ctx.h)
        assert n.value == it + 1
        ll = hist.reshape(-1, M)[it]             # the pass's own ll (MMCTM.jl:476-479), as fit! records it
        e_ll = np.abs(ll / o.loglik() - 1).max()
        e_mu = np.abs(np.asarray(g.μ) - o.mu).max() / np.abs(o.mu).max()
        e_S = np.abs(np.asarray(g.Σ).ravel(order="F") - o.Sigma).max() / np.abs(o.Sigma).max()
        assert e_ll < 1e-5 and e_mu < 1e-5 and e_S < 1e-5, "pass %d: ll %.2g mu %.2g Sigma %.2g" % (it + 1, e_ll, e_mu, e_S)
        worst["ll"] = max(worst["ll"], e_ll); worst["mu"] = max(worst["mu"], e_mu); worst["Sigma"] = max(worst["Sigma"], e_S)
        worst["gamma"] = max(worst["gamma"], np.abs(g._get("gamma") / o.gamma - 1).max())
        worst["theta"] = max(worst["theta"], np.abs(g._get("theta") - o.theta).max())
        worst["docs_1e9"] = min(worst["docs_1e9"], float(np.mean(rows < 1e-9))); worst["doc_max"] = max(worst["doc_max"], float(rows.max()))
    e, _ = mmm.calculate_elbo(g, terms=True)
    assert e == pytest.approx(o.elbo()[0], rel=1e-5)
    print("config 3, 30 passes, each against the index-order oracle from the same state: worst pass ll %.1e, mu %.1e, Sigma %.1e, gamma %.1e, theta (abs) %.1e; "
          "documents with lambda and nu within 1e-9: >= %.1f %%, worst document %.1e; ELBO rel %.1e" % (
              worst["ll"], worst["mu"], worst["Sigma"], worst["gamma"], worst["theta"], 100 * worst["docs_1e9"], worst["doc_max"], abs(e / o.elbo()[0] - 1)))


def test_restart_driver_on_brca(mmm):
    """`fit_model` of scripts/run_mmctm.jl:163-182 on the shipped tables: 6 restarts in one batch, per-modality selection,
    seeded second-stage fit.  The batched stage 1 must pick what R separate fits pick, and stage 2 must start from the
    selected topics."""
    from multimodalmusig_jl_amd import restarts as rs
    samples, snv, sv = _tables(mmm)
    X = mmm.format_counts_mmctm([snv, sv], samples)
    K, alpha, V = [7, 7], [0.1, 0.1], [96, 48]
    seeds = [11, 12, 13, 14, 15, 16]
    kw = dict(maxiter=25, tol=1e-3)
    opt_gamma, opt_ll, all_ll = rs.fit_seed_models(X, K, alpha, V, seeds, **kw)
    # the same sweep in two chunks of three, and restart by restart
    og2, ol2, al2 = rs.fit_seed_models(X, K, alpha, V, seeds, batch_size=3, **kw)
    np.testing.assert_array_equal(all_ll, al2)
    np.testing.assert_array_equal(opt_ll, ol2)
    for m in range(2):
        np.testing.assert_array_equal(opt_gamma[m], og2[m])
    for i, s in enumerate(seeds[:2]):
        rng = np.random.default_rng(s)
        g0 = [rng.integers(1, 101, size=(K[m], V[m])).astype(np.float64) for m in range(2)]
        single = mmm.MMCTM(K, alpha, V, X, γ0=g0)
        mmm.fit(single, verbose=False, **kw)
        np.testing.assert_array_equal(single.ll, all_ll[i])
    assert np.all(opt_ll == all_ll.max(axis=0))
    model = rs.seed_and_fit_restart(X, K, alpha, V, opt_gamma, maxiter=15, tol=1e-5)
    # stage 2 starts from the selected topics: its log-likelihood is at least as good as the per-modality optimum
    # it was seeded with, up to the slack of mixing two models' topics
    assert np.all(np.isfinite(model.ll)) and np.isfinite(model.elbo)
    assert np.all(model.ll > all_ll.mean(axis=0) - 0.05)
    print("restart driver: stage-1 ll per restart\n%s\nbest per modality %s, stage-2 ll %s" % (all_ll, opt_ll, model.ll))


def test_against_committed_golden_trajectories(mmm, tuning):
    """The same two configurations against tests/golden/oracle_trajectories.json: a committed target that needs no oracle
    build on the GPU box (generated by tests/golden/make_trajectories.py from the CPU oracle).  The bits of a CTM fit follow the launch
    geometry (it fixes the association of the sums across documents), and the geometry follows the CU count: the handle is created with
    mmm_tuning_opts.geometry_cus = 256 -- the geometry the fixture was generated for -- so the comparison holds on any gfx950 device or
    partition mode, not only on a 256-CU one."""
    tuning(geometry_cus=256)
    import json
    traj = json.load(open(os.path.join(GOLD, "oracle_trajectories.json")))
    samples, snv, sv = _tables(mmm)
    t = traj["config1_lda_k7"]
    lam0 = np.random.default_rng(t["lambda0_seed"]).integers(1, 101, size=(96, 7)).astype(np.float64)
    g = mmm.LDA(7, 0.1, 0.1, mmm.format_counts_lda(snv, samples), λ0=lam0)
    ll = mmm.fit(g, maxiter=t["maxiter"], tol=t["tol"], verbose=False)
    assert len(ll) == len(t["ll"]) and g.converged == t["converged"]
    np.testing.assert_allclose(ll, t["ll"], rtol=1e-9)
    assert g.elbo == pytest.approx(t["elbo"], rel=1e-5)                     # the north-star tolerance
    assert g.λ.sum() == pytest.approx(t["lambda_sum"], rel=1e-12)           # = eta V K + N: every count accounted for
    np.testing.assert_allclose(g.θ[:, 0], t["theta_first_doc"], rtol=1e-5)
    t = traj["config3_mmctm_77"]
    rng = np.random.default_rng(t["gamma0_seed"])
    g0 = [rng.integers(1, 101, size=(7, 96)).astype(np.float64), rng.integers(1, 101, size=(7, 48)).astype(np.float64)]
    c = mmm.MMCTM([7, 7], [0.1, 0.1], [96, 48], mmm.format_counts_mmctm([snv, sv], samples), γ0=g0)
    llc = mmm.fit(c, maxiter=t["maxiter"], tol=0.0, verbose=False)
    np.testing.assert_allclose(llc[:3], np.asarray(t["ll"])[:3], rtol=1e-9)
    np.testing.assert_allclose(llc, t["ll"], rtol=1e-4)                      # index-order oracle: the fork of DESIGN §2
    assert c.elbo == pytest.approx(t["elbo"], rel=1e-4)
    assert c._get("gamma").sum() == pytest.approx(t["gamma_sum"], rel=1e-9)
    # the same fit by the order-matched oracle, committed: bit-identical state, so everything agrees to rounding of the
    # ll / ELBO sums (which are outside the feedback loop)
    t = traj["config3_mmctm_77_device_order"]
    geo = c.geometry()
    assert all(geo[k] == t["geometry"][k] for k in ("L", "grid_e", "waves_e", "grid_m")), geo      # what geometry_cus = 256 pins
    np.testing.assert_allclose(llc, t["ll"], rtol=1e-11)
    assert c.elbo == pytest.approx(t["elbo"], rel=1e-10)
    assert np.array_equal(c.μ, t["mu"]) and np.array_equal(np.diag(c.invΣ), t["invSigma_diag"])
    assert np.array_equal(c._get("gamma")[:96], t["gamma_first_topic"])
    assert np.array_equal(c.lam_matrix()[0], t["lambda_doc0"]) and np.array_equal(c.nu_matrix()[0], t["nu_doc0"])
    assert c.lam_matrix().sum() == pytest.approx(t["lambda_sum"], rel=1e-13) and c.nu_matrix().sum() == pytest.approx(t["nu_sum"], rel=1e-13)
    st = c.solver_stats(per_doc=True)
    assert np.array_equal(st["per_doc_nu"], t["nev_nu_last_pass"]) and np.array_equal(st["per_doc_lambda"], t["nev_lambda_last_pass"])


def test_two_handles_of_one_process_with_different_tuning(mmm, oracle, tuning):
    """mmm_tuning_opts are captured per handle at create time (they used to be environment variables read once per process): two LDA
    handles and two MMCTM handles of ONE process, created under different options, each give the oracle's fit -- and a pinned geometry
    (geometry_cus) gives the CTM's bits of a device of that size: 64 and 256 pretended CUs associate the cross-document sums differently."""
    X, lam0 = np_ref.synth_lda(900, 96, 10, seed=8, mean_n=700)
    tuning(lda_build="dense")
    a = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    tuning(lda_build="sparse", grid_blocks=5)
    b = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    tuning()
    assert a.geometry()["dense"] == 1 and b.geometry()["dense"] == 0 and b.geometry()["grid_e"] == 5
    o = oracle.LdaOracle(10, 0.1, 0.1, X, V=96, lambda0=lam0)
    ll_o = o.fit(maxiter=10, tol=0.0)
    for g in (b, a):                                        # (used in the other order than created)
        np.testing.assert_allclose(mmm.fit(g, maxiter=10, tol=0.0, verbose=False), ll_o, rtol=1e-9)
        assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
    Xm, g0 = np_ref.synth_mm(700, [96, 38], [6, 4], seed=12, means=[900, 120])
    fits = {}
    for cus in (64, 256):
        tuning(geometry_cus=cus, ctm_build="dense" if cus == 64 else "sparse")
        c = mmm.MMCTM([6, 4], [0.1, 0.1], [96, 38], Xm, γ0=g0)
        tuning()
        geo = c.geometry()
        oc = oracle.CtmOracle([6, 4], [0.1, 0.1], Xm, V=[96, 38], gamma0=np.concatenate([x.ravel() for x in g0]), geometry=geo)
        ll = mmm.fit(c, maxiter=5, tol=0.0, verbose=False)
        llo = oc.fit(maxiter=5, tol=0.0)
        assert np.array_equal(c.lam_matrix().ravel(), oc.lam) and np.array_equal(c._get("gamma"), oc.gamma), "CTM state differs from the order-matched oracle at geometry_cus = %d" % cus
        np.testing.assert_allclose(ll, llo, rtol=1e-10)
        fits[cus] = (geo, c._get("gamma").copy())
    assert fits[64][0]["tdense"] == 1 and fits[256][0]["tdense"] == 0
