"""`fit!` as its callers use it (round 5): the seeded stage-2 fit of scripts/run_mmctm.jl:113-134 (construct, assign γ[m] / Elnϕ[m] / ϕ[m],
`fit!(tol=1e-5)`), the keyword `updateΣ=false` (MMCTM.jl:457-470), both NLopt x-tolerance rules over whole fits to convergence, the hyper-
parameters as mutable fields, and the error convention of SURVEY section 8b: non-finite values and capped solves are COUNTED, not raised
(the reference ignores NLopt's return code, MMCTM.jl:141,168)."""
import numpy as np
import pytest

import np_ref
from test_brca_gpu import _tables
from test_ctm_gpu import _fit_case, _pair, _same_bits, _same_state

pytestmark = pytest.mark.gpu


def _brca(mmm):
    samples, snv, sv = _tables(mmm)
    return mmm.format_counts_mmctm([snv, sv], samples)


def test_seeded_fit_by_field_assignment_is_the_reference_flow(mmm):
    """scripts/run_mmctm.jl:120-131 literally: `model = MMCTM(K, α, V, counts)`; `model.γ[m] = deepcopy(opt_models[m].γ[m])`, likewise Elnϕ[m]
    and ϕ[m]; `fit!(model, maxiter, tol=1e-5)`.  The fit must start from the assigned topics: bit for bit the fit of
    `restarts.seed_and_fit_restart` (which constructs from γ₀ = γ_opt), and not the fit from the constructor's random γ."""
    from multimodalmusig_jl_amd import restarts as rs
    X = _brca(mmm)
    K, V, alpha = [7, 7], [96, 48], [0.1, 0.1]
    seeds = [11, 12, 13, 14, 15, 16]
    g0 = []
    for s in seeds:
        rng = np.random.default_rng(s)
        g0.append([rng.integers(1, 101, size=(K[m], V[m])).astype(np.float64) for m in range(2)])
    batch = mmm.MMCTM(K, alpha, V, X, γ0=g0, restarts=len(seeds))
    mmm.fit_restarts(batch, maxiter=25, tol=1e-4)
    opt = mmm.pick_optimal_modality_models(batch)                      # run_mmctm.jl:86-97
    won = []
    for m in range(2):
        batch.select(int(opt[m]))
        # (explicit attributes: Python NFKC-normalises identifiers, `getattr(model, "Elnϕ")` with U+03D5 in a string would not find the property)
        won.append({"γ": [np.array(batch.γ[m][k]) for k in range(K[m])], "Elnϕ": [np.array(batch.Elnϕ[m][k]) for k in range(K[m])],
                    "ϕ": [np.array(batch.ϕ[m][k]) for k in range(K[m])]})
    batch.close()
    ref = rs.seed_and_fit_restart(X, K, alpha, V, [np.stack(won[m]["γ"]) for m in range(2)], maxiter=40, tol=1e-5)

    model = mmm.MMCTM(K, alpha, V, X, seed=123)                        # :122 -- random γ
    unseeded_gamma0 = model._get("gamma").copy()
    for m in range(2):                                                 # :125-129
        model.γ[m] = [x.copy() for x in won[m]["γ"]]
        model.Elnϕ[m] = [x.copy() for x in won[m]["Elnϕ"]]
        model.ϕ[m] = [x.copy() for x in won[m]["ϕ"]]
    assert not np.array_equal(model._get("gamma"), unseeded_gamma0)
    ll = mmm.fit(model, maxiter=40, tol=1e-5, verbose=False)           # :131
    for f in ("gamma", "Elnphi", "phi", "lambda", "nu", "mu", "Sigma", "invSigma", "props"):
        _same_bits(model._get(f), ref._get(f), f)
    assert model.elbo == ref.elbo and np.array_equal(model.ll, ref.ll) and model.converged == ref.converged

    plain = mmm.MMCTM(K, alpha, V, X, seed=123)
    llp = mmm.fit(plain, maxiter=40, tol=1e-5, verbose=False)
    assert not np.array_equal(plain._get("gamma"), model._get("gamma")), "the seeded fit equals the unseeded one: the assignments were ignored"
    print("seeded stage 2: %d passes, ll %s; unseeded from the same constructor: %d passes, ll %s" % (len(ll), model.ll, len(llp), plain.ll))
    for h in (ref, model, plain):
        h.close()


@pytest.mark.parametrize("case", ["cfg3_shape", "mm"])
def test_update_sigma_false_whole_fit(mmm, oracle, case):
    """`fit!(model; updateΣ=false)` (MMCTM.jl:468-470): Σ and Σ⁻¹ keep their constructor values (I) through the whole fit.  Against the
    order-matched oracle: same passes, state and evaluation counts identical; against the index-order oracle the usual 1e-4 of free-running
    CTM fits."""
    kw = _fit_case(case)
    D, MK = kw["D"], sum(kw["K"])
    X, g, o = _pair(mmm, oracle, order="device", **kw)
    ll_g = mmm.fit(g, maxiter=30, tol=1e-4, verbose=False, updateΣ=False)
    ll_o = o.fit(maxiter=30, tol=1e-4, update_sigma=False)
    assert len(ll_g) == len(ll_o) and g.converged == o.converged
    _same_state(g, o, D, MK)
    assert np.array_equal(np.asarray(g.Σ), np.eye(MK)) and np.array_equal(np.asarray(g.invΣ), np.eye(MK))
    st = g.solver_stats(per_doc=True)
    assert np.array_equal(st["per_doc_nu"], o.nev_nu[:D]) and np.array_equal(st["per_doc_lambda"], o.nev_lambda[:D])
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
    # ... and it is not the default fit
    X2, g2, _ = _pair(mmm, oracle, order="device", **kw)
    mmm.fit(g2, maxiter=30, tol=1e-4, verbose=False)
    assert not np.array_equal(np.asarray(g2.Σ), np.eye(MK))
    # the literal (index-order) restatement, free-running
    _, g3, oi = _pair(mmm, oracle, order="index", **kw)
    ll3 = mmm.fit(g3, maxiter=30, tol=1e-4, verbose=False, updateΣ=False)
    lli = oi.fit(maxiter=30, tol=1e-4, update_sigma=False)
    n = min(len(ll3), len(lli))
    assert abs(len(ll3) - len(lli)) <= 2
    np.testing.assert_allclose(ll3[:n], lli[:n], rtol=1e-4)
    assert g3.elbo == pytest.approx(oi.elbo_value, rel=1e-4)


def test_update_sigma_false_in_a_restart_batch(mmm):
    """a replica of a batch fitted with updateΣ=false is bitwise the single fit with updateΣ=false"""
    kw = _fit_case("mm")
    X, g0 = np_ref.synth_mm(kw["D"], kw["V"], kw["K"], seed=kw["seed"], means=kw["means"], empty_frac=0.15)
    rng = np.random.default_rng(5)
    inits = [[rng.integers(1, 101, size=(kw["K"][m], kw["V"][m])).astype(np.float64) for m in range(2)] for _ in range(3)]
    batch = mmm.MMCTM(kw["K"], [0.1, 0.1], kw["V"], X, γ0=inits, restarts=3)
    hists = mmm.fit_restarts(batch, maxiter=20, tol=1e-4, updateΣ=False)
    single = mmm.MMCTM(kw["K"], [0.1, 0.1], kw["V"], X, γ0=inits[1])
    h = mmm.fit(single, maxiter=20, tol=1e-4, verbose=False, updateΣ=False)
    batch.select(1)
    assert np.array_equal(h, hists[1])
    for f in ("gamma", "lambda", "nu", "mu", "Sigma", "invSigma"):
        _same_bits(batch._get(f), single._get(f), f)
    assert np.array_equal(single._get("Sigma"), np.eye(sum(kw["K"])).ravel())


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("case", ["cfg3_shape", "imm"])
def test_both_xtol_rules_over_a_whole_fit_to_convergence(mmm, oracle, case, rule):
    """Project.toml:15 admits NLopt 2.5 ... 2.7, whose x-tolerance tests differ (`xtol_rule`): a whole `fit!(tol = 1e-4)` to convergence under
    either rule -- same number of passes as the order-matched oracle, state and per-document evaluation counts identical."""
    kw = _fit_case(case)
    D, MK = kw["D"], sum(kw["K"])
    X, g, o = _pair(mmm, oracle, order="device", rule=rule, **kw)
    ll_g = mmm.fit(g, maxiter=60, tol=1e-4, verbose=False)
    ll_o = o.fit(maxiter=60, tol=1e-4)
    assert len(ll_g) == len(ll_o) and g.converged and o.converged, (len(ll_g), len(ll_o), g.converged, o.converged)
    _same_state(g, o, D, MK)
    st = g.solver_stats(per_doc=True)
    assert st["n_capped"] == 0 and g.events() == {"n_capped": 0, "n_nonfinite": 0, "n_nonfinite_ll": 0}
    assert np.array_equal(st["per_doc_nu"], o.nev_nu[:D]) and np.array_equal(st["per_doc_lambda"], o.nev_lambda[:D])
    np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
    print("%s, xtol_rule %d: converged after %d passes, %d + %d evaluations in the last pass" % (case, rule, len(ll_g), st["n_eval_nu"], st["n_eval_lambda"]))


@pytest.mark.parametrize("case", ["mm", "imm10", "cfg4_shape", "mm40_40"])
def test_nonfinite_values_are_counted_not_raised(mmm, case):
    """the reference ignores NLopt's return code and never checks for NaN -- it carries on.  So does the library, but the
    events can be counted: a NaN in one document's λ makes that document's two solves meet a non-finite objective (and run into the cap),
    the pass still returns 0, the other documents are solved as always; through μ the NaN then reaches every log-likelihood."""
    if case == "mm":
        D, K, V, means, feats = 70, [5, 4], [40, 24], [600, 80], None
    elif case == "imm10":           # k_ctm_solve_cpl<10, ...>
        from test_ctm_gpu import SNV3
        D, K, V, means, feats = 150, [10], [96], [1500], SNV3
    elif case == "cfg4_shape":      # k_ctm_solve_cpl<28, 16, ...>
        D, K, V, means, feats = 200, [10, 10, 8], [96, 38, 32], [2000, 150, 100], None
    else:                           # sum K = 80: k_ctm_solve_big
        D, K, V, means, feats = 40, [40, 40], [60, 40], [900, 300], None
    X, g0 = np_ref.synth_mm(D, V, K, seed=3, means=means, empty_frac=0.0)
    alpha = [0.1] * len(K)
    cap = 400       # (above what any healthy solve of these shapes needs in its third pass)
    mk = (lambda: mmm.MMCTM(K, alpha, V, X, γ0=g0, max_eval=cap)) if feats is None else (lambda: mmm.IMMCTM(K, alpha, feats, X, seed=4, max_eval=cap))
    g, clean = mk(), mk()
    check = mmm._lib.check
    for h in (g, clean):
        check(mmm.lib().mmm_ctm_iterate(h._h, 2, 1), h.ctx.h, "iterate")
    assert g.events() == {"n_capped": 0, "n_nonfinite": 0, "n_nonfinite_ll": 0}
    bad = 7
    lam = g.λ[bad].copy(); lam[1] = np.nan
    g.λ[bad] = lam
    mmm.update_ζ(g); mmm.update_ζ(clean)
    mmm.update_ν(g); mmm.update_ν(clean)             # every document's ν solve, one launch (no exception)
    ev = g.events()
    assert ev["n_nonfinite"] == 1 and ev["n_capped"] == 1, ev
    mmm.update_λ(g); mmm.update_λ(clean)
    ev = g.events()
    assert ev["n_nonfinite"] == 2 and ev["n_capped"] == 2, ev      # the document's ν solve (kept from the launch before) and its λ solve
    st = g.solver_stats(per_doc=True)
    assert st["n_capped"] == 2 and st["per_doc_nu"][bad] == -cap and st["per_doc_lambda"][bad] == -cap      # counts come back without the flag bit
    # every other document is solved exactly as in the clean model
    keep = np.arange(D) != bad
    assert np.array_equal(g.lam_matrix()[keep], clean.lam_matrix()[keep]) and np.array_equal(g.nu_matrix()[keep], clean.nu_matrix()[keep])
    assert np.array_equal(st["per_doc_lambda"][keep], clean.solver_stats(per_doc=True)["per_doc_lambda"][keep])
    # carry on: the NaN reaches μ and Σ⁻¹, then every document's solves and the log-likelihoods -- still status 0 (what the numbers are worth
    # after that is the caller's business, as upstream), now countable
    ll = mmm.fit(g, maxiter=3, tol=1e-4, verbose=False)
    ev = g.events()
    assert not g.converged and np.isnan(ll).any() and ev["n_nonfinite_ll"] == int((~np.isfinite(ll)).sum()) and ev["n_nonfinite"] == 2 * D, ev
    assert clean.events() == {"n_capped": 0, "n_nonfinite": 0, "n_nonfinite_ll": 0}


def test_lda_nonfinite_loglikelihood_is_counted_and_hyperparameters_are_fields(mmm, oracle):
    X, lam0 = np_ref.synth_lda(300, 96, 10, seed=8, mean_n=1500)
    g = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    mmm.fit(g, maxiter=12, tol=1e-4, verbose=False)
    assert g.events() == {"n_capped": 0, "n_nonfinite": 0, "n_nonfinite_ll": 0}
    # `model.α = 0.5; model.η = 0.2; fit!(model)`: the reference reads the fields at every update (LDA.jl:83,101)
    g.α, g.η = 0.5, 0.2
    ll = mmm.fit(g, maxiter=15, tol=0.0, verbose=False)
    o = oracle.LdaOracle(10, 0.1, 0.1, X, V=96, lambda0=lam0)
    o.fit(maxiter=12, tol=1e-4)
    o.alpha, o.eta = 0.5, 0.2
    llo = []
    for _ in range(15):             # (LdaOracle.fit is constructor + fit!: the continuation by stages)
        o.update_gamma(); o.update_phi(); o.update_lambda(); o.update_beta(); o.update_theta()
        llo.append(o.loglik())
    np.testing.assert_allclose(ll, llo, rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo()[0], rel=1e-9)
    E = g.Elnβ.copy(); E[5, 3] = np.nan
    g.Elnβ = E
    ll = mmm.fit(g, maxiter=14, tol=1e-4, verbose=False)
    assert len(ll) == 14 and not g.converged and np.isnan(ll).all()
    assert g.events()["n_nonfinite_ll"] == 14


def test_tuning_rejects_what_this_build_does_not_know(mmm):
    """mmm_ctx_set_tuning: unknown `disable` bits, non-zero reserved fields and lane counts without a build are MMM_ERR_ARG -- a caller built
    against a newer header must not have its choices dropped in silence (advisor, round 4)."""
    import ctypes as C
    ctx = mmm.default_context()
    lib = mmm.lib()

    def rc_of(**kw):
        t = mmm._lib.TuningOpts()
        for k, v in kw.items():
            if k == "reserved0":
                t.reserved[0] = v
            else:
                setattr(t, k, v)
        return lib.mmm_ctx_set_tuning(ctx.h, C.byref(t))
    try:
        assert rc_of() == 0 and rc_of(disable=1 << 3) == 0 and rc_of(solve_lanes=8) == 0 and rc_of(solve_waves=2) == 0
        assert rc_of(disable=1 << 20) == -1 and b"unknown bits" in lib.mmm_last_error(ctx.h)
        assert rc_of(reserved0=7) == -1 and b"reserved" in lib.mmm_last_error(ctx.h)
        assert rc_of(solve_lanes=3) == -1 and rc_of(solve_waves=9) == -1 and rc_of(lda_build=9) == -1
        assert ctx.get_tuning().solve_waves == 2          # a rejected call leaves the last good options in place
    finally:
        ctx.set_tuning()


def test_per_document_update_leaves_the_other_documents_and_their_counters_alone(mmm):
    """`update_ν!(model, d)` / `update_λ!(model, d)` (MMCTM.jl:127-170): only document d changes -- its ν / λ and ITS evaluation counter; the
    other documents keep values and counters of the last whole-corpus stage (advisor, round 4)."""
    kw = _fit_case("mm")
    X, g0 = np_ref.synth_mm(kw["D"], kw["V"], kw["K"], seed=kw["seed"], means=kw["means"], empty_frac=0.1)
    g = mmm.MMCTM(kw["K"], [0.1, 0.1], kw["V"], X, γ0=g0)
    mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 2, 1), g.ctx.h, "iterate")
    mmm.update_ζ(g); mmm.update_ν(g)
    before = g.solver_stats(per_doc=True)
    nu0, lam0 = g.nu_matrix().copy(), g.lam_matrix().copy()
    d = 11
    g.ν[d] = g.ν[d] * 3.0                       # move document d away from its optimum: its solve has work to do
    mmm.update_ν(g, d)
    after = g.solver_stats(per_doc=True)
    keep = np.arange(kw["D"]) != d
    assert np.array_equal(after["per_doc_nu"][keep], before["per_doc_nu"][keep]) and np.array_equal(after["per_doc_lambda"], before["per_doc_lambda"])
    assert np.array_equal(g.nu_matrix()[keep], nu0[keep]) and np.array_equal(g.lam_matrix(), lam0)
    np.testing.assert_allclose(g.nu_matrix()[d], nu0[d], rtol=2e-3)      # ... and document d is solved again (to the solver's tolerance)
    assert after["per_doc_nu"][d] > 1
