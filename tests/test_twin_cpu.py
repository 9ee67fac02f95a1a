"""The order-matched variant of the CPU oracle (oracle/mmm_twin.c: sums associated as the gfx950 kernels associate them,
exp/log/digamma from csrc/mmm_arith.h) against the index-order variant (oracle/mmm_oracle.c) and the reference's known
answers.  Both are restatements of the same reference lines; this file pins the second to the first and measures how far two
equally faithful evaluation orders drift apart over a whole fit (the discontinuous LD_MMA stopping tests amplify 1-ulp
differences) -- the bar the device-vs-oracle comparison has to be read against."""
import mpmath as mp
import numpy as np
import pytest

import np_ref


def _geom(D, MK, waves=8):
    L = 16 if MK <= 16 else (32 if MK <= 32 else 64)
    G = 64 // L
    return dict(L=L, waves_e=waves, grid_e=max(1, min((D + waves * G - 1) // (waves * G), 512)), grid_m=max(1, min((D + 31) // 32, 1024)))


def _pair(oracle, D, K, V, means, seed, geometry=None):
    X, g0 = np_ref.synth_mm(D, V, K, seed=seed, means=means, empty_frac=0.1)
    g0f = np.concatenate([x.ravel() for x in g0])
    a = oracle.CtmOracle(K, [0.1] * len(K), X, V=V, gamma0=g0f)
    b = oracle.CtmOracle(K, [0.1] * len(K), X, V=V, gamma0=g0f, geometry=geometry or _geom(D, sum(K)))
    return a, b


def test_shared_arithmetic_against_mpmath(oracle):
    """exp and log of csrc/mmm_arith.h: < 1 ulp; digamma: 1e-13 relative away from its zero."""
    L = oracle.lib()
    rng = np.random.default_rng(3)
    mp.mp.dps = 40

    def worst_ulp(fn, ref, xs):
        out = np.empty_like(xs); fn(xs.size, xs, out)
        return max(float(abs(mp.mpf(float(y)) - ref(mp.mpf(float(x)))) / mp.mpf(float(np.spacing(abs(float(ref(mp.mpf(float(x))))))))) for x, y in zip(xs, out))

    assert worst_ulp(L.orc_ar_exp_vec, mp.exp, np.concatenate([rng.uniform(-30, 30, 1500), rng.uniform(-700, 700, 500)])) < 1.0
    assert worst_ulp(L.orc_ar_log_vec, mp.log, np.concatenate([rng.uniform(1e-7, 10, 1500), 10.0 ** rng.uniform(-300, 300, 500)])) < 1.0
    xs = np.concatenate([rng.uniform(1e-3, 1.3, 300), rng.uniform(1.6, 30, 700), 10.0 ** rng.uniform(-7, 6, 500)])
    xs = xs[np.abs(xs - 1.4616321449683623) > 0.05]
    out = np.empty_like(xs); L.orc_ar_digamma_vec(xs.size, xs, out)
    rel = max(abs(float((mp.mpf(float(y)) - mp.digamma(mp.mpf(float(x)))) / mp.digamma(mp.mpf(float(x))))) for x, y in zip(xs, out))
    assert rel < 1e-13


def arr(x):
    return np.asarray(x, dtype=np.float64)


def test_twin_reference_kats(oracle, kats):
    """The reference's closed-form expectations through the order-matched variant, on the 2-document toy corpus: the two
    objectives with gradients (test/common.jl:79-97; test/mmctm.jl:135-148), update_ζ! (test/mmctm.jl:158-166), update_θ! /
    update_Elnϕ! (:168-209, :259-266), update_μ! / update_Σ! (:211-236), update_γ! (:238-257)."""
    C = oracle.C; L = oracle.lib()
    k = kats["lambda_objective"]; k2 = kats["nu_objective"]
    vals = np.zeros(2); gl = np.zeros(5); gn = np.zeros(5)
    L.orc_twin_objectives(5, arr(k["lambda"]), arr(k["nu"]), arr(k["Ndivzeta"]), arr(k["sumtheta"]), arr(k["mu"]), np.eye(5).ravel(), vals, gl, gn)
    assert vals[0] == pytest.approx(k["value"], rel=1e-13)
    np.testing.assert_allclose(gl, k["grad"], rtol=1e-13)
    L.orc_twin_objectives(5, arr(k2["lambda"]), arr(k2["nu"]), arr(k["Ndivzeta"]), arr(k["sumtheta"]), arr(k2["mu"]), np.eye(5).ravel(), vals, gl, gn)
    assert vals[1] == pytest.approx(k2["value"], rel=1e-13)
    np.testing.assert_allclose(gn, k2["grad"], rtol=1e-13)

    c = kats["corpora"]
    X = [[np.asarray(xm, dtype=np.int64) for xm in xd] for xd in c["X_mm"]]
    K = c["K_mm"]; MK = sum(K)

    def fresh():
        return oracle.CtmOracle(K, c["alpha_mm"], X, V=[4, 4], seed=5, geometry=_geom(2, MK))

    o = fresh()
    kt = kats["update_theta"]
    o.lam[:] = arr(kt["lambda"]).ravel()
    o.gamma[:] = np.concatenate([np.concatenate([arr(g) for g in gm]) for gm in kt["gamma"]])
    L.orc_twin_topics(C.byref(o.s), None)                        # update_Elnϕ! (+ the exp table)
    assert o.Elnphi[0] == pytest.approx(float(mp.digamma(1) - mp.digamma(11)), rel=1e-13)
    sG = o.twin_estep()                                          # ζ, θ, γ statistics (and the solves, which the reference does not pin)
    np.testing.assert_allclose(o.theta_dm(0, 0), arr(kt["theta_d1_m1"]), rtol=1e-12)
    np.testing.assert_allclose(o.theta_dm(1, 1), arr(kt["theta_d2_m2"]), rtol=1e-12)
    # the statistics are Σ_d n θ of exactly these θ (MMCTM.jl:230-240)
    ref = np.zeros_like(sG); D = 2
    for m_ in range(2):
        for d in range(D):
            th = o.theta_dm(d, m_)
            e0 = o.doc_ptr[m_ * (D + 1) + d]
            for w in range(th.shape[1]):
                v = o.term[e0 + w]; n = o.count[e0 + w]
                ref[o.goff[m_] + np.arange(K[m_]) * 4 + v] += n * th[:, w]
    np.testing.assert_allclose(sG, ref, rtol=1e-13)
    o = fresh()
    kz = kats["update_zeta"]
    o.lam[:] = arr(kz["lambda"]).ravel(); o.nu[:] = arr(kz["nu"]).ravel()
    o.twin_estep()
    np.testing.assert_allclose(o.zeta[:2], kz["zeta_doc1"], rtol=1e-14)
    o = fresh()
    km = kats["update_Sigma"]
    o.lam[:] = arr(km["lambda"]).ravel(); o.nu[:] = arr(km["nu"]).ravel()
    mom = np.zeros(2 * MK + MK * MK)
    L.orc_twin_moments(C.byref(o.s), mom)
    assert L.orc_twin_gauss(C.byref(o.s), mom, 1) == 0
    # update_μ! then update_Σ! with the NEW μ (fit! order, MMCTM.jl:467-469; the reference's update_Σ! test sets μ by hand)
    lam2, nu2 = arr(km["lambda"]), arr(km["nu"])
    np.testing.assert_allclose(o.mu, kats["update_mu"]["mu"] if np.array_equal(arr(kats["update_mu"]["lambda"]), lam2) else lam2.mean(axis=0), rtol=1e-14)
    dd = lam2 - lam2.mean(axis=0)
    S = (np.diag(nu2.sum(axis=0)) + dd.T @ dd) / 2.0                     # MMCTM.jl:204-210
    np.testing.assert_allclose(o.Sigma.reshape(5, 5), S, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(o.invSigma.reshape(5, 5), np.linalg.inv(S), rtol=1e-11, atol=1e-13)
    # update_γ! from statistics (test/mmctm.jl:238-257): γ = α + Σ n θ
    o = fresh()
    kg = kats["update_gamma"]
    stat = np.zeros(o.expE.size)
    for m_, key in ((0, "gamma_m1"), (1, "gamma_m2")):
        for kk in range(K[m_]):
            stat[o.goff[m_] + kk * 4:o.goff[m_] + (kk + 1) * 4] = arr(kg[key][kk]) - c["alpha_mm"][m_]
    L.orc_twin_topics(C.byref(o.s), stat.ctypes.data)
    for kk in range(2):
        np.testing.assert_allclose(o.gamma_mk(0, kk), kg["gamma_m1"][kk], rtol=1e-13)
    g = o.gamma_mk(0, 0)
    assert o.Elnphi[0] == pytest.approx(float(mp.digamma(mp.mpf(float(g[0]))) - mp.digamma(mp.mpf(float(g.sum())))), rel=1e-12)
    np.testing.assert_allclose(o.phi[:4], g / g.sum(), rtol=1e-14)


@pytest.mark.parametrize("D,K,V,means", [(120, [5, 4], [40, 24], [600, 80]), (90, [10, 10, 8], [96, 38, 32], [2000, 150, 100]),
                                         (60, [24, 17, 23], [30, 30, 30], [200, 200, 200])])
def test_one_pass_matches_index_order_variant(oracle, D, K, V, means):
    a, b = _pair(oracle, D, K, V, means, seed=21)
    np.testing.assert_allclose(b.Elnphi, a.Elnphi, rtol=1e-12, atol=1e-14)
    la = a.fit(maxiter=1, tol=0.0); lb = b.fit(maxiter=1, tol=0.0)
    same = (a.nev_nu == b.nev_nu) & (a.nev_lambda == b.nev_lambda)
    assert same.mean() > 0.9
    MK = sum(K)
    # documents whose solves took the same number of evaluations agree to rounding; the others moved by < xtol
    np.testing.assert_allclose(b.lam.reshape(D, MK)[same], a.lam.reshape(D, MK)[same], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(b.nu.reshape(D, MK)[same], a.nu.reshape(D, MK)[same], rtol=1e-8, atol=1e-10)
    assert np.abs(b.lam - a.lam).max() < 2e-3
    np.testing.assert_allclose(b.zeta, a.zeta, rtol=1e-13)
    np.testing.assert_allclose(b.theta, a.theta, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(b.gamma, a.gamma, rtol=1e-12)
    np.testing.assert_allclose(lb, la, rtol=1e-6)


def test_one_pass_identical_state_tight(oracle):
    """From identical (lambda, nu) the M-step of the two variants agrees to rounding: raw-moment Sigma and Gauss-Jordan
    inverse against the two-pass Sigma and LU inverse of the index-order variant."""
    a, b = _pair(oracle, 150, [7, 7], [96, 48], [3000, 60], seed=22)
    a.fit(maxiter=1, tol=0.0)
    b.lam[:] = 0.0; b.nu[:] = 1.0
    # run b's E-step, then overwrite its document state with a's and redo only the M-step pieces
    sG = b.twin_estep()
    b.lam[:] = a.lam; b.nu[:] = a.nu
    MK = 14
    mom = np.zeros(2 * MK + MK * MK)
    oracle.lib().orc_twin_moments(oracle.C.byref(b.s), mom)
    assert oracle.lib().orc_twin_gauss(oracle.C.byref(b.s), mom, 1) == 0
    np.testing.assert_allclose(b.mu, a.mu, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(b.Sigma, a.Sigma, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(b.invSigma, a.invSigma, rtol=1e-9, atol=1e-11)


def test_fork_between_two_faithful_orders(oracle, capsys):
    """How far do two equally literal evaluations of fit!(::MMCTM) drift apart?  Same algorithm, same inputs, sums associated
    differently and a different (also < 1 ulp) exp: the per-pass ll deviation grows from 1e-14 to the 1e-8 ... 1e-5 range
    within 15 passes.  (The device is compared with the order-matched variant bit for bit -- tests/test_ctm_gpu.py -- and with
    the index-order variant at this fork's magnitude.)"""
    a, b = _pair(oracle, 300, [7, 7], [96, 48], [3000, 60], seed=23)
    la = a.fit(maxiter=15, tol=0.0); lb = b.fit(maxiter=15, tol=0.0)
    dev = np.abs(la - lb).max(axis=1) / np.abs(la).max(axis=1)
    with capsys.disabled():
        print("\n  fork between the index-order and the device-order oracle, ll relative deviation per pass:\n  " + " ".join("%.1e" % x for x in dev))
        print("  ELBO relative deviation %.2e; gamma max rel %.2e" % (abs(a.elbo_value - b.elbo_value) / abs(a.elbo_value), (np.abs(a.gamma - b.gamma) / a.gamma).max()))
    assert dev[0] < 1e-12 and dev.max() < 1e-3
    assert abs(a.elbo_value - b.elbo_value) < 1e-4 * abs(a.elbo_value)


def test_geometry_changes_only_rounding(oracle):
    """The launch geometry enters only through the association of the sums across documents."""
    X, g0 = np_ref.synth_mm(100, [40, 24], [5, 4], seed=4, means=[600, 80], empty_frac=0.1)
    g0f = np.concatenate([x.ravel() for x in g0])
    res = []
    for geo in (dict(L=16, waves_e=8, grid_e=4, grid_m=4), dict(L=16, waves_e=4, grid_e=7, grid_m=1), dict(L=32, waves_e=8, grid_e=2, grid_m=3)):
        o = oracle.CtmOracle([5, 4], [0.1, 0.1], X, V=[40, 24], gamma0=g0f, geometry=geo)
        o.fit(maxiter=1, tol=0.0)
        res.append((o.gamma.copy(), o.mu.copy(), o.zeta.copy()))
    for g, m, z in res[1:]:
        np.testing.assert_allclose(g, res[0][0], rtol=1e-12)
        np.testing.assert_allclose(z, res[0][2], rtol=1e-13)
        np.testing.assert_allclose(m, res[0][1], rtol=1e-6, atol=1e-9)


def test_rows_of_counts_theta_phase_changes_only_rounding(oracle):
    """tdense = 1 (the device's k_ctm_theta_dense: 16 lanes per document, statistics per lane over the documents, one sweep per
    modality) is the slab version with its sums associated differently: one pass from the same state agrees to rounding -- zeta and
    theta (no sums across documents) to 1e-13, the gamma statistics to 1e-12."""
    K, V = [10, 10, 8], [96, 38, 32]
    X, g0 = np_ref.synth_mm(150, V, K, seed=8, means=[2000, 150, 100], empty_frac=0.1)
    g0f = np.concatenate([x.ravel() for x in g0])
    res = []
    for td in (0, 1):
        o = oracle.CtmOracle(K, [0.1] * 3, X, V=V, gamma0=g0f, geometry=dict(L=32, waves_e=8, grid_e=3, grid_m=2, Ls=32, cpl=1, tdense=td))
        assert o.twin_pass(True) == 0
        res.append((o.gamma.copy(), o.zeta.copy(), o.theta.copy(), o.lam.copy()))
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=1e-13)
    np.testing.assert_allclose(res[1][2], res[0][2], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-12)
    assert not np.array_equal(res[1][0], res[0][0])          # ... and it IS another association


def test_table_functions_at_every_table_boundary_and_over_the_range_the_solves_reach(oracle):
    """The order-matched oracle shares `ar_exp_tab` / `ar_log_tab` / `ar_digamma_pos_tab` (csrc/mmm_arith.h + the generated tables) with the
    kernels, so a wrong table entry or a reduction that breaks at an interval boundary would be common to both sides and invisible to the
    bit-identity tests.  Held against 40-digit mpmath here, independently of either side: at EVERY boundary of the tables (the rounding
    boundaries (k + 1/2) ln2/128 and the nodes k ln2/128 of the exp reduction; the mantissa boundaries 1 + j/128 of the log table at every
    binary exponent in range), each with its two neighbouring doubles, and on a dense random sample -- over the argument range the LD_MMA
    solves of BASELINE configs 3-5 really reach (tests/golden/table_argument_ranges.json, recorded by tests/golden/make_table_ranges.py through
    the oracle's debug hook at the configurations' full sizes) widened by a margin."""
    import json
    import os
    L = oracle.lib()
    mp.mp.dps = 40
    rg = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "table_argument_ranges.json")))
    cfgs = [v for k, v in rg.items() if k.startswith("config")]
    assert len(cfgs) == 3
    e_lo = min(c["exp_min"] for c in cfgs) - 2.0; e_hi = max(c["exp_max"] for c in cfgs) + 2.0
    l_lo = min(c["log_min"] for c in cfgs); l_hi = max(c["log_max"] for c in cfgs) * 2.0
    assert l_lo == 1e-7          # the lower bound of the nu solve (MMCTM.jl:157) is reached
    rng = np.random.default_rng(5)

    def run(fn, xs):
        xs = np.ascontiguousarray(xs, dtype=np.float64); out = np.empty_like(xs); fn(xs.size, xs, out); return xs, out

    def three(x):
        return [np.nextafter(x, -np.inf), x, np.nextafter(x, np.inf)]

    # ---- exp
    step = float(mp.log(2) / 128)
    pts = []
    for k in range(int(np.floor(e_lo / step)) - 1, int(np.ceil(e_hi / step)) + 2):
        pts += three(float(mp.mpf(k) * mp.log(2) / 128)) + three(float((mp.mpf(k) + mp.mpf(1) / 2) * mp.log(2) / 128))
    xs, ys = run(L.orc_ar_exptab_vec, np.concatenate([pts, rng.uniform(e_lo, e_hi, 20000), three(e_lo), three(e_hi)]))
    worst_e = 0.0
    for x, y in zip(xs, ys):
        r = mp.exp(mp.mpf(float(x)))
        worst_e = max(worst_e, float(abs(mp.mpf(float(y)) - r) / mp.mpf(float(np.spacing(float(r))))))
    assert worst_e < 0.52, worst_e
    # ---- log: absolute error (it enters a sum of O(10 .. 1e4)); the documented bound is 2.5e-15 for x <= 30
    pts = []
    for e in range(int(np.floor(np.log2(l_lo))) - 1, int(np.ceil(np.log2(l_hi))) + 2):
        for j in range(128):
            pts += three(float(np.ldexp(1.0 + j / 128.0, e)))
    pts = [p for p in pts if l_lo * 0.5 <= p <= l_hi]
    xs, ys = run(L.orc_ar_logtab_vec, np.concatenate([pts, 10.0 ** rng.uniform(np.log10(l_lo), np.log10(l_hi), 20000), three(1e-7), three(1.0), three(l_hi)]))
    worst_l = max(float(abs(mp.mpf(float(y)) - mp.log(mp.mpf(float(x))))) for x, y in zip(xs, ys))
    assert worst_l < 2.5e-15, worst_l
    # ---- digamma over the log table (LDA dense-row prologue; arguments: Dirichlet parameters 0.1 .. a document's total count)
    pts = []
    for e in range(2, 18):
        for j in range(128):
            for y in three(float(np.ldexp(1.0 + j / 128.0, e))):
                if y - 7.0 > 1e-3:
                    pts.append(y - 7.0)
    xs, ys = run(L.orc_ar_digammatab_vec, np.concatenate([pts, 10.0 ** rng.uniform(-1.5, 5, 5000)]))
    xs, ys = xs[np.abs(xs - 1.4616321449683623) > 0.05], ys[np.abs(xs - 1.4616321449683623) > 0.05]
    worst_d = max(float(abs(mp.mpf(float(y)) - mp.digamma(mp.mpf(float(x)))) / (abs(mp.digamma(mp.mpf(float(x)))) + 1)) for x, y in zip(xs, ys))
    assert worst_d < 4e-15, worst_d
    print("\n  table-driven functions against mpmath at every table boundary +- 1 ulp: exp %.3f ulp over [%.2f, %.2f], log abs %.2e over [%.1e, %.1f], "
          "digamma (|err| / (|psi| + 1)) %.2e" % (worst_e, e_lo, e_hi, worst_l, l_lo, l_hi, worst_d))
