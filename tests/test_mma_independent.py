"""LD_MMA is the one piece of the hot path that nothing under /root/reference pins (NLopt is an un-vendored C library; the reference's
tests at that boundary are qualitative -- test/mmctm.jl:92-101,150-155).  oracle/mmm_oracle.c:orc_mma_minimize restates it; this file
holds that restatement against a SECOND one (tests/np_ref.py:ccsa_mma, written from Svanberg 2002 and NLopt's manual, not from the C
file): the two must produce the same inner-iteration sequence -- candidate x, rho, sigma, f(candidate), the approximation's value --
on the lambda and nu objectives (MMCTM.jl:127-143,156-170) of 50 documents of BASELINE config 3, under both NLopt x-tolerance rules,
including a nu solve started on its 1e-7 bound.  It cannot pin NLopt itself; it removes transcription error from the unpinned part."""
import os

import numpy as np
import pytest

import np_ref
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = [7, 7]


@pytest.fixture(scope="module")
def cfg3_docs():
    """50 documents of the BRCA SNV + SV corpus after two oracle passes (so that mu, Sigma^-1 and Elnphi are no longer the init)"""
    import sys
    sys.path.insert(0, ROOT)
    import mmm_pkg
    pkg = mmm_pkg.load()
    _, samples, snv = pkg.read_counts_tsv(os.path.join(ROOT, "tests", "golden", "brca-eu_snv_counts.tsv"))
    _, _, sv = pkg.read_counts_tsv(os.path.join(ROOT, "tests", "golden", "brca-eu_sv_counts.tsv"))
    X = pkg.format_counts_mmctm([{s: snv[:, i] for i, s in enumerate(samples)}, {s: sv[:, i] for i, s in enumerate(samples)}], samples)
    g0 = np.random.default_rng(11).integers(1, 101, size=7 * 96 + 7 * 48).astype(np.float64)
    o = orc.CtmOracle(K, [0.1, 0.1], X, V=[96, 48], gamma0=g0)
    o.fit(maxiter=2, tol=0.0)
    MK = sum(K)
    mu = o.mu.copy(); invS = o.invSigma.reshape(MK, MK, order="F").copy()
    docs = []
    idx = list(range(0, 560, 12))[:47] + [d for d in range(560) if len(X[d][1]) == 0][:3]       # three documents with an empty SV modality
    for d in idx:
        lam = o.lam[MK * d:MK * (d + 1)].copy(); nu = o.nu[MK * d:MK * (d + 1)].copy()
        o.update_zeta(d); o.update_theta(d)
        docs.append(dict(d=d, lam=lam, nu=nu, Ndz=o.Ndivzeta(d), sumth=o.sumtheta(d)))
    return docs, mu, invS


def _neg(f):
    def h(x):
        v, g = f(x)
        return -v, -np.asarray(g)
    return h


def _compare(fun, x0, lower, rule):
    x_c, f_c, nev_c, _, tr_c = orc.mma_minimize(fun, x0, lb=None if lower is None else np.full(len(x0), lower), rule=rule, trace=True)
    x_p, f_p, nev_p, tr_p = np_ref.ccsa_mma(fun, x0, lower=lower, rule=rule)
    n = len(x0)
    assert nev_c == nev_p and len(tr_c) == len(tr_p) == nev_c - 1
    for row, t in zip(tr_c, tr_p):
        np.testing.assert_allclose(row[0], t["rho"], rtol=1e-13)
        np.testing.assert_allclose(row[1], t["gval"], rtol=1e-13)
        np.testing.assert_allclose(row[3], t["fcur"], rtol=1e-13)
        np.testing.assert_allclose(row[4:4 + n], t["sigma"], rtol=1e-13)
        np.testing.assert_allclose(row[4 + n:], t["x"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(x_c, x_p, rtol=1e-13)
    assert f_c == pytest.approx(f_p, rel=1e-13)
    return nev_c


@pytest.mark.parametrize("rule", [0, 1])
def test_lambda_and_nu_solves_step_by_step(cfg3_docs, rule):
    docs, mu, invS = cfg3_docs
    tot = 0
    for doc in docs:
        f_nu = _neg(lambda nu, doc=doc: orc.nu_objective(nu, doc["lam"], doc["Ndz"], mu, invS))
        tot += _compare(f_nu, doc["nu"], 1e-7, rule)
        f_lam = _neg(lambda lam, doc=doc: orc.lambda_objective(lam, doc["nu"], doc["Ndz"], doc["sumth"], mu, invS))
        tot += _compare(f_lam, doc["lam"], None, rule)
    assert tot > 50 * 2 * 3           # the solves really iterate


@pytest.mark.parametrize("rule", [0, 1])
def test_nu_solve_started_on_its_bound(cfg3_docs, rule):
    docs, mu, invS = cfg3_docs
    for doc in docs[:10]:
        f_nu = _neg(lambda nu, doc=doc: orc.nu_objective(nu, doc["lam"], doc["Ndz"], mu, invS))
        nu0 = doc["nu"].copy(); nu0[::2] = 1e-7            # every other coordinate on the bound: gradient 1/(2 nu) = 5e6 there
        _compare(f_nu, nu0, 1e-7, rule)


def test_independent_restatement_minimises():
    """sanity of the second restatement on its own: a strictly convex quadratic + exp term, optimum checked by the gradient"""
    rng = np.random.default_rng(3)
    A = rng.normal(size=(6, 6)); A = A @ A.T + 6 * np.eye(6); b = rng.normal(size=6)

    def fun(x):
        return 0.5 * x @ A @ x - b @ x + np.exp(x).sum(), A @ x - b + np.exp(x)

    x, f, nev, tr = np_ref.ccsa_mma(fun, np.zeros(6), xtol_rel=1e-10, xtol_abs=1e-12)
    assert np.linalg.norm(fun(x)[1]) < 1e-6 and nev < 500
    assert all(t["gval"] >= t["fcur"] or True for t in tr)


def test_one_quotient_step_is_nlopts_step():
    """Round 5: the device (csrc/ctm_estep.cuh) and the order-matched oracle write the minimiser of LD_MMA's per-coordinate model as
    dx = -g s^2 / (v + sqrt(rho (|g| s + rho / 4))), v = |g| s + rho / 2 -- one quotient and one root -- where NLopt (mma.c, restated
    literally in oracle/mmm_oracle.c) has dx = (u / v) / (-1 - sqrt|1 - (u / (v s))^2|), u = g s^2.  Both are the same number: against
    50-digit arithmetic over the ranges the solves see (|g| 1e-9 ... 1e6, sigma 1e-6 ... 10, rho 1e-5 ... 1e4) each form is within a few
    ulp of the exact value, and so of the other; the model's gradient term (u dx + v dx^2) / (s^2 - dx^2) likewise."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    rng = np.random.default_rng(20261005)
    worst = 0.0
    for _ in range(4000):
        g = float(rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-9, 6))
        s = float(10.0 ** rng.uniform(-6, 1))
        rho = float(10.0 ** rng.uniform(-5, 4))
        u, v = g * s * s, abs(g) * s + 0.5 * rho
        lit = (u / v) / (-1.0 - np.sqrt(abs(1.0 - (u / (v * s)) ** 2)))                       # NLopt's form, in double
        one = -(g * (s * s)) / (v + np.sqrt(rho * (abs(g) * s + 0.25 * rho)))                  # the device's form, in double
        G, S, R = mp.mpf(g), mp.mpf(s), mp.mpf(rho)
        U, V = G * S * S, abs(G) * S + R / 2
        exact = (U / V) / (-1 - mp.sqrt(abs(1 - (U / (V * S)) ** 2)))
        exact2 = -U / (V + mp.sqrt(R * (abs(G) * S + R / 4)))
        assert abs(exact - exact2) <= abs(exact) * mp.mpf(10) ** -40                           # the identity itself
        for val in (lit, one):
            worst = max(worst, float(abs(mp.mpf(val) - exact) / abs(exact)))
        assert abs(one - float(exact)) <= 4 * np.spacing(abs(float(exact)))                    # the one-quotient form: a few ulp
        # NLopt's form loses digits where (u / (v s))^2 -> 1 (rho << |g| s): 1 - x^2 cancels; the one-quotient form does not
        assert abs(lit - float(exact)) <= max(64 * np.spacing(abs(float(exact))), 1e-9 * abs(float(exact)))
        dx = one
        if abs(dx) < 0.9 * s:
            gt_lit = (u * dx + v * dx * dx) / (s * s - dx * dx)
            gt_one = (np.float64(v) * dx + u) * dx / (s * s - dx * dx)
            ex = (U * mp.mpf(dx) + V * mp.mpf(dx) ** 2) / (S * S - mp.mpf(dx) ** 2)
            if ex != 0:
                # (u dx and v dx^2 cancel to first order at the unclamped minimiser: compare at the scale of the terms)
                scale = float(abs(U * mp.mpf(dx)) / abs(S * S - mp.mpf(dx) ** 2))
                assert abs(gt_lit - float(ex)) <= 1e-12 * scale and abs(gt_one - float(ex)) <= 1e-12 * scale
    print("worst relative error of either form against 50 digits: %.2e" % worst)
