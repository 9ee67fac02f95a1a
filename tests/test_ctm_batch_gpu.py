"""Restart batching (scripts/run_mmctm.jl:77-134): R models over one resident corpus, advanced together.

The contract is that replica r of a batch computes exactly what a model created from its gamma0 alone computes.  The batched
launches run the same kernels with the replica on grid.y, so the comparison against R separately created models is BITWISE;
parity with the reference algorithm then follows from the single-model tests (test_ctm_gpu.py), and is checked directly
against the CPU oracle here as well."""
import numpy as np
import pytest

import np_ref
from test_ctm_gpu import SNV3

pytestmark = pytest.mark.gpu


def _inits(K, V, R, seed, imm_features=None):
    rng = np.random.default_rng(seed)
    if imm_features is None:
        return [[rng.integers(1, 101, size=(K[m], V[m])).astype(np.float64) for m in range(len(K))] for _ in range(R)]
    GM = sum(K[m] * int(np.asarray(imm_features[m]).max(axis=0).sum()) for m in range(len(K)))
    return [rng.integers(1, 101, size=GM).astype(np.float64) for _ in range(R)]


def _make(mmm, K, V, X, g0, feats, restarts=None, **kw):
    alpha = [0.1] * len(K)
    if feats is None:
        return mmm.MMCTM(K, alpha, V, X, γ0=g0, restarts=restarts, **kw)
    return mmm.IMMCTM(K, alpha, feats, X, γ0=g0, restarts=restarts, **kw)


FIELDS = ["mu", "Sigma", "invSigma", "gamma", "Elnphi", "lambda", "nu", "zeta", "props", "theta"]


@pytest.mark.parametrize("case", ["mm", "imm", "imm10", "mm66", "mm40_40"])
def test_batched_fit_is_bitwise_the_single_model_fit(mmm, case):
    if case == "mm40_40":       # sum K = 80: the generic kernels of csrc/ctm_big.cuh with replicas on grid.y
        D, K, V, means, feats = 40, [40, 40], [60, 40], [900, 300], None
    elif case == "mm":
        D, K, V, means, feats = 70, [5, 4], [40, 24], [600, 80], None
    elif case == "imm10":       # sum K = 10: the several-coordinates-per-lane solve kernel (persistent waves, slot refill) with replicas on grid.y
        D, K, V, means, feats = 150, [10], [96], [1500], SNV3
    elif case == "mm66":        # sum K = 12: the packed solve groups
        D, K, V, means, feats = 70, [6, 6], [40, 24], [600, 80], None
    else:
        D, K, V, means, feats = 50, [6], [96], [1500], SNV3
    R = 4
    X, _ = np_ref.synth_mm(D, V, K, seed=12, means=means, empty_frac=0.1)
    g0 = _inits(K, V, R, 99, feats)
    batch = _make(mmm, K, V, X, g0, feats, restarts=R)
    assert batch.R == R
    hists = mmm.fit_restarts(batch, maxiter=30, tol=2e-3)
    iters = []
    for r in range(R):
        single = _make(mmm, K, V, X, g0[r], feats)
        h = mmm.fit(single, maxiter=30, tol=2e-3, verbose=False)
        iters.append(len(h))
        assert len(h) == len(hists[r]) == batch.restart_iters[r]
        assert np.array_equal(h, hists[r]), "restart %d: ll history differs from the single-model fit" % r
        assert single.converged == bool(batch.restart_converged[r])
        assert single.elbo == batch.restart_elbo[r]
        batch.select(r)
        for f in FIELDS:
            assert np.array_equal(batch._get(f), single._get(f)), "restart %d field %s" % (r, f)
        np.testing.assert_array_equal(batch.restart_ll[r], single.ll)
        single.close()
    # the point of the test: the restarts stop at different passes, and a stopped replica is left untouched afterwards
    assert len(set(iters)) > 1 or case == "mm40_40", "choose a case where the restarts stop at different passes (got %s)" % iters
    assert mmm.pick_optimal_modality_models(batch) == [int(i) for i in np.argmax(batch.restart_ll, axis=0)]


def test_batch_against_oracle(mmm, oracle):
    D, K, V = 48, [7, 7], [96, 48]
    R = 3
    X, _ = np_ref.synth_mm(D, V, K, seed=3, means=[2500, 60], empty_frac=0.15)
    g0 = _inits(K, V, R, 5)
    batch = _make(mmm, K, V, X, g0, None, restarts=R)
    hists = mmm.fit_restarts(batch, maxiter=12, tol=1e-12)
    MK = sum(K)
    for r in range(R):
        o = oracle.CtmOracle(K, [0.1, 0.1], X, V=V, gamma0=np.concatenate([x.ravel() for x in g0[r]]))
        ll_o = o.fit(maxiter=12, tol=1e-12)
        np.testing.assert_allclose(hists[r], ll_o, rtol=1e-5)         # the north-star tolerance
        assert batch.restart_elbo[r] == pytest.approx(o.elbo_value, rel=1e-5)
        batch.select(r)
        # per-document lambda after 12 whole passes: the MMA x-tolerance (1e-4) plus the stopping flips fed back through the
        # M-step (test_ctm_gpu.py docstring) -- the pass-level checks at 1e-7 are in test_ctm_gpu.py
        lam_o = o.lam.reshape(D, MK)
        err = (np.abs(batch.lam_matrix() - lam_o) / np.maximum(1.0, np.abs(lam_o))).max(axis=1)
        assert np.median(err) < 1e-3 and err.max() < 5e-2
        # mu is the mean of 48 lambdas that agree to the solver's tolerance (above): a document whose solve stopped one evaluation apart
        # moves it by that document's difference / 48 (round 5, one-quotient LD_MMA step: one entry at 7.7e-4 -- the forks fall elsewhere)
        np.testing.assert_allclose(batch.μ, o.mu, rtol=5e-3, atol=1e-3)


def test_stage_api_on_a_selected_replica(mmm):
    """The per-function API acts on the selected replica only and leaves the others untouched."""
    D, K, V = 40, [4, 3], [30, 20]
    R = 3
    X, _ = np_ref.synth_mm(D, V, K, seed=8, means=[400, 90], empty_frac=0.1)
    g0 = _inits(K, V, R, 21)
    batch = _make(mmm, K, V, X, g0, None, restarts=R)
    single = _make(mmm, K, V, X, g0[1], None)
    before = {r: {f: batch.select(r)._get(f) for f in FIELDS} for r in (0, 2)}
    batch.select(1)
    for m in (batch, single):
        mmm.fitdoc(m)
        mmm.update_μ(m); mmm.update_Σ(m); mmm.update_γ(m); mmm.update_props(m)
    # the update_γ! stage kernel accumulates with f64 atomics (order not fixed): gamma and what follows from it to 1e-12,
    # everything else bitwise
    np.testing.assert_allclose(mmm.calculate_loglikelihoods(batch), mmm.calculate_loglikelihoods(single), rtol=1e-12)
    assert mmm.calculate_elbo(batch) == pytest.approx(mmm.calculate_elbo(single), rel=1e-12)
    for f in FIELDS:
        if f in ("gamma", "Elnphi"):
            np.testing.assert_allclose(batch._get(f), single._get(f), rtol=1e-12, atol=1e-13, err_msg=f)
        else:
            assert np.array_equal(batch._get(f), single._get(f)), f
    for r in (0, 2):
        batch.select(r)
        for f in FIELDS:
            assert np.array_equal(batch._get(f), before[r][f]), "replica %d field %s changed" % (r, f)
    # single-model iterate / fit on one replica of a batch, then theta of another replica is rebuilt on selection
    batch.select(2)
    s2 = _make(mmm, K, V, X, g0[2], None)
    h_b = mmm.fit(batch, maxiter=5, tol=1e-12, verbose=False)
    h_s = mmm.fit(s2, maxiter=5, tol=1e-12, verbose=False)
    assert np.array_equal(h_b, h_s)
    assert np.array_equal(batch._get("theta"), s2._get("theta"))
    batch.select(1)
    assert np.array_equal(batch._get("theta"), single._get("theta"))
    batch.select(0)
    np.testing.assert_array_equal(batch._get("theta"), before[0]["theta"])         # still the constructor's 1/K


def test_batch_argument_checks(mmm):
    X, g0 = np_ref.synth_mm(10, [12, 8], [3, 2], seed=1, means=[50, 20])
    with pytest.raises(ValueError):
        mmm.MMCTM([3, 2], [0.1, 0.1], [12, 8], X, γ0=[g0], restarts=2)
    b = mmm.MMCTM([3, 2], [0.1, 0.1], [12, 8], X, restarts=2, seed=4)
    with pytest.raises(mmm.MmmError):
        b.select(2)
    b.select(1)
    mmm._lib.check(mmm.lib().mmm_ctm_iterate(b._h, 1, 1), b.ctx.h, "iterate")
    with pytest.raises(mmm.MmmError, match="different histories"):
        mmm.fit_restarts(b, maxiter=3)
