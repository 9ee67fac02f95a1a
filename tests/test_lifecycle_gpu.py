"""Handle lifecycle on the device: models and contexts created, used and destroyed many times leave the card's free memory where it was, and a
long fixed-length fit stays finite and monotone.  (The reference's objects are garbage-collected Julia arrays; a backend that holds device
buffers behind finalizers has to show that it gives them back -- SURVEY section 8b "Ownership".)"""
import gc

import numpy as np
import pytest

import np_ref

pytestmark = pytest.mark.gpu


def _free_bytes(mmm):
    """hipMemGetInfo of the HIP runtime the library itself is linked against (torch ships its own copy of libamdhip64: asking torch would put
    a second runtime into this process, and two of them abort at exit)"""
    import ctypes as C
    mmm.lib()                               # (loaded, if this is the first thing a test does)
    hip = C.CDLL(None)                      # the process's global symbols: the library is loaded RTLD_GLOBAL, its HIP runtime with it
    assert hip.hipDeviceSynchronize() == 0
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _cycle(mmm, i):
    X, lam0 = np_ref.synth_lda(300 + 7 * i, 96, 10, seed=100 + i)
    g = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    mmm.fit(g, maxiter=4, tol=0.0, verbose=False)
    g.phi_flat()
    g.close()
    Xm, g0 = np_ref.synth_mm(120 + 5 * i, [96, 48], [7, 7], seed=200 + i)
    c = mmm.MMCTM([7, 7], [0.1, 0.1], [96, 48], Xm, γ0=g0)
    mmm.fit(c, maxiter=3, tol=0.0, verbose=False)
    c.close()
    feats = [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]
    Xi, _ = np_ref.synth_mm(100, [96], [6], seed=300 + i)
    GM = 6 * int(feats[0].max(axis=0).sum())
    im = mmm.IMMCTM([6], [0.1], feats, Xi, γ0=np.random.default_rng(i).integers(1, 101, size=GM).astype(np.float64))
    mmm.fit(im, maxiter=3, tol=0.0, verbose=False)
    im.close()


def test_models_give_their_device_memory_back(mmm):
    _cycle(mmm, 0)                      # code objects, the context's own buffers, torch's pool: paid once
    gc.collect()
    base = _free_bytes(mmm)
    for i in range(1, 25):
        _cycle(mmm, i)
    gc.collect()
    lost = base - _free_bytes(mmm)
    assert lost < 16 << 20, "%.1f MB of device memory not returned after 24 create / fit / destroy cycles of LDA, MMCTM and IMMCTM handles" % (lost / 2**20)


def test_contexts_give_their_device_memory_back(mmm):
    X, lam0 = np_ref.synth_lda(200, 96, 10, seed=5)

    def once():
        ctx = mmm.Context(0)
        g = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0, ctx=ctx)
        mmm.fit(g, maxiter=3, tol=0.0, verbose=False)
        ll = float(np.asarray(g.ll)[-1]) if np.ndim(g.ll) else float(g.ll)
        g.close(); ctx.close()
        return ll
    first = once()
    gc.collect()
    base = _free_bytes(mmm)
    for _ in range(20):
        assert once() == first                         # a fresh context computes the same bits
    gc.collect()
    lost = base - _free_bytes(mmm)
    assert lost < 16 << 20, "%.1f MB of device memory not returned after 20 context create / destroy cycles" % (lost / 2**20)


def test_a_failed_create_leaks_nothing_and_leaves_the_context_usable(mmm):
    X, lam0 = np_ref.synth_lda(100, 96, 10, seed=6)
    mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0).close()       # the default context and the code objects exist (paid once: ~150 MB)
    gc.collect()
    base = _free_bytes(mmm)
    for _ in range(30):
        with pytest.raises(Exception):
            mmm.LDA(300, 0.1, 0.1, 96, X)               # K > 256: MMM_ERR_UNSUPPORTED
        with pytest.raises(Exception):
            mmm.MMCTM([70, 7], [0.1, 0.1], [96, 48], np_ref.synth_mm(20, [96, 48], [7, 7], seed=1)[0])      # K_m > 64
    lost = base - _free_bytes(mmm)
    assert lost < 4 << 20, "%.1f MB lost on failing creates" % (lost / 2**20)
    g = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    assert np.all(np.isfinite(mmm.fit(g, maxiter=3, tol=0.0, verbose=False)))


def test_a_thousand_passes_against_the_oracle(mmm, oracle):
    """fit!(maxiter = 1000, tol = 0) on a BRCA-sized corpus: the ring buffers, the ll history (grown on the way) and the pass counters hold
    over a long fit, and the 1000-pass ll history is the oracle's to 1e-9 (LDA.jl:198-224; the per-pass ll is the plug-in likelihood, not the
    bound: it creeps DOWN by 2e-10 per pass at the end of such a fit, in the oracle as on the device)."""
    X, lam0 = np_ref.synth_lda(560, 96, 7, seed=9)
    g = mmm.LDA(7, 0.1, 0.1, 96, X, λ0=lam0)
    o = oracle.LdaOracle(7, 0.1, 0.1, X, V=96, lambda0=lam0)
    ll = np.asarray(mmm.fit(g, maxiter=1000, tol=0.0, verbose=False))
    ll_o = np.asarray(o.fit(maxiter=1000, tol=0.0))
    assert ll.shape == (1000,) and np.all(np.isfinite(ll))
    np.testing.assert_allclose(ll, ll_o, rtol=1e-9)
    assert g.elbo == pytest.approx(o.elbo_value, rel=1e-9)
    np.testing.assert_allclose(g.λ, o.lam.reshape(7, 96).T, rtol=1e-7)
