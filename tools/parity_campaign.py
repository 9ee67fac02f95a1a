"""One-off randomized parity campaign on the GPU box (not part of the test-suite: tests/test_ctm_gpu.py and tests/test_lda_gpu.py keep 20 + 16
fixed draws of the same generators): N random CTM shapes bit for bit against the order-matched oracle, N random LDA shapes through every
E-step build against the oracle at 1e-9.  python tools/parity_campaign.py [N] [seed]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmm_pkg, np_ref
from oracle import oracle as orc
import test_ctm_gpu as T
import test_lda_gpu as TL

mmm = mmm_pkg.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
geos = {}
for idx, (D, K, V, means, feats) in enumerate(T._random_shapes(N, seed)):
    try:
        X, g, o = T._pair(mmm, orc, D, K, V, seed=5000 + idx, means=means, imm_features=feats, order="device")
        geo = g.geometry(); key = (geo["L"], geo["Ls"], geo["cpl"], geo["waves_e"]); geos[key] = geos.get(key, 0) + 1
        for it in range(3):
            mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
            o.twin_pass(True)
            T._same_state(g, o, D, sum(K))
            st = g.solver_stats(per_doc=True)
            assert np.array_equal(st["per_doc_nu"], o.nev_nu[:D]) and np.array_equal(st["per_doc_lambda"], o.nev_lambda[:D])
        g.close()
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("CTM case %d FAILED: D=%d K=%s V=%s imm=%s: %s" % (idx, D, K, V, feats is not None, str(e)[:300]))
print("CTM: %d shapes, %d failures; (L, Ls, cpl, waves_e) seen: %s" % (N, bad, sorted(geos.items())))
badl = 0
builds = {}
for idx, (D, V, K, mean_n) in enumerate(TL._random_lda_shapes(N, seed + 1)):
    try:
        ref = None
        for env in [{}, {"grid_blocks": 3}, {"lda_build": "dense"}, {"lda_build": "wide"}]:
            mmm.default_context().set_tuning(**env)
            X, lam0 = np_ref.synth_lda(D, V, K, seed=7000 + idx, mean_n=mean_n)
            g = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
            mmm.default_context().set_tuning()
            geo = g.geometry(); key = (geo["single_step"], geo["dense"], geo["wide"], geo["L"]); builds[key] = builds.get(key, 0) + 1
            ll_g = mmm.fit(g, maxiter=6, tol=0.0, verbose=False)
            if ref is None:
                ref = orc.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0); ll_o = ref.fit(maxiter=6, tol=0.0)
            np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
            np.testing.assert_allclose(g.λ, ref.lam.reshape(V, K, order="F"), rtol=1e-9)
            np.testing.assert_allclose(g.γ, ref.gamma.reshape(D, K).T, rtol=1e-9)
            g.close()
    except Exception as e:      # noqa: BLE001
        badl += 1
        mmm.default_context().set_tuning()
        print("LDA case %d FAILED: D=%d V=%d K=%d mean_n=%d env=%s: %s" % (idx, D, V, K, mean_n, env, str(e)[:300]))
print("LDA: %d shapes x 4 builds, %d failures; (single_step, dense, wide, L) seen: %s" % (N, badl, sorted(builds.items())))
# ---- restart batches: R replicas in one handle == R single fits, bit for bit (the replicas ride on grid.y of every kernel)
import test_ctm_batch_gpu as TB
badb = 0
nb = max(1, N // 5)
for idx, (D, K, V, means, feats) in enumerate(T._random_shapes(nb, seed + 2)):
    try:
        R = 3
        X, _ = np_ref.synth_mm(D, V, K, seed=8000 + idx, means=means, empty_frac=0.1)
        g0 = TB._inits(K, V, R, 8100 + idx, feats)
        batch = TB._make(mmm, K, V, X, g0, feats, restarts=R)
        hists = mmm.fit_restarts(batch, maxiter=8, tol=2e-3)
        for r in range(R):
            single = TB._make(mmm, K, V, X, g0[r], feats)
            h = mmm.fit(single, maxiter=8, tol=2e-3, verbose=False)
            assert np.array_equal(h, hists[r]), "ll history of restart %d" % r
            batch.select(r)
            for f in TB.FIELDS:
                assert np.array_equal(batch._get(f), single._get(f)), "restart %d field %s" % (r, f)
            single.close()
        batch.close()
    except Exception as e:      # noqa: BLE001
        badb += 1
        print("BATCH case %d FAILED: D=%d K=%s V=%s imm=%s: %s" % (idx, D, K, V, feats is not None, str(e)[:300]))
print("restart batches: %d shapes x 3 replicas, %d failures" % (nb, badb))
# ---- ILDA: random topic counts / corpora over the 96-term, 3-feature factorisation; default build and the dense-row build
import test_ilda_gpu as TI
badi = 0
rng = np.random.default_rng(seed + 3)
for idx in range(nb):
    D, K = int(rng.integers(5, 500)), int(rng.integers(1, 16))
    try:
        for env in [{}, {"lda_build": "dense"}, {"grid_blocks": 2}]:
            mmm.default_context().set_tuning(**env)
            X, g, o = TI._pair(mmm, orc, D, K, seed=9000 + idx)
            mmm.default_context().set_tuning()
            ll_g = mmm.fit(g, maxiter=8, tol=0.0, verbose=False)
            ll_o = o.fit(maxiter=8, tol=0.0)
            np.testing.assert_allclose(ll_g, ll_o, rtol=1e-9)
            for i in range(3):
                np.testing.assert_allclose(g.λ[i], o.mat(o.lam, i), rtol=1e-8)
            g.close()
    except Exception as e:      # noqa: BLE001
        badi += 1
        mmm.default_context().set_tuning()
        print("ILDA case %d FAILED: D=%d K=%d env=%s: %s" % (idx, D, K, env, str(e)[:300]))
print("ILDA: %d shapes x 3 builds, %d failures" % (nb, badi))
sys.exit(1 if bad or badl or badb or badi else 0)
