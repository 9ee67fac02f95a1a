#!/usr/bin/env python3
"""Wall time of whole `fit` calls (create excluded): LDA on the BASELINE shape, and the 560-document MMCTM of cfg 3.
Usage: python tools/bench_fit_wall.py  -> one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
mmm = mmm_pkg.load()
X, lam0 = np_ref.synth_lda(10000, 96, 10, seed=1, mean_n=400)
out = {}
best = 1e9
for rep in range(6):
    g = mmm.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
    g.ctx.synchronize()
    t0 = time.perf_counter(); ll = mmm.fit(g, maxiter=1000, tol=0.0, verbose=False); dt = time.perf_counter() - t0
    best = min(best, dt)
    out["lda_passes"] = len(ll); out["lda_fit_ms"] = best * 1e3; out["lda_us_per_pass_wall"] = best / len(ll) * 1e6
    g.close()
Xm, g0 = np_ref.synth_mm(560, [96, 48], [7, 7], seed=4, means=[3000, 60], empty_frac=0.1)
best = 1e9
for rep in range(5):
    c = mmm.MMCTM([7, 7], [0.1, 0.1], [96, 48], Xm, γ0=g0)
    c.ctx.synchronize()
    t0 = time.perf_counter(); ll = mmm.fit(c, maxiter=100, tol=0.0, verbose=False); dt = time.perf_counter() - t0
    best = min(best, dt)
    out["ctm_passes"] = len(ll); out["ctm_fit_ms"] = best * 1e3; out["ctm_us_per_pass_wall"] = best / len(ll) * 1e6
    c.close()
print(json.dumps(out))
