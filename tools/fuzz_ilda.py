#!/usr/bin/env python3
"""Differential fuzzing of ILDA fits (merged and split launches, LDS and wide tables) against the CPU oracle over random feature
factorisations (not part of the test-suite).  Usage: python tools/fuzz_ilda.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
from oracle import oracle
mmm = mmm_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    I = int(rng.choice([1, 2, 3, 4]))
    J = [int(rng.choice([2, 3, 4, 6, 9])) for _ in range(I)]
    V = int(np.prod(J))
    if V > 600: J = J[:2]; I = 2; V = int(np.prod(J))
    # every combination of feature values is a term (mixed-radix digits), so every value of every feature occurs
    feats = np.zeros((V, I), dtype=np.int64)
    for v in range(V):
        r = v
        for i in range(I - 1, -1, -1):
            feats[v, i] = r % J[i] + 1; r //= J[i]
    K = int(rng.choice([1, 2, 5, 8, 10, 12, 16, 24]))
    D = int(rng.choice([1, 4, 50, 300]))
    mode = str(rng.choice(["default", "wide"]))
    mmm.default_context().set_tuning(lda_build="wide" if mode == "wide" else "auto")
    try:
        X, _ = np_ref.synth_lda(D, V, K, seed=int(rng.integers(1 << 30)), mean_n=int(rng.choice([20, 400, 3000])))
    except ValueError:
        print("skip", case); continue
    eta = [float(rng.choice([0.05, 0.1, 0.5])) for _ in range(I)]
    lam0 = [rng.integers(1, 101, size=(j, K)).astype(np.float64) for j in J]
    try:
        g = mmm.ILDA(K, 0.1, eta, feats, X, λ0=lam0)
        o = oracle.IldaOracle(K, 0.1, eta, feats, X, lambda0=np.concatenate([l.ravel(order="F") for l in lam0]))
        it = int(rng.choice([1, 3, 12]))
        ll_g = mmm.fit(g, maxiter=it, tol=0.0, verbose=False); ll_o = o.fit(maxiter=it, tol=0.0)
        err = np.max(np.abs(ll_g - ll_o) / np.maximum(np.abs(ll_o), 1e-300))
        le = max(np.max(np.abs(g.λ[i] - o.mat(o.lam, i)) / np.abs(o.mat(o.lam, i))) for i in range(I))
        el = abs(g.elbo - o.elbo_value) / max(abs(o.elbo_value), 1e-3)
        ok = err < 1e-9 and le < 1e-7 and el < 1e-8
        g.close()
    except Exception as e:      # noqa: BLE001
        ok = False; err = le = el = float("nan"); print("EXC", repr(e)[:200])
    bad += not ok
    print("%s case %d mode=%s D=%d J=%s V=%d K=%d it=%d ll %.1e lam %.1e elbo %.1e" % ("ok " if ok else "BAD", case, mode, D, J, V, K, it, err, le, el))
print("failures: %d of %d" % (bad, n))
sys.exit(1 if bad else 0)
