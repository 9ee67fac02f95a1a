#!/usr/bin/env python3
"""Differential fuzzing of the LDA / ILDA paths against the CPU oracle over random shapes (not part of the test-suite; run on a GPU
box when kernels change).  Usage: python tools/fuzz_lda.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
from oracle import oracle
mmm = mmm_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    K = int(rng.choice([1, 2, 3, 5, 7, 8, 10, 11, 12, 13, 16, 20, 24, 25, 32, 33, 47, 64, 65, 90, 128, 129, 200, 256]))
    V = int(rng.choice([1, 3, 16, 17, 32, 48, 78, 83, 96, 100, 128, 130, 200, 256, 257, 400, 1536]))
    D = int(rng.choice([1, 2, 5, 37, 64, 150, 401, 1500]))
    mean_n = int(rng.choice([5, 50, 500, 5000]))
    mode = rng.choice(["default", "wide", "split"])
    mmm.default_context().set_tuning(lda_build="wide" if mode == "wide" else "auto", disable=("lda_merged",) if mode == "split" else ())
    try:
        X, lam0 = np_ref.synth_lda(D, V, K, seed=int(rng.integers(1 << 30)), mean_n=mean_n)
    except ValueError:
        print('skip case %d (generator cannot make D=%d V=%d K=%d)' % (case, D, V, K)); continue
    for d in rng.choice(D, size=min(D - 1, int(rng.integers(0, 3))), replace=False) if D > 1 else []:
        X[int(d)] = np.zeros((0, 2), dtype=np.int64)
    try:
        g = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0)
        o = oracle.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0)
        it = int(rng.choice([1, 2, 5, 13]))
        ll_g = mmm.fit(g, maxiter=it, tol=0.0, verbose=False); ll_o = o.fit(maxiter=it, tol=0.0)
        # (V = 1: the ll is 0 up to rounding on both sides -- a relative error against 1e-16 means nothing, hence the floor on the denominator)
        # (V = 1: every document is its one term, log p = 0 up to rounding on both sides -- the figure is relative to max(|ll|, 1e-3), i.e. an
        #  ABSOLUTE 1e-12 where the ll itself is 1e-16)
        err = np.max(np.abs(ll_g - ll_o) / np.maximum(np.abs(ll_o), 1e-3)) if len(ll_g) == len(ll_o) else np.inf
        lam_err = np.max(np.abs(g.λ - o.lam.reshape(V, K, order="F")) / np.abs(o.lam.reshape(V, K, order="F")))
        el = abs(g.elbo - o.elbo_value) / max(abs(o.elbo_value), 1e-3)      # (V = K = 1: every ELBO term is exactly 0)
        ok = err < 1e-9 and lam_err < 1e-8 and el < 1e-8
        g.close()
    except Exception as e:      # noqa: BLE001
        ok = False; err = lam_err = el = float("nan"); print("EXC", repr(e)[:200])
    if not ok:
        bad += 1
    print("%s case %d mode=%s D=%d V=%d K=%d n=%d it=%d ll %.1e lam %.1e elbo %.1e" % ("ok " if ok else "BAD", case, mode, D, V, K, mean_n, it, err, lam_err, el))
print("failures: %d of %d" % (bad, n))
sys.exit(1 if bad else 0)
