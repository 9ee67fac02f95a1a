#!/usr/bin/env python3
"""Differential fuzzing of one MMCTM pass (E-step for every document + M-step + ll) against the CPU oracle over random shapes, LDS and
wide tables (not part of the test-suite).  Usage: python tools/fuzz_ctm.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
from oracle import oracle
import test_ctm_gpu as T
mmm = mmm_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    M = int(rng.choice([1, 2, 3, 4]))
    big = bool(rng.integers(0, 4) == 0)       # a quarter of the cases: beyond 32 topics in a modality / beyond 64 coordinates (round 3)
    while True:
        K = [int(rng.choice([1, 2, 3, 5, 7, 10, 14, 17, 20, 32] + ([33, 40, 50, 64] if big else []))) for _ in range(M)]
        if sum(K) <= (256 if big else 64): break
    V = [int(rng.choice([2, 5, 16, 38, 48, 96, 200, 700])) for _ in range(M)]
    D = int(rng.choice([1, 3, 40, 130, 300]))
    means = [int(rng.choice([5, 60, 800, 4000])) for _ in range(M)]
    mode = str(rng.choice(["lds", "wide"]))
    mmm.default_context().set_tuning(ctm_build="wide" if mode == "wide" else "auto")
    try:
        X, g, o = T._pair(mmm, oracle, D, K, V, seed=int(rng.integers(1 << 30)), means=means, empty_frac=float(rng.choice([0.0, 0.2])))
        MK = sum(K)
        mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
        o.estep_range(0, D); o.update_mu(); so = o.update_Sigma(); o.update_gamma(); o.update_props(); o.update_phi()
        T._cmp_docs(g, o, D, MK, M, frac=0.95 if MK <= 64 or D < 20 else 0.8)
        np.testing.assert_allclose(g._get("gamma"), o.gamma, rtol=1e-4)
        np.testing.assert_allclose(g.μ, o.mu, rtol=1e-4, atol=1e-6)
        ll = np.zeros(M); nn = mmm._lib.C.c_int()
        mmm._lib.check(mmm.lib().mmm_ctm_ll_history(g._h, ll.ctypes.data, 1, mmm._lib.C.byref(nn)), g.ctx.h)
        np.testing.assert_allclose(ll, o.loglik(), rtol=1e-5)
        ok = True; msg = ""
        g.close()
    except Exception as e:      # noqa: BLE001
        ok = False; msg = repr(e).replace("\\n", " ")[:300]
    bad += not ok
    print("%s case %d mode=%s D=%d K=%s V=%s means=%s %s" % ("ok " if ok else "BAD", case, mode, D, K, V, means, msg))
print("failures: %d of %d" % (bad, n))
sys.exit(1 if bad else 0)
