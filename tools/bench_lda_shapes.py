#!/usr/bin/env python3
"""LDA iteration time for arbitrary shapes (bench.py is fixed to BASELINE configs[1]): the LDS path vs the wide-vocabulary path.
Usage: python tools/bench_lda_shapes.py D V K mean_n [--wide] [--steps N]   -> one JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("D", type=int); ap.add_argument("V", type=int); ap.add_argument("K", type=int); ap.add_argument("mean_n", type=int)
ap.add_argument("--wide", action="store_true"); ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--ilda", action="store_true", help="ILDA over the SNV factorisation (V must be 96): I = 3, J = [6, 4, 4]")
a = ap.parse_args()
import mmm_pkg, np_ref
mmm = mmm_pkg.load()
if a.wide:
    mmm.default_context().set_tuning(lda_build="wide")
X, lam0 = np_ref.synth_lda(a.D, a.V, a.K, seed=1, mean_n=a.mean_n)
if a.ilda:
    feats = np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])
    g = mmm.ILDA(a.K, 0.1, 0.1, feats, X, seed=1)
else:
    g = mmm.LDA(a.K, 0.1, 0.1, a.V, X, λ0=lam0)
lib, check = mmm.lib(), mmm._lib.check
def steps(n):
    check(lib.mmm_lda_iterate(g._h, n), g.ctx.h, "iterate"); g.ctx.synchronize()
steps(20)
t0 = time.perf_counter(); steps(a.steps); dt = (time.perf_counter() - t0) / a.steps
nnz = int(g._doc_ptr[-1])
print(json.dumps({"D": a.D, "V": a.V, "K": a.K, "nnz": nnz, "forced_wide": a.wide, "ilda": a.ilda, "us_per_iteration": dt * 1e6, "docs_per_s": a.D / dt,
                  "phi_GBps": 2 * 8.0 * a.K * nnz / dt / 1e9}))
