#!/usr/bin/env python3
"""Secondary measurement (not the driver's bench.py): MMCTM / IMMCTM E-step docs/sec on BASELINE configs 4 and 5 (1 GPU).
Usage: python tools/bench_ctm.py [--config 3|4|5] [--docs N] [--steps K] [--warmup W] [--cpu] [--restarts R]
--restarts R: R models over the same corpus advanced together (restart batching, scripts/run_mmctm.jl:77-134); docs/s then
counts documents x restarts.  Config 3 is the BRCA-sized restart case (560 docs, K = [7, 7])."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmm_pkg, np_ref

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--docs", type=int, default=0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--restarts", type=int, default=0)
ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle on a 300-document sample")
args = ap.parse_args()
pkg = mmm_pkg.load()
SNV3 = [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]
if args.config == 3:
    K, V, D, feats = [7, 7], [96, 48], args.docs or 560, None
elif args.config == 4:
    K, V, D, feats = [10, 10, 8], [96, 38, 32], args.docs or 50000, None
else:
    K, V, D, feats = [10], [96], args.docs or 100000, SNV3
t0 = time.time()
X, g0 = np_ref.synth_mm(D, V, K, seed=20261003 + args.config)
print("corpus built in %.1f s" % (time.time() - t0), file=sys.stderr)
alpha = [0.1] * len(K)
R = max(args.restarts, 1)
rs = args.restarts if args.restarts else None
if feats is None:
    m = pkg.MMCTM(K, alpha, V, X, γ0=None if rs else g0, restarts=rs, seed=7)
else:
    GM = sum(K[i] * int(feats[i].max(axis=0).sum()) for i in range(len(K)))
    gflat = np.random.default_rng(1).integers(1, 101, size=GM).astype(np.float64)
    m = pkg.IMMCTM(K, alpha, feats, X, γ0=None if rs else gflat, restarts=rs, seed=7)
lib = pkg.lib(); chk = pkg._lib.check


def passes(n):
    if rs:      # tol = -1: no restart ever stops, every pass advances all R
        ll = np.zeros(R * n * len(K)); ni = np.zeros(R, dtype=np.int32); cv = np.zeros(R, dtype=np.int32)
        chk(lib.mmm_ctm_fit_batch(m._h, n, -1.0, 1, ll.ctypes.data, ni.ctypes.data, cv.ctypes.data, None), m.ctx.h)
    else:
        chk(lib.mmm_ctm_iterate(m._h, n, 1), m.ctx.h)
    m.ctx.synchronize()


passes(args.warmup)
m.ctx.profile_begin()
t0 = time.perf_counter()
passes(args.steps)
dt = time.perf_counter() - t0
n, kms = m.ctx.profile_end()
st = m.solver_stats()
nnz = sum(m._nnz); MK = sum(K)
# algorithmic bytes per document per pass (theta not stored): X 8 B/nonzero + lambda in/out + nu in/out + zeta
algo = R * (8.0 * nnz + (4 * MK + len(K)) * 8.0 * D)
res = {"config": args.config, "docs": D, "restarts": R, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3, "docs_per_s": R * D * args.steps / dt,
       "estep_kernel_avg_us": kms / max(n, 1) * 1e3, "estep_algo_GBps": algo / (kms / max(n, 1) * 1e-3) / 1e9,
       "mma_evals_per_doc_last_pass": (st["n_eval_nu"] + st["n_eval_lambda"]) / D, "n_capped": st["n_capped"]}
if args.cpu:
    from oracle import oracle as orc
    Ds = 300
    if feats is None:
        o = orc.CtmOracle(K, alpha, X[:Ds], V=V, gamma0=np.concatenate([g.ravel() for g in g0]))
    else:
        o = orc.CtmOracle(K, alpha, X[:Ds], features=feats, gamma0=gflat)
    t0 = time.perf_counter()
    o.fit(maxiter=5, tol=0.0)
    dtc = time.perf_counter() - t0
    res["cpu_oracle_docs_per_s_1thread"] = Ds * 5 / dtc
    res["speedup_vs_cpu_oracle"] = res["docs_per_s"] / res["cpu_oracle_docs_per_s_1thread"]
print(json.dumps(res))
