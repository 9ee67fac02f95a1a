#!/usr/bin/env python3
"""The restart sweep of scripts/run_mmctm.jl (`fit_model`, :163-182) on the shipped BRCA-EU tables through the batched HIP fit.
Usage: python tools/run_restarts.py [--restarts R] [--k 7 7] [--batch B] [--seed S]
Prints one JSON line: wall time of stage 1 (R restarts, maxiter 1000, tol 1e-4) and stage 2 (tol 1e-5), passes executed."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mmm_pkg

ap = argparse.ArgumentParser()
ap.add_argument("--restarts", type=int, default=100)
ap.add_argument("--k", type=int, nargs="+", default=[7, 7])
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--out", default="", help="directory for sigs.tsv, props.tsv, mean.tsv, cov.tsv, cor.tsv (run_mmctm.jl:272-290)")
args = ap.parse_args()
mmm = mmm_pkg.load()
from multimodalmusig_jl_amd import restarts as rs  # noqa: E402
GOLD = os.path.join(ROOT, "tests", "golden")
terms_snv, samples, snv = mmm.read_counts_tsv(os.path.join(GOLD, "brca-eu_snv_counts.tsv"))
terms_sv, _, sv = mmm.read_counts_tsv(os.path.join(GOLD, "brca-eu_sv_counts.tsv"))
X = mmm.format_counts_mmctm([{s: snv[:, i] for i, s in enumerate(samples)}, {s: sv[:, i] for i, s in enumerate(samples)}], samples)
K, V = args.k, [96, 48]
alpha = [0.1, 0.1]
seeds = np.random.default_rng(args.seed).integers(1, 2 ** 62, size=args.restarts)
mmm.MMCTM(K, alpha, V, X, seed=0).close()        # context + module load outside the timing
t0 = time.perf_counter()
g, best, all_ll = rs.fit_seed_models(X, K, alpha, V, seeds, batch_size=args.batch or None)
t1 = time.perf_counter()
model = rs.seed_and_fit_restart(X, K, alpha, V, g)
t2 = time.perf_counter()
if args.out:
    from multimodalmusig_jl_amd import io as mio
    os.makedirs(args.out, exist_ok=True)
    mio.write_sigs(os.path.join(args.out, "sigs.tsv"), model, [terms_snv, terms_sv], ["snv", "sv"])
    mio.write_props(os.path.join(args.out, "props.tsv"), model, samples, ["snv", "sv"])
    mio.write_matrix(os.path.join(args.out, "mean.tsv"), model.μ)
    mio.write_matrix(os.path.join(args.out, "cov.tsv"), model.Σ)
    mio.write_matrix(os.path.join(args.out, "cor.tsv"), mio.cov2cor(model.Σ))
print(json.dumps({"restarts": args.restarts, "K": K, "docs": len(X), "stage1_s": t1 - t0, "stage2_s": t2 - t1,
                  "stage1_best_ll": best.tolist(), "stage1_ll_spread": [float(all_ll[:, m].min()) for m in range(2)],
                  "stage2_ll": model.ll.tolist(), "stage2_converged": bool(model.converged), "stage2_elbo": model.elbo}))
