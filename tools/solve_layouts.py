#!/usr/bin/env python3
"""One GPU: the CTM solve phase under every lane layout (mmm_tuning_opts.solve_lanes) at the shard sizes of an N-GPU strong run of BASELINE
configs 4 and 5 -- where the layout thresholds of csrc/ctm.hip (kSolve10Lanes8Below, kSolve28Lanes32Below) come from.
usage: python3 tools/solve_layouts.py [--configs 4,5] [--shards 1,2,4,8] > gpurun_out/solve_layouts.jsonl"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LANES = {4: (16, 32), 5: (2, 8, 16)}


def nev_stats(env, c, n, corpus):
    """distribution of the per-document LD_MMA evaluation counts in pass 8 of the shard (what a lock-step wave waits for)"""
    import numpy as np
    pkg, ctx = env.pkg, env.ctx
    cfg = bench.CONFIGS[c]
    X, init = corpus
    d0, d1 = pkg.shard_documents(X, n, 0) if n > 1 else (0, len(X))
    K, V = cfg["K"], cfg["V"]
    env.ctx.set_tuning()
    m = pkg.MMCTM(K, [0.1] * len(K), V, X[d0:d1], γ0=init, ctx=ctx) if cfg["model"] == "mmctm" else pkg.IMMCTM(K, [0.1] * len(K), bench.snv3(), X[d0:d1], γ0=init, ctx=ctx)
    pkg._lib.check(pkg.lib().mmm_ctm_iterate(m._h, 8, 1), ctx.h, "iterate")
    st = m.solver_stats(per_doc=True)
    out = {"config": "cfg%d" % c, "n": n, "docs": d1 - d0, "pass": 8}
    for k in ("per_doc_nu", "per_doc_lambda"):
        v = np.abs(st[k]).astype(np.float64)
        out[k] = {"mean": float(v.mean()), "p50": float(np.percentile(v, 50)), "p90": float(np.percentile(v, 90)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
    print(json.dumps(out), flush=True)
    m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="4,5")
    ap.add_argument("--shards", default="1,2,4,8")
    ap.add_argument("--waves", default="0", help="mmm_tuning_opts.solve_waves values to sweep (0 = the library's choice)")
    a = ap.parse_args()
    env = bench.Env(1)
    for c in [int(x) for x in a.configs.split(",")]:
        cfg = bench.CONFIGS[c]
        corpus = bench.make_corpus(c, cfg["docs"], 20261003 + c)
        for n in [int(x) for x in a.shards.split(",")]:
            nev_stats(env, c, n, corpus)
            for lanes in LANES[c]:
              for wv in [int(x) for x in a.waves.split(",")]:
                env.ctx.set_tuning(solve_lanes=lanes, solve_waves=wv)
                r = bench.run_config(env, c, "weak", 10, 2, 3, 0, False, probe=False, proxy_shard=n, corpus=corpus)
                print(json.dumps({"config": "cfg%d" % c, "n": n, "docs": r["config"]["docs_rank0"], "solve_lanes": lanes, "solve_waves": wv,
                                  "ms_per_step": r["ms_per_step"], "ms_per_step_min": r["ms_per_step_min"], "kernel_us": r["iteration"]["kernel_us"],
                                  "kernel": r["roofline"]["kernel"],
                                  "mma_evaluations_per_document": r["roofline"]["f64_valu"]["mma_evaluations_per_document"]}), flush=True)
        env.ctx.set_tuning()
    env.close()


if __name__ == "__main__":
    main()
