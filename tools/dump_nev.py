#!/usr/bin/env python3
"""Per-document LD_MMA evaluation counts of consecutive passes (configs 4 and 5, full size) -> gpurun_out/r05/nev_cfg{c}.npz;
input of tools/sim_solve_schedule.py (how far the previous pass' counts predict the next one's, and what a schedule built on them is worth).
usage: python3 tools/dump_nev.py [--configs 4,5] [--passes 2,3,4,8,9,30,31]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="4,5")
    ap.add_argument("--passes", default="1,2,3,4,8,9,30,31")
    a = ap.parse_args()
    want = sorted(int(x) for x in a.passes.split(","))
    env = bench.Env(1)
    pkg, ctx = env.pkg, env.ctx
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r05"), exist_ok=True)
    for c in [int(x) for x in a.configs.split(",")]:
        cfg = bench.CONFIGS[c]
        X, init = bench.make_corpus(c, cfg["docs"], 20261003 + c)
        K, V = cfg["K"], cfg["V"]
        m = pkg.MMCTM(K, [0.1] * len(K), V, X, γ0=init, ctx=ctx) if cfg["model"] == "mmctm" else pkg.IMMCTM(K, [0.1] * len(K), bench.snv3(), X, γ0=init, ctx=ctx)
        out, done = {}, 0
        for p in want:
            pkg._lib.check(pkg.lib().mmm_ctm_iterate(m._h, p - done, 1), ctx.h, "iterate")
            done = p
            st = m.solver_stats(per_doc=True)
            out["nu_%d" % p] = np.abs(st["per_doc_nu"]).astype(np.int16)
            out["lam_%d" % p] = np.abs(st["per_doc_lambda"]).astype(np.int16)
            print("cfg%d pass %d: nu mean %.1f max %d, lambda mean %.1f max %d" % (c, p, out["nu_%d" % p].mean(), out["nu_%d" % p].max(), out["lam_%d" % p].mean(), out["lam_%d" % p].max()), flush=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r05", "nev_cfg%d.npz" % c), **out)
        m.close()
    env.close()


if __name__ == "__main__":
    main()
