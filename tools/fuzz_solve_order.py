#!/usr/bin/env python3
"""Randomised check of the persistent solve kernel's document hand-out (order_range, refill, balanced ranges) against the order-matched oracle:
the shapes with a several-coordinates-per-lane build (sum K = 10 as 2 x 5 / 8 x 2, sum K = 28 as 16 x 2 / 32 x 1) on a PRETENDED small device
(mmm_tuning_opts.geometry_cus 1...8), so that a wave's range holds anything from fewer documents than slots to several hundred; 4 passes,
every bit of lambda / nu / zeta / mu / Sigma / gamma and every evaluation count.  Not part of the test-suite.
usage: python3 tools/fuzz_solve_order.py [n_cases=24] [seed=0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import mmm_pkg  # noqa: E402
from oracle import oracle  # noqa: E402
import test_ctm_gpu as T  # noqa: E402

mmm = mmm_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    imm = bool(rng.integers(0, 2))
    lanes = int(rng.choice([2, 8])) if imm else int(rng.choice([16, 32]))
    cus = int(rng.choice([1, 2, 3, 5, 8]))
    D = int(rng.choice([37, 150, 333, 800, 1500, 2600]))
    off = () if rng.integers(0, 4) else ("ctm_solve_order",)
    mmm.default_context().set_tuning(solve_lanes=lanes, geometry_cus=cus, disable=off)
    seed = int(rng.integers(1 << 30))
    if imm:
        kw = dict(D=D, K=[10], V=[96], seed=seed, means=[int(rng.choice([200, 1500]))], imm_features=T.SNV3)
    else:
        kw = dict(D=D, K=[10, 10, 8], V=[96, 38, 32], seed=seed, means=[int(rng.choice([300, 2000])), 150, 100])
    try:
        X, g, o = T._pair(mmm, oracle, order="device", empty_frac=float(rng.choice([0.0, 0.15])), **kw)
        geo = g.geometry()
        MK = sum(kw["K"])
        for it in range(4):
            mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
            assert o.twin_pass(True) == 0
            st = g.solver_stats(per_doc=True)
            assert np.array_equal(st["per_doc_nu"], o.nev_nu[:D]) and np.array_equal(st["per_doc_lambda"], o.nev_lambda[:D]), "evaluation counts differ in pass %d" % (it + 1)
            T._same_state(g, o, D, MK)
        per = D / max(1, geo["solve_waves"])
        print("case %2d ok: %s D=%d lanes=%d cus=%d waves=%d (%.1f documents per wave, %d slots) order %s" % (
            case, "imm10" if imm else "mm28", D, geo["Ls"], cus, geo["solve_waves"], per, 64 // geo["Ls"], "off" if off else "on"), flush=True)
        g.close()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("case %2d FAILED: imm=%s D=%d lanes=%d cus=%d seed=%d: %s" % (case, imm, D, lanes, cus, seed, str(e)[:300]), flush=True)
mmm.default_context().set_tuning()
print("%d cases, %d failed" % (n, bad))
sys.exit(1 if bad else 0)
