#!/usr/bin/env python3
"""Records the argument ranges the table-driven exp / log of the LD_MMA objectives (csrc/mmm_arith.h: ar_exp_tab, ar_log_tab) actually see in
BASELINE configs 3-5, through the debug hook of the order-matched CPU restatement (orc_twin_arg_ranges): config 3 on the shipped BRCA
tables, configs 4 and 5 on the SURVEY 8d corpora at their FULL sizes, six passes each from the random initialisation (the first passes
reach furthest).  Output: tests/golden/table_argument_ranges.json, which tests/test_twin_cpu.py sweeps against mpmath.
usage: python3 tests/golden/make_table_ranges.py [--docs4 50000] [--docs5 100000] [--passes 6]"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_ref  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SNV3 = [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]


def geom(D, MK):
    L = 16 if MK <= 16 else (32 if MK <= 32 else 64)
    G = 64 // L
    return dict(L=L, waves_e=8, grid_e=max(1, min((D + 8 * G - 1) // (8 * G), 512)), grid_m=max(1, min((D + 31) // 32, 1024)))


def ranges(o, passes):
    r = np.zeros(4)
    orc.lib().orc_twin_arg_ranges(r, 1)
    for _ in range(passes):
        assert o.twin_pass(True) == 0
    orc.lib().orc_twin_arg_ranges(r, 1)
    return {"exp_min": r[0], "exp_max": r[1], "log_min": r[2], "log_max": r[3]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs4", type=int, default=50000); ap.add_argument("--docs5", type=int, default=100000); ap.add_argument("--passes", type=int, default=6)
    a = ap.parse_args()
    import mmm_pkg
    pkg = mmm_pkg.load()           # (TSV reader and count formatting only: no device)
    _, samples, snv = pkg.read_counts_tsv(os.path.join(HERE, "brca-eu_snv_counts.tsv"))
    _, _, sv = pkg.read_counts_tsv(os.path.join(HERE, "brca-eu_sv_counts.tsv"))
    X3 = pkg.format_counts_mmctm([{s: snv[:, i] for i, s in enumerate(samples)}, {s: sv[:, i] for i, s in enumerate(samples)}], samples)
    rng = np.random.default_rng(2)
    g3 = np.concatenate([rng.integers(1, 101, size=(7, 96)).astype(np.float64).ravel(), rng.integers(1, 101, size=(7, 48)).astype(np.float64).ravel()])
    out = {"passes": a.passes, "what": "min / max finite argument of ar_exp_tab and ar_log_tab in the nu- and lambda-objectives of every LD_MMA evaluation"}
    out["config3_brca_560_docs"] = ranges(orc.CtmOracle([7, 7], [0.1, 0.1], X3, V=[96, 48], gamma0=g3, geometry=geom(560, 14)), a.passes)
    X4, g4 = np_ref.synth_mm(a.docs4, [96, 38, 32], [10, 10, 8], seed=20261003 + 4)
    out["config4_%d_docs" % a.docs4] = ranges(orc.CtmOracle([10, 10, 8], [0.1] * 3, X4, V=[96, 38, 32], gamma0=np.concatenate([x.ravel() for x in g4]),
                                                            geometry=geom(a.docs4, 28)), a.passes)
    X5, _ = np_ref.synth_mm(a.docs5, [96], [10], seed=20261003 + 5)
    g5 = np.random.default_rng(1).integers(1, 101, size=10 * 14).astype(np.float64)
    out["config5_%d_docs" % a.docs5] = ranges(orc.CtmOracle([10], [0.1], X5, features=SNV3, gamma0=g5, geometry=geom(a.docs5, 10)), a.passes)
    json.dump(out, open(os.path.join(HERE, "table_argument_ranges.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
