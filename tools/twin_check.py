"""Pass-by-pass bit comparison of the HIP CTM path with the order-matched CPU restatement (oracle/mmm_twin.c).
Run on the GPU box: python tools/twin_check.py [npass].  Prints, per case and pass, the number of differing entries per array."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mmm_pkg
import np_ref
from oracle import oracle as orc

mmm = mmm_pkg.load()
SNV3 = [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]


def ndiff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel(); b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    return int((a.view(np.int64) != b.view(np.int64)).sum()), float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def case(name, D, K, V, means, seed, feats=None, npass=12):
    X, g0 = np_ref.synth_mm(D, V, K, seed=seed, means=means, empty_frac=0.05)
    alpha = [0.1] * len(K)
    if feats is None:
        g = mmm.MMCTM(K, alpha, V, X, γ0=g0)
        geo = g.geometry()
        o = orc.CtmOracle(K, alpha, X, V=V, gamma0=np.concatenate([x.ravel() for x in g0]), geometry=geo)
    else:
        GM = sum(K[m] * int(np.asarray(feats[m]).max(axis=0).sum()) for m in range(len(K)))
        g0f = np.random.default_rng(seed).integers(1, 101, size=GM).astype(np.float64)
        g = mmm.IMMCTM(K, alpha, feats, X, γ0=g0f)
        geo = g.geometry()
        o = orc.CtmOracle(K, alpha, X, features=feats, gamma0=g0f, geometry=geo)
    MK = sum(K)
    print("== %s D=%d K=%s geometry=%s" % (name, D, K, geo))
    print("   init: Elnphi", ndiff(g._get("Elnphi"), o.Elnphi))
    for it in range(npass):
        mmm._lib.check(mmm.lib().mmm_ctm_iterate(g._h, 1, 1), g.ctx.h, "iterate")
        o.twin_pass(True)
        st = g.solver_stats(per_doc=True)
        r = {"lam": ndiff(g.lam_matrix(), o.lam), "nu": ndiff(g.nu_matrix(), o.nu), "zeta": ndiff(g._get("zeta"), o.zeta),
             "mu": ndiff(g.μ, o.mu), "Sigma": ndiff(np.asarray(g.Σ).ravel(order="F"), o.Sigma),
             "invSigma": ndiff(np.asarray(g.invΣ).ravel(order="F"), o.invSigma),
             "gamma": ndiff(g._get("gamma"), o.gamma), "Elnphi": ndiff(g._get("Elnphi"), o.Elnphi)}
        nn = int((st["per_doc_nu"] != o.nev_nu[:D]).sum()); nl = int((st["per_doc_lambda"] != o.nev_lambda[:D]).sum())
        bad = {k: v for k, v in r.items() if v[0]}
        print("   pass %2d: nev mismatches nu %d lambda %d; differing arrays: %s" % (it + 1, nn, nl, bad if bad else "none"))
        if bad or nn or nl:
            return False
    return True


if __name__ == "__main__":
    npass = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    ok = True
    ok &= case("mm [5,4]", 300, [5, 4], [40, 24], [600, 80], 4, npass=npass)
    ok &= case("mm [7,7]", 560, [7, 7], [96, 48], [3000, 60], 7, npass=npass)
    ok &= case("mm [10,10,8]", 700, [10, 10, 8], [96, 38, 32], [2000, 150, 100], 8, npass=npass)
    ok &= case("imm [10]", 800, [10], [96], [2500], 9, feats=SNV3, npass=npass)
    ok &= case("mm [3,3] packed 6", 500, [3, 3], [40, 24], [600, 80], 41, npass=npass)
    ok &= case("mm [6,6] packed 12", 500, [6, 6], [40, 24], [600, 80], 42, npass=npass)
    ok &= case("mm [20,6]", 90, [20, 6], [96, 32], [2500, 120], 15, npass=npass)
    ok &= case("mm [24,17,23]", 40, [24, 17, 23], [30, 30, 30], [200, 200, 200], 91, npass=npass)
    print("ALL BIT-IDENTICAL" if ok else "DIFFERENCES FOUND")
