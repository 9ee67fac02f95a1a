"""Diagnostic: how much of the solve phase's wave time is lock-step loss?  From the per-document LD_MMA evaluation counts of a pass:
sum over waves of max over the wave's G documents, against the sum over documents / G.  python tools/lockstep_stats.py [config] [passes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
pkg = mmm_pkg.load()
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 4
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if cfgn == 4:
    K, V, D = [10, 10, 8], [96, 38, 32], 50000
else:
    K, V, D = [10], [96], 100000
X, init = np_ref.synth_mm(D, V, K, seed=20261003 + cfgn)
if cfgn == 4:
    m = pkg.MMCTM(K, [0.1] * len(K), V, X, γ0=init)
else:
    SNV3 = [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]
    GM = sum(K[i] * int(f.max(axis=0).sum()) for i, f in enumerate(SNV3))
    m = pkg.IMMCTM(K, [0.1], SNV3, X, γ0=np.random.default_rng(1).integers(1, 101, size=GM).astype(np.float64))
lib = pkg.lib()
for it in range(npass):
    pkg._lib.check(lib.mmm_ctm_iterate(m._h, 1, 1), m.ctx.h, "iterate")
    if it in (0, 1, 2, 5, 10, npass - 1):
        st = m.solver_stats(per_doc=True)
        out = []
        for name in ("per_doc_nu", "per_doc_lambda"):
            n = np.abs(st[name]).astype(np.int64)
            row = "%s mean %.2f max %d" % (name[8:], n.mean(), n.max())
            for G in (2, 4, 16, 32):
                pad = (-len(n)) % G
                nn = np.concatenate([n, np.zeros(pad, dtype=np.int64)]).reshape(-1, G)
                row += "  G=%d: x%.3f" % (G, nn.max(axis=1).sum() * G / n.sum())
            out.append(row)
        print("pass %2d: " % (it + 1) + " | ".join(out))
    if it == npass - 2:
        stp = m.solver_stats(per_doc=True)
        prev = (np.abs(stp["per_doc_nu"]).copy(), np.abs(stp["per_doc_lambda"]).copy())
    if it == npass - 1:
        st = m.solver_stats(per_doc=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "nev_cfg%d.npz" % cfgn), nu=np.abs(st["per_doc_nu"]), lam=np.abs(st["per_doc_lambda"]), nu_prev=prev[0], lam_prev=prev[1])
