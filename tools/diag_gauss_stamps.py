"""Diagnostic (not a benchmark): s_memtime stamps of the Gaussian M-step block (update_mu! / update_Sigma!: the Gauss-Jordan inversion).
make -C multimodalmusig.jl_amd/csrc diag && MMM_LIB_PATH=.../libmmmusig_hip_diag.so python tools/diag_gauss_stamps.py [fused|alone]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
pkg = mmm_pkg.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
ctx = pkg.default_context()
if mode == "alone":
    ctx.set_tuning(disable=("ctm_fused_gauss",))
elif mode == "alone_wide":
    ctx.set_tuning(disable=("ctm_fused_gauss", "ctm_pipe_gauss"))
K, V = [10, 10, 8], [96, 38, 32]
X, g0 = np_ref.synth_mm(6249, V, K, seed=20261007)
m = pkg.MMCTM(K, [0.1] * 3, V, X, γ0=g0)
lib = pkg.lib()
lib.mmm_diag_gauss_stamps.argtypes = [C.c_void_p]
for rep in range(3):
    pkg._lib.check(lib.mmm_ctm_iterate(m._h, 4, 1), m.ctx.h)
    m.ctx.synchronize()
    st = (C.c_ulonglong * 96)(); assert lib.mmm_diag_gauss_stamps(st) == 0
    s = np.array(st[:], dtype=np.int64)
    n = 28
    real_us = (s[95] - s[94]) / 100.0
    tot = s[91] - s[0]
    print("%s rep %d: block %.2f us real, %d ticks (%.2f GHz); mu %d, Sigma fill %d, inverse %d, invSigma stores %d" % (
        mode, rep, real_us, tot, tot / real_us / 1e3, s[1] - s[0], s[2] - s[1], s[90] - s[2], s[91] - s[90]))
    if s[4] == 0:      # the one-barrier-per-column variant stamps once per column
        per = np.diff(np.array([s[2]] + [s[3 + 3 * c] for c in range(n)]))
        print("   per column (ticks): median %d, first %s, last %s" % (np.median(per), per[:4].tolist(), per[-3:].tolist()))
        continue
    cols = np.array([[s[3 + 3 * c] - (s[2] if c == 0 else s[5 + 3 * (c - 1)]), s[4 + 3 * c] - s[3 + 3 * c], s[5 + 3 * c] - s[4 + 3 * c]] for c in range(n)])
    print("   per column (ticks): pivot search + barrier median %d, swap/scale + barrier %d, eliminate + barrier %d; first columns %s" % (
        np.median(cols[:, 0]), np.median(cols[:, 1]), np.median(cols[:, 2]), cols[:3].tolist()))
