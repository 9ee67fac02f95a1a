"""Diagnostic (not a benchmark): start and numerator-posted time (s_memrealtime, 100 MHz) of every ll block of the merged reduce + ll + M-step launch.
make -C multimodalmusig.jl_amd/csrc diag && MMM_LIB_PATH=.../libmmmusig_hip_diag.so python tools/diag_ll_blocks.py [D]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
pkg = mmm_pkg.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
X, lam0 = np_ref.synth_lda(D, 96, 10, seed=3)
m = pkg.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
lib = pkg.lib()
lib.mmm_diag_ll_times.argtypes = [C.c_void_p]
for rep in range(3):
    pkg._lib.check(lib.mmm_lda_iterate(m._h, 5), m.ctx.h)
    m.ctx.synchronize()
    st = (C.c_ulonglong * 1024)(); assert lib.mmm_diag_ll_times(st) == 0
    t = np.array(st[:], dtype=np.int64).reshape(2, 512)
    n = int((t[1] > 0).sum())
    s, e = t[0, :n], t[1, :n]
    t0 = s.min()
    us = lambda a: (a - t0) / 100.0
    dur = us(e) - us(s)
    late = np.argsort(e)[-8:][::-1]
    print("D=%d rep %d: %d ll blocks; start %.2f .. %.2f us (median %.2f); numerator posted %.2f .. %.2f (median %.2f, 90 %% %.2f); duration %.2f .. %.2f (median %.2f)" % (
        D, rep, n, us(s).min(), us(s).max(), np.median(us(s)), us(e).min(), us(e).max(), np.median(us(e)), np.quantile(us(e), 0.9), dur.min(), dur.max(), np.median(dur)))
    print("   the 8 latest: " + ", ".join("lb %d (start %.2f, dur %.2f)" % (i, us(s)[i], dur[i]) for i in late))
