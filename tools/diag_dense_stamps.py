"""Diagnostic (not a benchmark): in-kernel s_memtime stamps of the dense-row LDA E-step kernel (k_lda_estep_dense), block 0 / wave 0.
Usage on the GPU box: make -C multimodalmusig.jl_amd/csrc diag && MMM_LIB_PATH=.../libmmmusig_hip_diag.so python tools/diag_dense_stamps.py [D]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
pkg = mmm_pkg.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
X, lam0 = np_ref.synth_lda(D, 96, 10, seed=3)
pkg.default_context().set_tuning(lda_build="dense")
m = pkg.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
assert m.geometry()["dense"], "not the dense-row build"
lib = pkg.lib()
for rep in range(3):
    pkg._lib.check(lib.mmm_lda_iterate(m._h, 5), m.ctx.h)
    m.ctx.synchronize()
    st = (C.c_ulonglong * 16)()
    lib.mmm_diag_lda_stamps.argtypes = [C.c_void_p]
    assert lib.mmm_diag_lda_stamps(st) == 0
    s = np.array(st[:8], dtype=np.int64)
    names = ["staging + first loads + first prologue + barrier", "steps (to the last term phase)", "last gamma sums", "statistics -> slab", "barrier", "block sums + partial", "-"]
    d = np.diff(s)
    rt = np.array(st[8:12], dtype=np.int64)
    us = (rt[1] - rt[0]) / 100.0
    print("D=%d rep %d block0 %.2f us (%.2f GHz); last block start +%.2f us, end +%.2f us | " % (D, rep, us, (s[7] - s[0]) / us / 1e3, (rt[2] - rt[0]) / 100.0, (rt[3] - rt[0]) / 100.0)
          + ", ".join("%s %.2f us" % (n, x / (s[7] - s[0]) * us) for n, x in zip(names, d)))
