"""Diagnostic (not a benchmark): s_memrealtime stamps (100 MHz) of the merged reduce + ll + M-step launch -- reduce block 1 / wave 0, the tail wave of
block 0, the first ll block.  make -C multimodalmusig.jl_amd/csrc diag && MMM_LIB_PATH=.../libmmmusig_hip_diag.so python tools/diag_red_stamps.py [D]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mmm_pkg, np_ref
pkg = mmm_pkg.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
X, lam0 = np_ref.synth_lda(D, 96, 10, seed=3)
m = pkg.LDA(10, 0.1, 0.1, 96, X, λ0=lam0)
lib = pkg.lib()
lib.mmm_diag_red_stamps.argtypes = [C.c_void_p]; lib.mmm_diag_lda_stamps.argtypes = [C.c_void_p]
for rep in range(3):
    pkg._lib.check(lib.mmm_lda_iterate(m._h, 5), m.ctx.h)
    m.ctx.synchronize()
    st = (C.c_ulonglong * 32)(); assert lib.mmm_diag_red_stamps(st) == 0
    es = (C.c_ulonglong * 16)(); assert lib.mmm_diag_lda_stamps(es) == 0
    s = np.array(st[:], dtype=np.int64)
    t0 = min(s[0], s[8], s[16])
    us = lambda i: (s[i] - t0) / 100.0
    print("D=%d rep %d (us after the first stamped wave of the launch; E-step block 0 ended %.2f us before it):" % (D, rep, (t0 - np.int64(es[9])) / 100.0))
    print("   reduce block 1: start %.2f, partial loads done %.2f, tree %.2f, column sum in hand %.2f, M-step stores done %.2f" % (us(0), us(1), us(2), us(3), us(4)))
    print("   first ll block: start %.2f, own loads arrived %.2f, tables staged %.2f, sweep done %.2f, all waves done %.2f, cell posted %.2f" % (us(16), us(20), us(21), us(22), us(23), us(17)))
    print("   tail wave: start %.2f, starts waiting %.2f, has the ll cells %.2f, done %.2f" % (us(8), us(9), us(10), us(11)))
