"""The C ABI from plain C: include/mmmusig.h must be valid C99, and examples/lda_fit.c -- a binding's call sequence for
`LDA(...)` + `fit!` with no Python or Julia in the process -- must link against the library (CPU) and reproduce the Python
mirror's fit bit for bit on the shipped BRCA SNV table (GPU)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "multimodalmusig.jl_amd", "lib")
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")


def _build(tmp_path):
    exe = str(tmp_path / "lda_fit")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "lda_fit.c"),
           "-L", LIBDIR, "-lmmmusig_hip", "-Wl,-rpath," + LIBDIR, "-L", ROCM_LIB, "-Wl,-rpath-link," + ROCM_LIB, "-lm", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_header_is_c99_and_example_links(mmm, tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "mmmusig.h"\nint main(void) { return MMM_VERSION > 0 ? 0 : 1; }\n')
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "hdr.o")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    _build(tmp_path)          # every symbol the example uses resolves against the built library


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_c_program_reproduces_the_python_fit(mmm, tmp_path):
    exe = _build(tmp_path)
    table = os.path.join(ROOT, "tests", "golden", "brca-eu_snv_counts.tsv")
    K, maxiter, seed = 7, 30, 5
    p = subprocess.run([exe, table, str(K), str(maxiter), "0", str(seed)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    c = json.loads(p.stdout)
    terms, samples, counts = mmm.read_counts_tsv(table)
    X = mmm.format_counts_lda({s: counts[:, i] for i, s in enumerate(samples)}, samples)
    V = len(terms)
    lcg, lam0 = seed, np.zeros(V * K)
    for i in range(V * K):
        lcg = (lcg * 6364136223846793005 + 1442695040888963407) % 2 ** 64
        lam0[i] = 1 + (lcg >> 33) % 100
    g = mmm.LDA(K, 0.1, 0.1, V, X, λ0=lam0.reshape(V, K, order="F"))
    ll = mmm.fit(g, maxiter=maxiter, tol=0.0, verbose=False)
    assert (c["D"], c["V"], c["K"], c["n_iter"]) == (len(samples), V, K, maxiter)
    assert c["ll"] == ll.tolist() and c["elbo"] == g.elbo             # same library, same inputs: same bits
    assert c["theta_doc1"] == g.θ[:, 0].tolist()
