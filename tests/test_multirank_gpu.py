"""The RCCL code path on a single-GPU box: a one-rank communicator with MMM_FORCE_RCCL=1 makes every all-reduce of the
packed sufficient statistics go through ncclAllReduce on the context stream.  Results must equal the plain path (a one-rank
sum is the identity): exactly for the CTM path, to the last bits for LDA (whose single-GPU path merges the M-step into the reduce launch and sums a topic's
column in a different order).  Runs in a child process because the flag is read once per process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, mmm_pkg, np_ref
pkg = mmm_pkg.load()
ctx = pkg.Context(0)
if os.environ.get("TEST_LDA_WIDE"):
    ctx.set_tuning(lda_build="wide")
if os.environ.get("MMM_FORCE_RCCL"):
    ctx.init_comm(1, 0, pkg.comm_unique_id())
X, lam0 = np_ref.synth_lda(300, 96, 10, seed=9, mean_n=800)
g = pkg.LDA(10, 0.1, 0.1, 96, X, λ0=lam0, ctx=ctx)
ll = pkg.fit(g, maxiter=14, tol=0.0, verbose=False)
Xm, g0 = np_ref.synth_mm(120, [40, 24], [5, 4], seed=4, means=[600, 80], empty_frac=0.1)
c = pkg.MMCTM([5, 4], [0.1, 0.1], [40, 24], Xm, γ0=g0, ctx=ctx)
llc = pkg.fit(c, maxiter=5, tol=0.0, verbose=False)
print("RESULT " + json.dumps({"transport": ctx.transport, "ll": ll.tolist(), "elbo": g.elbo, "lam": g.λ.sum(), "llc": llc.tolist(), "elboc": c.elbo, "mu": c.μ.tolist()}))
"""


def _run(force, mailboxes=False, wide=False):
    env = dict(os.environ)
    env.pop("MMM_FORCE_RCCL", None); env.pop("MMM_P2P_ONE_RANK", None); env.pop("TEST_LDA_WIDE", None)
    if wide:
        env["TEST_LDA_WIDE"] = "1"
    if force:
        env["MMM_FORCE_RCCL"] = "1"
    if mailboxes:
        env["MMM_P2P_ONE_RANK"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_one_rank_rccl_path_equals_plain_path():
    a, b = _run(False), _run(True)
    # LDA: same statistics, but the merged single-GPU launch adds a topic's column sum block by block while k_lda_mstep adds it lane
    # by lane: last-bit differences, nothing more
    np.testing.assert_allclose(a["ll"], b["ll"], rtol=1e-13)
    np.testing.assert_allclose([a["elbo"], a["lam"]], [b["elbo"], b["lam"]], rtol=1e-13)
    assert a["llc"] == b["llc"] and a["elboc"] == b["elboc"] and a["mu"] == b["mu"]


def test_one_rank_mailbox_setup_over_rccl_and_folded_exchange():
    """What `bench.py --gpus N` runs on a multi-GPU node, with N = 1: mmm_comm_init_rank sets the xGMI mailboxes up over the
    RCCL communicator (handle all-gather, attach, rehearsal, unanimous agreement) and the LDA iteration then sends its
    statistics from the reduce launch and receives them in the M-step launch.  With one rank there is no peer to hear from,
    so the results must be the plain path's."""
    a, b = _run(False), _run(True, mailboxes=True)
    assert a["transport"] == "none" and b["transport"] == "p2p"
    np.testing.assert_allclose(a["ll"], b["ll"], rtol=1e-13)
    np.testing.assert_allclose([a["elbo"], a["lam"]], [b["elbo"], b["lam"]], rtol=1e-13)
    assert a["llc"] == b["llc"] and a["elboc"] == b["elboc"] and a["mu"] == b["mu"]


def test_one_rank_collectives_on_the_wide_vocabulary_path():
    """The wide LDA data flow (tests/test_lda_wide_gpu.py) with its statistics going through the all-reduce: ncclAllReduce, and
    the stand-alone mailbox kernel (the exchange is not folded into the wide kernels)."""
    a = _run(False, wide=True)
    for b in (_run(True, wide=True), _run(True, mailboxes=True, wide=True)):
        np.testing.assert_allclose(a["ll"], b["ll"], rtol=1e-13)
        np.testing.assert_allclose([a["elbo"], a["lam"]], [b["elbo"], b["lam"]], rtol=1e-13)
