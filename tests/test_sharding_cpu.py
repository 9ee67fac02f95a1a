"""world_size-2 gloo test (CPU) of the N > 1 data path: documents are sharded with the product's `shard_documents`,
every rank runs the E-step on its shard only (here with the CPU oracle standing in for the kernels), the packed sufficient
statistics [sum lambda | sum nu | sum lambda lambda' | gamma sums] (+ the ll numerators) are summed with ONE all-reduce
per iteration, and every rank applies the same M-step.  Checks: (i) every rank ends with bit-identical globals, (ii) they
equal the unsharded run to rounding, (iii) the raw-moment covariance Sigma = (diag sum nu + sum ll')/D - mu mu' the GPU
M-step uses equals the reference's two-pass formula (MMCTM.jl:204-210).  Same for the LDA lambda statistics."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mmm_pkg, np_ref
    from oracle import oracle as orc
    pkg = mmm_pkg.load()
    K, V = [3, 4], [20, 12]
    X, g0 = np_ref.synth_mm(41, V, K, seed=3, means=[300, 60], empty_frac=0.2)
    gam0 = np.concatenate([g.ravel() for g in g0])
    d0, d1 = pkg.shard_documents(X, world, rank)
    Xs = X[d0:d1]
    m = orc.CtmOracle(K, [0.1, 0.1], Xs, V=V, gamma0=gam0)
    MK, D = 7, len(X)
    GT = sum(k * v for k, v in zip(K, V))
    lls = []
    for it in range(3):
        m.estep_range(0, len(Xs))
        lam = m.lam.reshape(len(Xs), MK); nu = m.nu.reshape(len(Xs), MK)
        # gamma sums of the shard = update_gamma minus alpha
        m.update_gamma()
        gsum = m.gamma - 0.1
        packed = np.concatenate([lam.sum(0), nu.sum(0), (lam.T @ lam).ravel(order="F"), gsum])
        t = torch.from_numpy(packed.copy())
        dist.all_reduce(t)                                   # the ONE collective of the iteration
        s = t.numpy()
        mu = s[:MK] / D
        Sig = (np.diag(s[MK:2 * MK]) + s[2 * MK:2 * MK + MK * MK].reshape(MK, MK, order="F")) / D - np.outer(mu, mu)
        m.mu[:] = mu; m.Sigma[:] = Sig.ravel(); m.invSigma[:] = np.linalg.inv(Sig).ravel(order="F")
        m.gamma[:] = 0.1 + s[2 * MK + MK * MK:]
        m.update_Elnphi(); m.update_props(); m.update_phi()
        # per-modality ll numerators / denominators
        llo = m.loglik()
        Nm = np.array([sum(x[mm][:, 1].sum() for x in Xs) for mm in range(2)], dtype=np.float64)
        t2 = torch.from_numpy(np.concatenate([np.nan_to_num(llo) * Nm, Nm]))
        dist.all_reduce(t2)
        lls.append((t2[:2] / t2[2:]).numpy().copy())
    q.put((rank, d0, d1, m.mu.copy(), m.Sigma.copy(), m.gamma.copy(), np.array(lls)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_sharded_estep_single_allreduce_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, a0, a1, mu0, S0, g0, ll0), (_, b0, b1, mu1, S1, g1, ll1) = res
    assert a0 == 0 and a1 == b0 and b1 == 41 and 0 < a1 < 41
    # (i) bit-identical globals on both ranks
    assert np.array_equal(mu0, mu1) and np.array_equal(S0, S1) and np.array_equal(g0, g1) and np.array_equal(ll0, ll1)
    # (ii)+(iii) equal to the unsharded two-pass reference formulas
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import np_ref
    from oracle import oracle as orc
    K, V = [3, 4], [20, 12]
    X, gg = np_ref.synth_mm(41, V, K, seed=3, means=[300, 60], empty_frac=0.2)
    m = orc.CtmOracle(K, [0.1, 0.1], X, V=V, gamma0=np.concatenate([g.ravel() for g in gg]))
    lls = []
    for it in range(3):
        m.estep_range(0, 41); m.update_mu(); assert m.update_Sigma() == 0; m.update_gamma(); m.update_props(); m.update_phi()
        lls.append(m.loglik())
    np.testing.assert_allclose(mu0, m.mu, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(S0, m.Sigma, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(g0, m.gamma, rtol=1e-6)
    np.testing.assert_allclose(ll0, np.array(lls), rtol=1e-7)


def test_shard_documents_balances_nonzeros():
    sys.path.insert(0, ROOT)
    import mmm_pkg
    pkg = mmm_pkg.load()
    rng = np.random.default_rng(0)
    X = [np.stack([np.arange(1, w + 1), np.ones(w, dtype=np.int64)], axis=1) for w in rng.integers(0, 97, size=1000)]
    nz = np.array([x.shape[0] for x in X])
    for world in (2, 4, 8):
        bounds = [pkg.shard_documents(X, world, r) for r in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == 1000 and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
        loads = np.array([nz[a:b].sum() for a, b in bounds])
        assert loads.max() <= 1.05 * loads.mean() + 96
