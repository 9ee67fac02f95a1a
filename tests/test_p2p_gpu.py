"""The N > 1 data path on the GPU with the xGMI mailbox all-reduce (csrc/p2p.hip), rehearsed on ONE card: three processes,
each a rank with its own context on device 0, exchange IPC handles over gloo, attach each other's mailboxes and fit LDA and
MMCTM on their shards of the documents.  What this checks: handle exchange / mapping, the cell protocol (sequence tags,
slot reuse over many calls, payloads from 1 to 2,450 doubles), rank-order summation (all ranks end with the same bits), and
agreement with the unsharded single-context fit.  What it cannot check on one card is the cross-device visibility of
fine-grained memory -- that is what the library's own known-answer rehearsal at set-up time guards (it falls back to RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 3


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["MMM_P2P_TIMEOUT_S"] = "20"
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import mmm_pkg, np_ref
        pkg = mmm_pkg.load()

        def allgather(b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out

        def allmin(v):
            out = [None] * world
            dist.all_gather_object(out, int(v))
            return min(out)

        ctx = pkg.Context(0)
        ctx.init_p2p(world, rank, allgather, allmin)
        assert ctx.transport == "p2p"
        res = {"rank": rank}
        # ---- LDA: 961-double statistics + ll, every pass
        X, lam0 = np_ref.synth_lda(600, 96, 10, seed=9, mean_n=800)
        d0, d1 = pkg.shard_documents(X, world, rank)
        g = pkg.LDA(10, 0.1, 0.1, 96, X[d0:d1], λ0=lam0, ctx=ctx)
        ll = pkg.fit(g, maxiter=40, tol=1e-4, verbose=False)
        res.update(lda_ll=ll.tolist(), lda_elbo=g.elbo, lda_beta=g.β.tolist(), lda_conv=g.converged, shard=(d0, d1))
        # ---- the same LDA fit through the dense-row E-step build, with the reduce blocks joining the ll sweep (residency lowered so that
        #      the ll blocks loop over their documents) -- the folded exchange rides in that launch too
        ctx.set_tuning(lda_build="dense", resident_cap=64)
        gd = pkg.LDA(10, 0.1, 0.1, 96, X[d0:d1], λ0=lam0, ctx=ctx)
        assert gd.geometry()["dense"] == 1
        lld = pkg.fit(gd, maxiter=40, tol=1e-4, verbose=False)
        res.update(ldad_ll=lld.tolist(), ldad_beta=gd.β.tolist())
        ctx.set_tuning()
        # ---- MMCTM: moments + gamma sums (1 x 7 x ... doubles) and M-double ll
        Xm, g0 = np_ref.synth_mm(240, [40, 24], [5, 4], seed=4, means=[600, 80], empty_frac=0.1)
        e0, e1 = pkg.shard_documents(Xm, world, rank)
        c = pkg.MMCTM([5, 4], [0.1, 0.1], [40, 24], Xm[e0:e1], γ0=g0, ctx=ctx)
        llc = pkg.fit(c, maxiter=6, tol=0.0, verbose=False)
        res.update(ctm_ll=llc.tolist(), ctm_elbo=c.elbo, ctm_mu=c.μ.tolist(), ctm_gamma=c._get("gamma").tolist())
        # ---- IMMCTM over rows of counts (forced: the shards are small), fit with a stopping tolerance: the rule runs on the device from the
        #      GLOBAL ll row, so every rank must stop in the same pass; update_alpha off
        SNV3 = [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]
        Xi, _ = np_ref.synth_mm(330, [96], [6], seed=14, means=[900], empty_frac=0.05)
        gi0 = np.random.default_rng(3).integers(1, 101, size=6 * 14).astype(np.float64)
        f0, f1 = pkg.shard_documents(Xi, world, rank)
        ctx.set_tuning(ctm_build="dense")
        ci = pkg.IMMCTM([6], [0.1], SNV3, Xi[f0:f1], γ0=gi0, ctx=ctx)
        assert ci.geometry()["tdense"]
        lli = pkg.fit(ci, maxiter=25, tol=2e-3, verbose=False)
        res.update(imm_ll=np.asarray(lli).tolist(), imm_elbo=ci.elbo, imm_gamma=ci._get("gamma").tolist(), imm_conv=ci.converged)
        ctx.set_tuning()
        if rank == 0:
            plain_i = pkg.Context(0)
            cis = pkg.IMMCTM([6], [0.1], SNV3, Xi, γ0=gi0, ctx=plain_i)
            llis = pkg.fit(cis, maxiter=25, tol=2e-3, verbose=False)
            res.update(ref_imm_ll=np.asarray(llis).tolist(), ref_imm_elbo=cis.elbo)
        if rank == 0:
            # the same fits, unsharded, on a plain context
            plain = pkg.Context(0)
            gs = pkg.LDA(10, 0.1, 0.1, 96, X, λ0=lam0, ctx=plain)
            lls = pkg.fit(gs, maxiter=40, tol=1e-4, verbose=False)
            cs = pkg.MMCTM([5, 4], [0.1, 0.1], [40, 24], Xm, γ0=g0, ctx=plain)
            llcs = pkg.fit(cs, maxiter=6, tol=0.0, verbose=False)
            res.update(ref_lda_ll=lls.tolist(), ref_lda_elbo=gs.elbo, ref_lda_beta=gs.β.tolist(), ref_ctm_ll=llcs.tolist(), ref_ctm_elbo=cs.elbo,
                       ref_ctm_mu=cs.μ.tolist())
        dist.barrier()
        q.put(res)
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:      # noqa: BLE001 -- report instead of leaving the parent waiting
        import traceback
        q.put({"rank": rank, "error": "%s\n%s" % (e, traceback.format_exc())})


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_three_ranks_on_one_card_p2p_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    errs = [r["error"] for r in res if "error" in r]
    assert not errs, "\n".join(errs)
    res = sorted(res, key=lambda r: r["rank"])
    r0 = res[0]
    for r in res[1:]:
        # rank-order summation: every rank holds the same bits
        assert r["lda_ll"] == r0["lda_ll"] and r["lda_beta"] == r0["lda_beta"] and r["lda_conv"] == r0["lda_conv"]
        assert r["ctm_ll"] == r0["ctm_ll"] and r["ctm_mu"] == r0["ctm_mu"] and r["ctm_gamma"] == r0["ctm_gamma"]
        assert r["ldad_ll"] == r0["ldad_ll"] and r["ldad_beta"] == r0["ldad_beta"]
        assert r["imm_ll"] == r0["imm_ll"] and r["imm_gamma"] == r0["imm_gamma"] and r["imm_conv"] == r0["imm_conv"]
    assert sum(b - a for a, b in (r["shard"] for r in res)) == 600
    # sharded == unsharded up to the order of the sums
    assert len(r0["lda_ll"]) == len(r0["ref_lda_ll"])
    np.testing.assert_allclose(r0["lda_ll"], r0["ref_lda_ll"], rtol=1e-11)
    np.testing.assert_allclose(r0["lda_beta"], r0["ref_lda_beta"], rtol=1e-9)
    np.testing.assert_allclose(sum(r["lda_elbo"] for r in res) / WORLD, r0["lda_elbo"], rtol=1e-13)     # the ELBO is a global sum
    np.testing.assert_allclose(r0["lda_elbo"], r0["ref_lda_elbo"], rtol=1e-10)
    assert len(r0["ldad_ll"]) == len(r0["ref_lda_ll"])
    np.testing.assert_allclose(r0["ldad_ll"], r0["ref_lda_ll"], rtol=1e-11)
    np.testing.assert_allclose(r0["ldad_beta"], r0["ref_lda_beta"], rtol=1e-9)
    np.testing.assert_allclose(r0["ctm_ll"], r0["ref_ctm_ll"], rtol=1e-5)
    np.testing.assert_allclose(r0["ctm_elbo"], r0["ref_ctm_elbo"], rtol=1e-5)
    np.testing.assert_allclose(r0["ctm_mu"], r0["ref_ctm_mu"], rtol=1e-3, atol=1e-5)
    # IMMCTM over rows of counts, stopped by its tolerance: the same number of passes on every rank (asserted above through the ll
    # rows) and, up to the forking of the LD_MMA trajectories, the unsharded fit
    n = min(len(r0["imm_ll"]), len(r0["ref_imm_ll"]))
    assert n >= 11 and abs(len(r0["imm_ll"]) - len(r0["ref_imm_ll"])) <= 2
    np.testing.assert_allclose(np.asarray(r0["imm_ll"])[:n], np.asarray(r0["ref_imm_ll"])[:n], rtol=1e-4)
    np.testing.assert_allclose(sum(r["imm_elbo"] for r in res) / WORLD, r0["imm_elbo"], rtol=1e-12)
