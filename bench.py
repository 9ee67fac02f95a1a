#!/usr/bin/env python3
"""bench.py -- E-step docs/sec of the MI355X backend on BASELINE.json's configurations.

A "step" = one outer EM iteration (the body of fit!: E-step over every document, M-step reduction, log-likelihood;
LDA.jl:201-209, MMCTM.jl:462-479, IMMCTM.jl:440-451) over one batch of synthetic documents resident in HBM.

  --config 2 (default)  BASELINE configs[1]: LDA K=10, alpha=eta=0.1, 10,000 documents x 96 SNV terms      (headline metric)
  --config 4            BASELINE configs[3]: MMCTM K=[10,10,8], 50,000 documents x (96,38,32) terms
  --config 5            BASELINE configs[4]: IMMCTM K=[10], 100,000 documents x 96 terms, features I=3, J=[6,4,4]
  --scaling weak        (default) every rank holds its own corpus of the configuration's size
  --scaling strong      the ONE corpus of the configuration's size, documents sharded over the ranks (balanced by nonzeros)

The default invocation (config 2) also measures configs 4 and 5 for a bounded number of steps, and config 2's model on 640k documents
per GPU, and attaches them to the same JSON line under "also" (--no-also switches that off); the headline keys are config 2's.

N > 1, default invocation: the headline stays config 2 weak (10k documents per GPU) on the default transport, and the line also carries
  "strong"           config 2's ONE 10k-document corpus sharded over the N ranks (the literal ">= 6x at 8 GPUs" target of BASELINE.json)
  also.cfg4 / cfg5   the 50k / 100k-document corpus SHARDED over the N ranks (strong: that is BASELINE configs[3] / configs[4]), each with
                     a "weak" entry (every rank its own 50k / 100k corpus) beside it
and every one of these entries once per transport: the xGMI mailboxes ("p2p", the default) and ncclAllReduce ("rccl": north_star's mandated
collective) under the entry's "transports" key -- each with docs_per_rank, allreduce_per_rank, comm_nranks_per_rank and the per-GPU
HBM-roofline fraction.

N > 1: one process per GPU.  `python3 bench.py --gpus N` starts its N rank processes ITSELF (children of a parent that never touches
the GPU; rank r -> device r mod #devices) and relays rank 0's line; under `python -m torch.distributed.run` (WORLD_SIZE set by the
launcher) the process is a rank and starts nothing.  One all-reduce of the packed sufficient statistics and of the ll numerators per
iteration (xGMI mailboxes, ncclAllReduce as fallback; the line reports which, per rank, in config.allreduce_per_rank).  Ranks that have to
share a card (more ranks than devices: a one-card rehearsal of the N > 1 path) talk over gloo on the host side and set the mailboxes
up from host-exchanged IPC handles -- RCCL refuses two ranks on one device.

Contract: W untimed warm-up steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over
ranks.  That timed region is repeated R times (--repeats, default 9): `ms_per_step` / `value` are the MEDIAN region, min / max beside
them (a 20-step region of config 2 lasts 0.4 ms; one region alone carries the +-40 us of the bracketing synchronisations).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# dmabuf IPC (the only kind the host driver supports): RCCL and the xGMI mailboxes both map peer memory through it.  Already
# exported on the GPU boxes; set here as well so that a launcher with a scrubbed environment still works.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_VALU_PEAK_TF = 78.6     # MI355X vector FP64; 256 CUs x 4 SIMD x 16 f64 lanes x 2.4 GHz x 2 flop
N_SIMD = 1024               # 256 CUs x 4 SIMDs
SCLK_HZ = 2.4e9

CONFIGS = {
    2: dict(model="lda", K=10, V=96, docs=10000, name="LDA K=10 alpha=eta=0.1, %d docs x 96 SNV terms (BASELINE configs[1])"),
    4: dict(model="mmctm", K=[10, 10, 8], V=[96, 38, 32], docs=50000, name="MMCTM K=[10,10,8] alpha=0.1, %d docs x (96,38,32) terms (BASELINE configs[3])"),
    5: dict(model="immctm", K=[10], V=[96], docs=100000, name="IMMCTM K=[10] alpha=0.1, %d docs x 96 terms, features I=3 J=[6,4,4] (BASELINE configs[4])"),
}


def host_cores():
    """cores this process may actually use: the cgroup CPU quota if there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return n


def snv3():
    import numpy as np
    return [np.array([[t // 16 + 1, (t // 4) % 4 + 1, t % 4 + 1] for t in range(96)])]      # SURVEY section 8d cfg 5 factorisation


# ----------------------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline_lda(X, lam0, K, alpha, eta, target_s=12.0):
    """The CPU oracle (a compiled, allocation-free C port of the reference's Julia loop; 1 thread like the reference) timed on a
    bounded number of passes over the SAME corpus."""
    from oracle import oracle as orc
    V = lam0.shape[0]
    o = orc.LdaOracle(K, alpha, eta, X, V=V, lambda0=lam0)

    def one_pass():
        o.update_gamma(); o.update_phi(); o.update_lambda(); o.update_beta(); o.update_theta()
        return o.loglik()

    t0 = time.perf_counter(); one_pass(); t1 = time.perf_counter() - t0
    n = max(2, min(400, int(target_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        one_pass()
    dt = time.perf_counter() - t0
    res = {"value": len(X) * n / dt, "unit": "docs/s", "cores": 1, "kind": "port",
           "sample": "%d passes over the same %d-doc corpus, %.1f s, single thread (C oracle; the Julia reference "
                     "cannot run on this box)" % (n, len(X), dt)}
    # the same iteration with its document loops on every core the box grants (OpenMP; SURVEY section 8d asks for both figures)
    try:
        cores = host_cores()
        L = orc.lib_omp()
        L.orc_omp_set_threads(cores)
        t0 = time.perf_counter(); o.pass_omp(); t1 = time.perf_counter() - t0
        m = max(2, min(2000, int(4.0 / max(t1, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(m):
            o.pass_omp()
        dtm = time.perf_counter() - t0
        res["all_cores"] = {"value": len(X) * m / dtm, "unit": "docs/s", "cores": int(cores), "threads": int(L.orc_omp_threads()),
                            "kind": "port (OpenMP over documents)",
                            "sample": "%d passes, %.1f s; cores = cgroup quota / affinity of this process" % (m, dtm)}
    except Exception as e:       # noqa: BLE001 -- the single-thread figure is the contract; this one is an extra
        res["all_cores"] = {"error": str(e)}
    return res


def cpu_baseline_ctm(cfg, X, g0, target_s=14.0):
    """fit!(::MMCTM / ::IMMCTM) by the index-order C oracle, one thread, on a bounded sample of the same corpus: whole passes
    (E-step with the two LD_MMA solves per document, M-step, log-likelihood)."""
    import numpy as np
    from oracle import oracle as orc
    Ds = min(len(X), 4000)
    alpha = [0.1] * len(cfg["K"])
    pick = np.sort(np.random.default_rng(20261005).choice(len(X), size=Ds, replace=False))      # a random draw of the corpus, not its head
    Xs = [X[int(d)] for d in pick]
    if cfg["model"] == "mmctm":
        o = orc.CtmOracle(cfg["K"], alpha, Xs, V=cfg["V"], gamma0=np.concatenate([g.ravel() for g in g0]))
    else:
        o = orc.CtmOracle(cfg["K"], alpha, Xs, features=snv3(), gamma0=g0)
    t0 = time.perf_counter(); o.fit(maxiter=1, tol=0.0); t1 = time.perf_counter() - t0
    n = max(2, min(200, int(target_s / max(t1, 1e-3))))
    t0 = time.perf_counter(); o.fit(maxiter=n, tol=0.0); dt = time.perf_counter() - t0
    return {"value": Ds * n / dt, "unit": "docs/s", "cores": 1, "kind": "port",
            "sample": "%d passes over %d documents drawn at random (seeded) from the same corpus, %.1f s, single thread (C oracle, index-order "
                      "variant; the Julia reference cannot run on this box)" % (n, Ds, dt)}


# ----------------------------------------------------------------------------------------------------------- parity probes
def parity_probe_lda(pkg, K, alpha, eta, V, seed):
    """ELBO / phi relative error GPU vs oracle on a bounded sample (200 docs, 12 passes)."""
    import numpy as np
    import np_ref
    from oracle import oracle as orc
    X, lam0 = np_ref.synth_lda(200, V, K, seed=seed)
    g = pkg.LDA(K, alpha, eta, V, X, λ0=lam0)
    pkg.fit(g, maxiter=12, tol=0.0, verbose=False)
    o = orc.LdaOracle(K, alpha, eta, X, V=V, lambda0=lam0)
    o.fit(maxiter=12, tol=0.0)
    phi_g, phi_o = g.phi_flat(), o.phi.reshape(-1, K)
    rel_phi = float(np.max(np.abs(phi_g - phi_o) / np.maximum(np.abs(phi_o), 1e-12)))
    rel_elbo = abs(g.elbo - o.elbo_value) / abs(o.elbo_value)
    g.close()
    return {"elbo_rel_err_vs_oracle": rel_elbo, "phi_max_rel_err_vs_oracle": rel_phi}


def parity_full_lda(pkg, ctx, X, lam0, K, V):
    """The benchmark's own corpus (BASELINE configs[1] at full size) through 12 passes on the GPU and on the oracle: ll history, γ, λ, Elnβ
    and ELBO at 1e-9, ϕ / θ at 1e-5 (the reference's bar), then fit!(tol = 1e-4): same pass count, same `converged`."""
    import numpy as np
    from oracle import oracle as orc
    t0 = time.perf_counter()
    g = pkg.LDA(K, 0.1, 0.1, V, X, λ0=lam0, ctx=ctx)
    ll = pkg.fit(g, maxiter=12, tol=0.0, verbose=False)
    o = orc.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0)
    llo = o.fit(maxiter=12, tol=0.0)

    def rel(a, b):
        a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0
    D = len(X)
    res = {"docs": D, "passes": 12, "ll_history_max_rel_err": rel(ll, llo), "gamma_max_rel_err": rel(g.γ.T, o.gamma), "lambda_max_rel_err": rel(g.λ.T, o.lam),
           "Elnbeta_max_rel_err": rel(g.Elnβ.T, o.Elnbeta), "theta_max_rel_err": rel(g.θ.T, o.theta),
           "phi_max_rel_err": float(np.max(np.abs(g.phi_flat() - o.phi.reshape(-1, K)) / np.maximum(np.abs(o.phi.reshape(-1, K)), 1e-12))),
           "elbo_rel_err": abs(g.elbo - o.elbo_value) / abs(o.elbo_value)}
    g.close()
    g = pkg.LDA(K, 0.1, 0.1, V, X, λ0=lam0, ctx=ctx)
    ll = pkg.fit(g, maxiter=60, tol=1e-4, verbose=False)
    o = orc.LdaOracle(K, 0.1, 0.1, X, V=V, lambda0=lam0)
    llo = o.fit(maxiter=60, tol=1e-4)
    res.update({"fit_tol_1e-4_passes_gpu": int(len(ll)), "fit_tol_1e-4_passes_oracle": int(len(llo)), "fit_converged_gpu": bool(g.converged),
                "fit_converged_oracle": bool(o.converged), "fit_elbo_rel_err": abs(g.elbo - o.elbo_value) / abs(o.elbo_value),
                "wall_s": time.perf_counter() - t0})
    g.close()
    return res


def parity_probe_ctm(pkg, cfg, seed):
    """ELBO / theta relative error and per-document LD_MMA evaluation counts, GPU vs the order-matched oracle, on a bounded sample
    (400 docs, 12 passes)."""
    import numpy as np
    import np_ref
    from oracle import oracle as orc
    K, V = cfg["K"], cfg["V"]
    alpha = [0.1] * len(K)
    X, g0 = np_ref.synth_mm(400, V, K, seed=seed)
    if cfg["model"] == "mmctm":
        g = pkg.MMCTM(K, alpha, V, X, γ0=g0)
        o = orc.CtmOracle(K, alpha, X, V=V, gamma0=np.concatenate([x.ravel() for x in g0]), geometry=g.geometry())
    else:
        GM = sum(K[i] * int(f.max(axis=0).sum()) for i, f in enumerate(snv3()))
        g0f = np.random.default_rng(seed).integers(1, 101, size=GM).astype(np.float64)
        g = pkg.IMMCTM(K, alpha, snv3(), X, γ0=g0f)
        o = orc.CtmOracle(K, alpha, X, features=snv3(), gamma0=g0f, geometry=g.geometry())
    pkg.fit(g, maxiter=12, tol=0.0, verbose=False)
    o.fit(maxiter=12, tol=0.0)
    st = g.solver_stats(per_doc=True)
    th_g, th_o = g._get("theta"), o.theta
    res = {"elbo_rel_err_vs_oracle": abs(g.elbo - o.elbo_value) / abs(o.elbo_value),
           "theta_max_rel_err_vs_oracle": float(np.max(np.abs(th_g - th_o) / np.maximum(np.abs(th_o), 1e-300))),
           "mma_evaluation_counts_equal_for_all_documents": bool(np.array_equal(st["per_doc_nu"], o.nev_nu[:400]) and np.array_equal(st["per_doc_lambda"], o.nev_lambda[:400])),
           "oracle": "order-matched variant (oracle/mmm_twin.c), 400 docs x 12 passes"}
    g.close()
    return res


# ----------------------------------------------------------------------------------------------------------- self-launch
def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(n, argv, timeout_s):
    """`python3 bench.py --gpus N` without a launcher: start the N rank processes as CHILDREN (this parent has made no GPU call --
    nothing above imports torch or loads the library), relay rank 0's JSON line, fail if any rank fails or the job times out."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MMM_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    deadline = time.time() + timeout_s
    rc = 0
    first_bad = None                      # (rank, code) of the first rank seen with a non-zero exit code -- before the parent ends the others
    # rank 0's stdout is drained by a thread (its JSON line can exceed a pipe buffer); the parent polls: the moment one rank fails, the
    # others -- which would sit in a collective until its own time-out -- are ended too
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed and first_bad is None:
                first_bad = failed[0]
            if failed or all(c is not None for c in codes):
                break
            if time.time() > deadline:
                rc = 124
                print("bench.py: the %d-rank job did not finish within %d s" % (n, timeout_s), file=sys.stderr)
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:       # exactly the processes started above, by handle
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
    reader.join(timeout=10)
    out0 = "".join(c for c in chunks if c)
    bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
    line = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if bad or rc or not line:
        print("bench.py: rank exit codes %s (first failure: %s)%s" % (bad, first_bad, "" if line else "; rank 0 printed no JSON line"), file=sys.stderr)
        sys.stdout.write(out0 or "")
        sys.exit(rc or (first_bad[1] if first_bad else (bad[0][1] if bad else 1)) or 1)
    print(line[-1])
    sys.exit(0)


class Env:
    """process group, device and library context of one rank"""

    def __init__(self, gpus):
        import numpy as np
        import torch                       # first: the library then binds to the HIP runtime torch has loaded
        import torch.distributed as dist
        import mmm_pkg
        self.np, self.torch, self.dist = np, torch, dist
        self.pkg = pkg = mmm_pkg.load()
        self.world = world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world != gpus:
            if rank == 0:
                print("bench.py: --gpus %d but WORLD_SIZE=%d" % (gpus, world), file=sys.stderr)
            sys.exit(2)
        ndev = torch.cuda.device_count()
        if ndev < 1:
            print("bench.py: no GPU visible; the HIP path is the only path", file=sys.stderr)
            sys.exit(3)
        self.ndev = ndev
        self.device = local % ndev
        self.shared_card = world > ndev          # several ranks per device: host-side gloo + host-exchanged mailbox handles
        torch.cuda.set_device(self.device)
        self.ctx = ctx = pkg.Context(self.device)
        self.cuda_pg = False
        self.has_comm = False              # an RCCL communicator exists on ctx
        if world > 1 and not self.shared_card:
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            self.cuda_pg = True
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(pkg.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            ctx.init_comm(world, rank, bytes(uid.cpu().numpy().tobytes()))
            self.has_comm = True
        elif world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            ctx.init_p2p(world, rank, self.allgather_obj, lambda v: min(self.allgather_obj(int(v))))
        elif os.environ.get("MMM_FORCE_RCCL"):
            # single-GPU rehearsal of the collective path: a one-rank communicator, every all-reduce goes through RCCL
            ctx.init_comm(1, 0, pkg.comm_unique_id())
            self.has_comm = True

    def transports(self):
        """[(name, switch)] of the all-reduce transports this job can run, the current one first; switch() is called by every rank.  A
        transport that cannot run here is listed in self.transport_unavailable with the reason."""
        ctx = self.ctx
        self.transport_unavailable = {}
        if self.world == 1 and not self.has_comm:
            return [(ctx.transport, lambda: None)]
        cur = ctx.transport
        out = [(cur, lambda: None)]
        if cur == "p2p":
            if self.has_comm:     # an RCCL communicator exists beside the mailboxes
                out.append(("rccl", lambda: ctx.p2p_enable(False)))
            else:
                self.transport_unavailable["rccl"] = "the ranks share one card (a rehearsal): RCCL refuses two ranks on one device, the mailboxes were set up from host-exchanged handles"
        elif cur == "rccl":
            self.transport_unavailable["p2p"] = "the mailboxes were not set up (MMM_P2P=0, or their rehearsal failed on some rank)"
        return out

    def restore_transport(self, name):
        if name == "p2p" and self.ctx.transport != "p2p":
            self.ctx.p2p_enable(True)

    def allgather_obj(self, o):
        if self.world == 1:
            return [o]
        out = [None] * self.world
        self.dist.all_gather_object(out, o)
        return out

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def bcast(self, arr):
        """rank 0's float64 array on every rank"""
        if self.world == 1:
            return arr
        t = self.torch.from_numpy(self.np.ascontiguousarray(arr))
        if self.cuda_pg:
            t = t.cuda()
        self.dist.broadcast(t, 0)
        return t.cpu().numpy()

    def allmax(self, v):
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device="cuda" if self.cuda_pg else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def csrc_sha16(model):
    """hash of the sources the kernels of one model family are built from (the counter files record it, tools/pmc_summary.py)"""
    import hashlib
    d = os.path.join(ROOT, "multimodalmusig.jl_amd", "csrc")
    import glob
    files = ["dev_math.h", "mmm_arith.h", "mmm_logtab.h", "mmm_exptab.h", "mmm_internal.h"] + sorted(
        os.path.basename(f) for f in glob.glob(os.path.join(d, ("lda" if model == "lda" else "ctm") + "*.[hc]*")))      # lda.hip + lda_*.cuh | ctm.hip + ctm_*.cuh
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def load_pmc(tname, model, kernel_prefix):
    """Committed counter summary (rocprofv3 --pmc passes of this command, tools/pmc_run.sh + tools/pmc_summary.py) of the dominant kernel
    -- attached only when it was taken from THIS build of THIS kernel: the file's kernel name must start with the template instance the
    running handle launches and its csrc hash must equal the sources'.  Returns (summary or None, file name or None, reason or None)."""
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        tp = os.path.join(ROOT, "profiles", "%s_traffic_%s.json" % (rnd, tname))
        if not os.path.exists(tp):
            continue
        tr = json.load(open(tp))
        base = os.path.basename(tp)
        if kernel_prefix not in tr.get("kernel", ""):
            return None, base, "profiles/%s holds counters of `%s`, the handle launches `%s...`: not attached" % (base, tr.get("kernel"), kernel_prefix)
        if tr.get("csrc_sha16") != csrc_sha16(model):
            return None, base, "profiles/%s was collected from sources with hash %s, the running build has %s (re-run tools/refresh_evidence.sh pmc): not attached" % (
                base, tr.get("csrc_sha16"), csrc_sha16(model))
        return tr, base, None
    return None, None, "no counter file for %s under profiles/" % tname


def make_corpus(cfg_id, D, seed):
    """SURVEY section 8d generator: (X, init) of one configuration at D documents"""
    import numpy as np
    import np_ref
    key = (cfg_id, D, seed)
    if key in _CORPORA:
        return _CORPORA[key]
    cfg = CONFIGS[cfg_id]
    K, V = cfg["K"], cfg["V"]
    if cfg["model"] == "lda":
        X, init = np_ref.synth_lda(D, V, K, seed=seed)
    else:
        X, init = np_ref.synth_mm(D, V, K, seed=seed)
        if cfg["model"] == "immctm":
            GM = sum(K[i] * int(f.max(axis=0).sum()) for i, f in enumerate(snv3()))
            init = np.random.default_rng(1).integers(1, 101, size=GM).astype(np.float64)
    if D <= 100000:      # (the shard proxy re-uses the three BASELINE corpora; the 640k-document one is used once)
        _CORPORA[key] = (X, init)
    return X, init


_CORPORA = {}


def run_config(env, cfg_id, scaling, steps, warmup, repeats, docs, cpu_baseline, cpu_target_s=None, every_transport=False, probe=True,
               proxy_shard=0, corpus=None):
    """Measure one configuration on the ranks of `env`; rank 0 gets the result dictionary, the others None.  every_transport (N > 1): after
    the full measurement on the current all-reduce transport, the timed regions once more on every other transport the job can run, on a
    fresh model over the same shard, under res["transports"].  proxy_shard = N (one GPU): the model runs over rank 0's shard of an N-rank
    strong run of the corpus (`shard_documents(X, N, 0)`), no communicator -- the per-GPU work of that run without its exchange
    (shard_proxy).  corpus = (X, init): use this corpus instead of generating one."""
    np, pkg, ctx, world, rank = env.np, env.pkg, env.ctx, env.world, env.rank
    import np_ref
    cfg = CONFIGS[cfg_id]
    t_start = time.perf_counter()

    def note(what):      # progress on stderr (stdout carries the one JSON line): where a multi-rank run is, should it ever stall
        print("[bench rank %d/%d] cfg %d%s: %s (+%.1f s)" % (rank, world, cfg_id, " docs=%d" % docs if docs else "", what, time.perf_counter() - t_start),
              file=sys.stderr, flush=True)
    # ---- corpus: SURVEY section 8d generator, corpus seed = 20261003 + config index; weak: every rank its own corpus (seed + 1000 rank),
    # strong: every rank generates the one corpus and keeps its nnz-balanced contiguous shard
    Dcfg = docs or cfg["docs"]
    seed = 20261003 + (1 if cfg_id == 2 else cfg_id)
    K, V = cfg["K"], cfg["V"]
    corpus_seed = seed + (1000 * rank if scaling == "weak" else 0)
    if corpus is not None:
        X, init = corpus
    else:
        X, init = make_corpus(cfg_id, Dcfg, corpus_seed)
    if scaling == "strong" and world > 1:
        d0, d1 = pkg.shard_documents(X, world, rank)
        X = X[d0:d1]
    elif proxy_shard > 1:
        d0, d1 = pkg.shard_documents(X, proxy_shard, 0)
        X = X[d0:d1]
    D = len(X)
    if world > 1 and cfg["model"] == "lda":                  # same lambda0 everywhere: take rank 0's
        init = env.bcast(init)
    elif world > 1 and cfg["model"] == "mmctm":
        flat = env.bcast(np.concatenate([g.ravel() for g in init])); o = 0; new = []
        for g in init:
            new.append(flat[o:o + g.size].reshape(g.shape).copy()); o += g.size
        init = new
    lib = pkg.lib()
    note("corpus ready, %d documents" % D)

    def make_model():
        if cfg["model"] == "lda":
            mdl = pkg.LDA(K, 0.1, 0.1, V, X, λ0=init, ctx=ctx)
            return mdl, int(mdl._doc_ptr[-1]), lambda n: pkg._lib.check(lib.mmm_lda_iterate(mdl._h, n), ctx.h, "mmm_lda_iterate")
        if cfg["model"] == "mmctm":
            mdl = pkg.MMCTM(K, [0.1] * len(K), V, X, γ0=init, ctx=ctx)
        else:
            mdl = pkg.IMMCTM(K, [0.1] * len(K), snv3(), X, γ0=init, ctx=ctx)
        return mdl, int(sum(mdl._nnz)), lambda n: pkg._lib.check(lib.mmm_ctm_iterate(mdl._h, n, 1), ctx.h, "mmm_ctm_iterate")

    def timed_regions(run_fn):
        """the contract's timed region, `repeats` times: K steps between barrier + device synchronisation on both sides, MAX over ranks"""
        run_fn(warmup)
        ctx.synchronize()
        out = []
        for _ in range(max(1, repeats)):
            env.barrier()
            t0 = time.perf_counter()
            run_fn(steps)
            env.barrier()
            out.append(env.allmax(time.perf_counter() - t0))
        return out

    def event_regions(run_fn):
        """the same K steps, `repeats` times, between ONE pair of HIP events recorded on the library's own stream (what the device spends on
        the K steps; the contract's wall-clock region above also carries its two bracketing synchronisations and barriers)"""
        torch = env.torch
        ext = torch.cuda.ExternalStream(int(ctx.stream), device=torch.device("cuda", env.device))
        out = []
        for _ in range(max(1, repeats)):
            ctx.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            run_fn(steps)
            e1.record(ext)
            e1.synchronize()
            out.append(e0.elapsed_time(e1) * 1e-3)
        return out

    transports_avail = env.transports()
    primary = transports_avail[0][0]
    model, nnz, run = make_model()
    note("model created")
    regions = timed_regions(run)
    dt = float(np.median(regions))
    note("timed regions done (%.4f ms per step, all-reduce: %s)" % (dt / steps * 1e3, primary))
    try:
        ev_regions = event_regions(run)
    except Exception as e:       # noqa: BLE001 -- an extra: the contract's figure does not depend on it
        ev_regions, ev_error = None, "%s: %s" % (type(e).__name__, e)

    # Per-kernel durations: the same K steps again with the launches of one phase of the pass bracketed by a HIP event pair on the
    # library's stream (mmm_ctx_profile_select).  Kept out of the timed regions above because each hipEventRecord opens a ~5.6 us bubble
    # in the otherwise back-to-back kernel stream (rocprofv3 trace: profiles/), i.e. +25 % on config 2's ms_per_step.
    def span_us(phase, repeat=1):
        ctx.profile_begin(repeat=repeat, phase=phase)
        run(steps)
        n, ms = ctx.profile_end()
        return n, (ms / max(n, 1)) * 1e3

    if cfg["model"] == "lda":
        n_launch, span1 = span_us(0)
        # ... and once more with the (idempotent) E-step kernel launched twice inside every span: the difference of the two spans is
        # the kernel's own duration, without the ~4 us an event pair adds around a single launch (what rocprofv3 reports)
        _, span2 = span_us(0, repeat=2)
        avg_us = span2 - span1 if span2 > span1 > 0 else span1
        pair_us = max(span1 - avg_us, 0.0)               # what an event pair adds around one launch
        _, tail = span_us(1)
        phases = {"estep": avg_us, "reduce_ll_mstep": max(tail - pair_us, 0.0)}
        step_us_profiled = dt / steps * 1e6
    else:
        # millisecond passes: every phase bracketed in ONE repeat of the K steps (the solves get shorter as the fit converges, so
        # phases measured in different passes would not add up), and that repeat's own wall time beside them
        span2 = None
        ctx.profile_begin(repeat=1, phase=8)
        env.barrier(); t0 = time.perf_counter()
        run(steps)
        ph = ctx.profile_end_phases()
        step_us_profiled = (time.perf_counter() - t0) / steps * 1e6
        names = {0: "solve", 1: "theta", 2: "moments_reduce_topics", 3: "gauss_props_loglik"}
        phases = {names[i]: ms / steps * 1e3 for i, (n, ms) in ph.items()}
        n_launch = ph[0][0]
        avg_us = span1 = ph[0][1] / max(ph[0][0], 1) * 1e3        # a millisecond kernel: the ~4 us of the event pair are < 0.5 %
    ctx.profile_begin(repeat=1, phase=0); ctx.profile_end()
    avg_s = avg_us * 1e-6

    note("phase spans done")
    ll_last = None
    if cfg["model"] == "lda":
        # on EVERY rank: reading the history flushes the lagged log-likelihood of the last pass, and that ends in the ranks' exchange -- called
        # by rank 0 alone it left rank 0 one exchange ahead of its peers (harmless at the very end of a run, fatal before the next configuration)
        ll = np.zeros(1); n = pkg._lib.C.c_int()
        pkg._lib.check(lib.mmm_lda_ll_history(model._h, ll.ctypes.data, 1, pkg._lib.C.byref(n)), ctx.h)
        ll_last = float(ll[0])
    transports = env.allgather_obj(ctx.transport)
    nranks_seen = env.allgather_obj(int(lib.mmm_comm_nranks(ctx.h)))
    docs_per_rank = env.allgather_obj(D)
    res = None
    if rank == 0:
        docs_step = sum(docs_per_rank)
        ms_step = dt / steps * 1e3
        res = {
            "metric": "E-step docs/sec", "value": docs_step * steps / dt, "unit": "docs/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": ms_step,
            "ms_per_step_min": min(regions) / steps * 1e3, "ms_per_step_max": max(regions) / steps * 1e3, "repeats": len(regions),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "ms_per_step_events": (float(np.median(ev_regions)) / steps * 1e3) if ev_regions else None,
            "ms_per_step_events_min": (min(ev_regions) / steps * 1e3) if ev_regions else None,
            # (LDA: every pass does the same work, the two kinds of regions are comparable; a CTM fit's later passes need fewer LD_MMA
            # evaluations, and the event regions come after the wall-clock ones)
            "bracketing_us_per_region": ((dt - float(np.median(ev_regions))) * 1e6) if (ev_regions and cfg["model"] == "lda") else None,
            "timing_note": ("ms_per_step: the contract's wall clock over K steps between barrier + device synchronisation on both sides, median of the "
                            "regions.  ms_per_step_events: the same K steps between one HIP-event pair on the library's stream (rank 0), median of as many "
                            "regions; bracketing_us_per_region = what the wall-clock region carries beyond the device's K steps -- at K = 20 steps of 20 us it "
                            "is several per cent of the region" + ("" if cfg["model"] == "lda" else ".  CTM: the event regions are the passes AFTER the wall-clock "
                            "regions of the same fit (passes %d-%d against %d-%d), whose solves need fewer evaluations -- the two figures are not the same work" % (
                                warmup + len(regions) * steps + 1, warmup + 2 * len(regions) * steps, warmup + 1, warmup + len(regions) * steps))) if ev_regions else ("event timing unavailable: " + ev_error),
            "config": {"workload": (cfg["name"] % Dcfg) + (" per GPU" if scaling == "weak" and world > 1 else "") +
                                   ", nnz(rank 0)=%d, one EM iteration per step" % nnz,
                       "docs_rank0": D, "docs_total": docs_step, "docs_per_rank": docs_per_rank, "terms": V, "topics": K,
                       "sharding": "docs x%d (%s)" % (world, scaling), "allreduce": ctx.transport, "allreduce_per_rank": transports,
                       "comm_nranks": nranks_seen[0], "comm_nranks_per_rank": nranks_seen,
                       "devices_visible": env.ndev, "ranks_share_a_card": bool(env.shared_card)},
        }
        if scaling == "strong" and world > 1:
            res["config"]["note"] = ("strong scaling of a %d-document corpus: the per-iteration exchange has a fixed latency of the order of the "
                                     "iteration itself, so the 10k-document LDA corpus cannot speed up 6x on 8 GPUs; weak scaling is the "
                                     "curve that can" % Dcfg)
        res["iteration"] = {"kernel_us": phases, "sum_kernel_us": sum(phases.values()), "step_us": step_us_profiled,
                            "kernel_fraction_of_step": sum(phases.values()) / step_us_profiled if step_us_profiled > 0 else None,
                            "timing": "HIP-event spans per phase of the pass (mmm_ctx_profile_select) in repeats of the timed K steps; LDA: one phase per "
                                      "repeat, spans corrected by the event-pair overhead measured differentially on the E-step kernel, step_us = "
                                      "the timed regions' median; CTM: all phases in one repeat, step_us = that repeat's wall time per step"}
        if cfg["model"] == "lda":
            res["ll_last"] = ll_last
            # dominant kernel: k_lda_estep (update_γ!/ϕ! sweep + λ statistics).  Algorithmic bytes per launch by SURVEY section 8d's
            # figure: 8 B per nonzero (term,count) + gamma_t read + Elntheta and gamma_{t+1} writes (3 x 8 B x K per document); phi stays in
            # registers, the topic table (7.7 KB) is L2-resident and excluded.  (The ll of the previous pass, which re-reads X and
            # gamma_{t-1}, runs in extra blocks of the reduce launch and is not part of this kernel.)
            # As implemented the kernel may read rows of counts instead of (term,count) pairs: 16 SL slots of 2 or 4 bytes per document.
            geo = model.geometry()
            # single-step build since round 4: Elntheta_t / exp(Elntheta_t) are formed inside the PREVIOUS pass's merged launch; this kernel reads
            # exp(Elntheta_t) and writes gamma_{t+1} -- 2 x 8 B x K per document; the gamma_t read and the Elntheta_t write are the other launch's
            # (the first pass of every API call forms its own prologue and moves 24 B x K: the launches of a K-step call are a mix)
            per_doc = ((steps - 1) * 16.0 + 24.0) / steps if geo.get("prologue_moved") else 24.0
            algo_bytes = 8.0 * nnz + per_doc * K * D
            row_bytes = geo.get("row_bytes", 0)
            impl_bytes = (float(row_bytes) * D if row_bytes else 8.0 * nnz) + per_doc * K * D
            kname = ("k_lda_estep_dense%s<%d,%d> (rows of counts)" % ("32" if geo["dense"] == 2 else "", geo["KP"], geo["SL"] // (2 if geo["dense"] == 2 else 1))) if geo["dense"] else \
                    ("k_lda_estep<%d,%d,..,%s>" % (geo["KP"], geo["L"], "single step" if geo["single_step"] else "grid stride"))
            achieved = algo_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
            achieved_impl = impl_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
            res["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": kname, "launches": n_launch, "avg_us": avg_us,
                               "algorithmic_bytes_per_launch": algo_bytes,
                               "algorithmic_bytes_as_implemented": impl_bytes, "achieved_as_implemented": achieved_impl,
                               "frac_as_implemented": achieved_impl / HBM_PEAK_GBS,
                               "bytes_model": "SURVEY 8d: 8 B x nnz + %.1f B x K x D%s.  As implemented: %s + %.1f B x K x D" %
                                              (per_doc, " (16 B: exp(Elntheta_t) read, gamma_{t+1} written -- the gamma_t read and the Elntheta_t write moved into the "
                                               "previous pass's merged launch with the prologue and are not this kernel's bytes; the first launch of each %d-step call "
                                               "forms its own prologue: 24 B)" % steps if per_doc < 24.0 else "",
                                               ("%d B per document (rows of counts / padded rows)" % row_bytes) if row_bytes else "8 B x nnz", per_doc),
                               "event_span_1_launch_us": span1, "event_span_2_launches_us": span2,
                               "timing": "HIP events on the library's stream around the kernel, in repeats of the timed K steps: span with two "
                                         "back-to-back launches minus span with one (an event pair around a single launch adds ~4 us)"}
            if geo["dense"]:
                # The dense-row build streams, and what binds it is not HBM (1.6 TB/s of real traffic) but vector-f64 ISSUE: per term slot three
                # K-wide fused multiply-adds (normaliser, gamma sums, lambda statistics) = 2 x 3 x K x V flops per document, plus the prologue.
                # The line names that ceiling, as the CTM lines do; the HBM figures the contract's object asks for ride along under "hbm".
                flops = 2.0 * 3.0 * K * V * D
                tf = flops / avg_s / 1e12 if avg_s > 0 else 0.0
                hbm_keys = ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "algorithmic_bytes_as_implemented", "achieved_as_implemented",
                            "frac_as_implemented")
                hbm = {k_: res["roofline"][k_] for k_ in hbm_keys}
                res["roofline"].update({"bound": "f64_valu_issue", "achieved": tf, "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK_TF,
                                        "hbm": hbm,
                                        "f64_valu": {"achieved": tf, "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK_TF, "flops_per_launch": flops,
                                                     "model": "the three K x V contractions of the dense-row sweep as fused multiply-adds: 2 x 3 x K x V x D "
                                                              "(the prologue's digammas / exps and the epilogue are not counted)"}})
            tname = "lda_estep_dense_640k" if (geo["dense"] and D == 640000) else "lda_estep"
            kprefix = ("k_lda_estep_dense<%d, %d," % (geo["KP"], geo["SL"])) if geo["dense"] == 1 else ("k_lda_estep<%d, %d," % (geo["KP"], geo["L"]))
        else:
            st = model.solver_stats()
            MK, M = sum(K), len(K)
            # dominant kernel: the solve phase k_ctm_estep<L,1,..> (update_ν! + update_λ!, the two LD_MMA solves per document).
            # Algorithmic bytes per document: lambda, nu, sumtheta read + lambda, nu written (5 MK doubles), zeta and N_dm read (2 M
            # doubles), two evaluation counters.  Useful f64 work per document, from the evaluation counts the kernel reports:
            # a lambda evaluation = invSigma mat-vec (2 MK^2) + objective/gradient/MMA step algebra (~35 MK), a nu evaluation ~35 MK.
            algo_bytes = ((5 * MK + 2 * M) * 8.0 + 8.0) * D
            flops = st["n_eval_lambda"] * (2.0 * MK * MK + 35.0 * MK) + st["n_eval_nu"] * 35.0 * MK
            achieved = algo_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
            tf = flops / avg_s / 1e12 if avg_s > 0 else 0.0
            # The solve phase moves 1.2 KB per document and spends ~40 LD_MMA evaluations of ~640 issue cycles each on it: it is bound by
            # vector-f64 ISSUE, and the line says so; the HBM figures the contract's object asks for ride along under "hbm".
            hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo_bytes}
            res["roofline"] = {"bound": "f64_valu_issue", "achieved": tf, "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK_TF, "traffic": None,
                               "kernel": "solve phase (update_nu! + update_lambda!): %d lanes per document, %d coordinate(s) per lane" % (model.geometry()["Ls"], max(model.geometry()["cpl"], 1)),
                               "launches": n_launch, "avg_us": avg_us,
                               "algorithmic_bytes_per_launch": algo_bytes, "hbm": hbm,
                               "f64_valu": {"achieved": tf, "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK_TF,
                                            "flops_per_launch": flops,
                                            "model": "useful f64 work: n_eval_lambda x (2 MK^2 + 35 MK) + n_eval_nu x 35 MK, evaluation counts of the last pass",
                                            "mma_evaluations_per_document": (st["n_eval_nu"] + st["n_eval_lambda"]) / max(D, 1)},
                               "timing": "HIP events on the library's stream around the kernel, inside a repeat of the timed K steps"}
            res["n_capped"] = st["n_capped"]
            tname = "ctm_solve_cfg%d" % cfg_id
            kprefix = "k_ctm_solve_cpl<%d, %d," % (MK, model.geometry()["Ls"])
        tr, tfile, why_not = (None, None, "counters are collected at N = 1 on the configuration's own corpus size")
        if world == 1 and (D == cfg["docs"] or tname == "lda_estep_dense_640k"):
            tr, tfile, why_not = load_pmc(tname, cfg["model"] if cfg["model"] == "lda" else "ctm", kprefix)
        res["build"] = {"mmm_version": int(lib.mmm_version()), "csrc_sha16": csrc_sha16(cfg["model"] if cfg["model"] == "lda" else "ctm")}
        if not tr:
            res["roofline"]["traffic_not_attached"] = why_not
        if tr:
            res["roofline"]["traffic"] = tr["hbm_bytes_per_launch_gfx950_corrected"]
            res["roofline"]["traffic_over_algorithmic"] = tr["hbm_bytes_per_launch_gfx950_corrected"] / algo_bytes
            res["roofline"]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed as profiles/" + tfile
            cnt = tr.get("counters_per_launch", {})
            if cnt.get("SQ_INSTS_VALU") and avg_s > 0:
                # the ceiling that binds these kernels (a cache-resident working set in LDA, a compute-heavy solve in the CTMs): vector
                # instruction ISSUE.  A wave's VALU instruction holds its SIMD for 4 cycles, a v_rcp / v_rsq / v_sqrt_f64 for 16 (measured:
                # profiles/experiments/r03_f64_rates.txt); SQ_ACTIVE_INST_VALU counts exactly those 4-cycle units.  The floor is that time
                # spread evenly over every SIMD at the nominal clock (under a chip-wide f64 load the clock itself drops to ~2.1 GHz,
                # profiles/experiments/r03_f64_clock.txt, so the true floor is ~12 % higher).
                inst = float(cnt["SQ_INSTS_VALU"])
                quads = float(cnt.get("SQ_ACTIVE_INST_VALU") or inst)
                floor_us = quads * 4.0 / N_SIMD / SCLK_HZ * 1e6
                blk = {"valu_wave_instructions_per_launch": inst, "valu_issue_quad_cycles_per_launch": quads, "issue_floor_us": floor_us,
                       "frac": floor_us / avg_us,
                       "unit": "fraction of the kernel's duration that the vector pipes need (4 cycles per wave instruction, 16 per f64 rcp / rsq / sqrt), "
                               "evenly spread over %d SIMDs at %.1f GHz" % (N_SIMD, SCLK_HZ / 1e9),
                       "source": "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU of profiles/" + tfile}
                if cfg["model"] == "lda" and "f64_valu" in res["roofline"]:
                    res["roofline"]["f64_valu"]["issue"] = blk
                elif cfg["model"] == "lda":
                    res["roofline"]["f64_valu"] = blk
                else:
                    blk["note"] = ("the counters are averages over the full-size launches of the profiled command (its first ~40 passes, whose solves need "
                                   "more evaluations than the later passes timed here), so this fraction overstates the busy share of the timed launches; "
                                   "at equal passes the pipes are ~80 % busy (DESIGN.md section 4.2)")
                    res["roofline"]["f64_valu"]["issue"] = blk
        if world == 1 and probe:
            if cfg["model"] == "lda":
                res.update(parity_probe_lda(pkg, K, 0.1, 0.1, V, seed + 7))
                if D <= 20000:      # ... and the benchmark's OWN corpus against the oracle: 12 passes of the 1-thread C port cost ~0.5 s at 10k documents
                    res["parity_on_the_bench_corpus"] = parity_full_lda(pkg, ctx, X, init, K, V)
            else:
                res.update(parity_probe_ctm(pkg, cfg, seed + 7))
        if world == 1:
            if cpu_baseline:
                kw = {} if cpu_target_s is None else {"target_s": cpu_target_s}
                res["cpu_baseline"] = cpu_baseline_lda(X, init, K, 0.1, 0.1, **kw) if cfg["model"] == "lda" else cpu_baseline_ctm(cfg, X, init, **kw)
                res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
    model.close()
    # ---- the same shard on every other transport the job can run (N > 1): timed regions only
    if every_transport and (world > 1 or len(transports_avail) > 1):
        entries = {}
        if rank == 0:
            entries[primary] = {"value": res["value"], "ms_per_step": res["ms_per_step"], "ms_per_step_min": res["ms_per_step_min"],
                                "allreduce_per_rank": transports, "roofline_frac_per_gpu": res["roofline"]["frac"],
                                "hbm_roofline_frac_per_gpu": res["roofline"].get("hbm", res["roofline"])["frac"]}
        for name, switch in transports_avail[1:]:
            switch()
            m2, _, run2 = make_model()
            reg2 = timed_regions(run2)
            if cfg["model"] == "lda":      # (flushes the lagged ll: an exchange every rank must make)
                ll = np.zeros(1); n = pkg._lib.C.c_int()
                pkg._lib.check(lib.mmm_lda_ll_history(m2._h, ll.ctypes.data, 1, pkg._lib.C.byref(n)), ctx.h)
            tr2 = env.allgather_obj(ctx.transport)
            m2.close()
            d2 = float(np.median(reg2))
            note("transport %s: %.4f ms per step" % (name, d2 / steps * 1e3))
            if rank == 0:
                entries[name] = {"value": sum(docs_per_rank) * steps / d2, "ms_per_step": d2 / steps * 1e3, "ms_per_step_min": min(reg2) / steps * 1e3,
                                 "allreduce_per_rank": tr2,
                                 "roofline_frac_per_gpu": res["roofline"]["frac"], "hbm_roofline_frac_per_gpu": res["roofline"].get("hbm", res["roofline"])["frac"],
                                 "note": "timed regions only; the dominant kernel is the same launch as on the primary transport (its fraction is repeated)"}
        env.restore_transport(primary)
        if rank == 0:
            for name, why in env.transport_unavailable.items():
                entries[name] = {"unavailable": why}
            res["transports"] = entries
    return res


def restart_sweep_brca(env, restarts=256, cpu_restarts=3):
    """The package's real workload (scripts/run_mmctm.jl:77-182, `fit_model`): R randomly initialised MMCTM [7,7] fits of the shipped BRCA-EU
    SNV + SV tables (560 documents; stage 1: maxiter 1000, tol 1e-4 -- here the R replicas of ONE batch handle, the reference uses `pmap`
    over worker processes), the best model per modality, and the seeded stage-2 fit (tol 1e-5).  Reported: wall time and fits/s of the whole
    flow through multimodalmusig_jl_amd.restarts, a CPU baseline (the index-order C oracle, one thread, `cpu_restarts` of the same stage-1
    fits) and the parity probe (a restart of a batch is bitwise the single fit from its initialisation)."""
    import numpy as np
    pkg, ctx = env.pkg, env.ctx
    from multimodalmusig_jl_amd import restarts as rs
    from oracle import oracle as orc
    gold = os.path.join(ROOT, "tests", "golden")
    _, samples, snv = pkg.read_counts_tsv(os.path.join(gold, "brca-eu_snv_counts.tsv"))
    _, _, sv = pkg.read_counts_tsv(os.path.join(gold, "brca-eu_sv_counts.tsv"))
    X = pkg.format_counts_mmctm([{s_: snv[:, i] for i, s_ in enumerate(samples)}, {s_: sv[:, i] for i, s_ in enumerate(samples)}], samples)
    K, V, alpha = [7, 7], [96, 48], [0.1, 0.1]
    seeds = np.random.default_rng(1).integers(1, 2 ** 62, size=int(restarts))
    pkg.MMCTM(K, alpha, V, X, seed=0, ctx=ctx).close()                       # module / kernel load outside the timing
    t0 = time.perf_counter()
    g, best, all_ll = rs.fit_seed_models(X, K, alpha, V, seeds, ctx=ctx)
    t1 = time.perf_counter()
    model = rs.seed_and_fit_restart(X, K, alpha, V, g, ctx=ctx)
    t2 = time.perf_counter()
    res = {"workload": "scripts/run_mmctm.jl fit_model: %d restarts x MMCTM K=[7,7] on the shipped BRCA-EU SNV + SV tables (560 documents), stage 1 maxiter 1000 "
                       "tol 1e-4, per-modality selection, seeded stage 2 tol 1e-5" % restarts,
           "restarts": int(restarts), "stage1_s": t1 - t0, "stage2_s": t2 - t1, "wall_s": t2 - t0, "fits_per_s": (restarts + 1) / (t2 - t0),
           "stage1_best_ll": [float(x) for x in best], "stage2_ll": [float(x) for x in model.ll], "stage2_converged": bool(model.converged),
           "stage2_elbo": float(model.elbo)}
    model.close()
    # parity probe: replica r of a batch == the single fit from the same initialisation, bit for bit
    g0 = []
    for s_ in seeds[:4]:
        rng = np.random.default_rng(int(s_))
        g0.append([rng.integers(1, 101, size=(K[m], V[m])).astype(np.float64) for m in range(2)])
    batch = pkg.MMCTM(K, alpha, V, X, γ0=g0, restarts=4, ctx=ctx)
    hists = pkg.fit_restarts(batch, maxiter=1000, tol=1e-4)
    batch.select(2)
    single = pkg.MMCTM(K, alpha, V, X, γ0=g0[2], ctx=ctx)
    ll1 = pkg.fit(single, maxiter=1000, tol=1e-4, verbose=False)
    res["parity"] = {"replica_2_of_a_4_restart_batch_bitwise_equals_the_single_fit": bool(
        np.array_equal(batch._get("gamma"), single._get("gamma")) and np.array_equal(batch.lam_matrix(), single.lam_matrix()) and
        np.array_equal(np.asarray(hists[2]), np.asarray(ll1))), "passes_of_that_fit": int(len(ll1))}
    res["stage1_passes_mean"] = float(np.mean(batch.restart_iters))
    batch.close(); single.close()
    # CPU baseline: the same stage-1 fits by the one-thread C oracle
    tc = time.perf_counter()
    npass = 0
    for r in range(cpu_restarts):
        o = orc.CtmOracle(K, alpha, X, V=V, gamma0=np.concatenate([x.ravel() for x in g0[r]]))
        npass += len(o.fit(maxiter=1000, tol=1e-4))
    dtc = time.perf_counter() - tc
    res["cpu_baseline"] = {"value": cpu_restarts / dtc, "unit": "fits/s", "cores": 1, "kind": "port",
                           "sample": "%d of the same stage-1 fits (maxiter 1000, tol 1e-4; %d passes in all), %.1f s, single thread (C oracle, index-order variant; the "
                                     "reference's own sweep is `pmap` over Julia worker processes)" % (cpu_restarts, npass, dtc)}
    res["speedup_vs_cpu_baseline"] = res["fits_per_s"] / res["cpu_baseline"]["value"]
    return res


def shard_proxy(env, full, shards=(2, 4, 8), cfgs=(2, 4, 5)):
    """What ONE GPU does in an N-GPU strong run of BASELINE's configurations, measured on the one GPU at hand: rank 0's nnz-balanced shard of
    the configuration's corpus (D/2, D/4, D/8 documents) through the same timed regions, no communicator.  T(D) / T(D/N) is an UPPER BOUND
    on the strong-scaling speed-up of that configuration at N GPUs -- the exchange (one all-reduce of 8-20 KB per iteration, folded into a
    launch for LDA) and the slowest rank are not in it -- and NOT a measurement of it.  `full`: {cfg: result of run_config at full size}."""
    out = {"what": "per-GPU work of an N-GPU strong run, measured on one GPU: rank 0's shard (shard_documents(X, N, 0)) of the configuration's corpus, "
                   "no exchange; speedup_upper_bound = ms_per_step(D) / ms_per_step(D/N) -- an upper bound on strong scaling, not a measurement of it"}
    for c in cfgs:
        cfg = CONFIGS[c]
        seed = 20261003 + (1 if c == 2 else c)
        lda = cfg["model"] == "lda"
        rows = []
        f = full.get(c)
        if f and "ms_per_step" in f:
            rows.append({"n": 1, "docs": f["config"]["docs_rank0"], "ms_per_step": f["ms_per_step"], "ms_per_step_min": f["ms_per_step_min"],
                         "kernel_us": f["iteration"]["kernel_us"]})
        for n in shards:
            try:
                r = run_config(env, c, "weak", 20 if lda else 10, 5 if lda else 2, 5 if lda else 3, 0, False, probe=False, proxy_shard=n,
                               corpus=make_corpus(c, cfg["docs"], seed))
                row = {"n": n, "docs": r["config"]["docs_rank0"], "ms_per_step": r["ms_per_step"], "ms_per_step_min": r["ms_per_step_min"],
                       "kernel_us": r["iteration"]["kernel_us"]}
                if not lda:
                    row["mma_evaluations_per_document"] = r["roofline"]["f64_valu"]["mma_evaluations_per_document"]
                if rows and rows[0]["n"] == 1:
                    row["speedup_upper_bound"] = rows[0]["ms_per_step"] / r["ms_per_step"]
                rows.append(row)
            except Exception as e:       # noqa: BLE001
                rows.append({"n": n, "error": "%s: %s" % (type(e).__name__, e)})
        out["cfg%d" % c] = rows
    return out


def compact(r):
    """the keys of a nested entry (strong / weak variants inside the one line)"""
    if r is None:
        return None
    keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "repeats", "scaling", "dtype", "transports", "ll_last",
            "n_capped", "error")
    out = {k: r[k] for k in keep if k in r}
    c = r.get("config", {})
    out["config"] = {k: c[k] for k in ("workload", "docs_total", "docs_per_rank", "sharding", "allreduce", "allreduce_per_rank", "comm_nranks_per_rank", "note") if k in c}
    rf = r.get("roofline", {})
    out["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_us", "algorithmic_bytes_per_launch", "hbm") if k in rf}
    out["roofline"]["per"] = "GPU (rank 0's launch over rank 0's shard)"
    if "iteration" in r:
        out["kernel_us"] = r["iteration"]["kernel_us"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--repeats", type=int, default=9, help="timed regions of --steps iterations each; the median is reported")
    ap.add_argument("--docs", type=int, default=0, help="documents per GPU (weak) / in total (strong); default: the configuration's size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="default invocation only: do not attach configs 4 and 5 under \"also\"")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="self-launched --gpus N: seconds before the ranks are killed")
    ap.add_argument("--lda-build", choices=["auto", "sparse", "dense", "wide"], default="auto", help="mmm_tuning_opts.lda_build of the LDA handles (A/B runs)")
    ap.add_argument("--ctm-build", choices=["auto", "sparse", "dense", "wide"], default="auto", help="mmm_tuning_opts.ctm_build of the CTM handles (A/B runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout)       # never returns; no GPU call has been made by this process
    cfg = CONFIGS[args.config]
    if args.steps is None:
        args.steps = 50 if cfg["model"] == "lda" else 10
    if args.warmup is None:
        args.warmup = 5 if cfg["model"] == "lda" else 2

    env = Env(args.gpus)
    if args.lda_build != "auto" or args.ctm_build != "auto":
        env.ctx.set_tuning(lda_build=args.lda_build, ctm_build=args.ctm_build)
    multi = env.world > 1
    # (MMM_FORCE_RCCL=1 MMM_P2P_ONE_RANK=1 at N = 1: a one-rank communicator with mailboxes -- the transport switch rehearsed on one GPU)
    res = run_config(env, args.config, args.scaling, args.steps, args.warmup, args.repeats, args.docs, not args.no_cpu_baseline,
                     every_transport=multi or env.has_comm)
    # the other two throughput configurations of BASELINE.json, bounded, on the same ranks, in the same line
    if args.config == 2 and not args.docs and not args.no_also:
        def attempt(fn):
            try:
                return fn()
            except Exception as e:       # noqa: BLE001 -- one GPU: the headline stands on its own; several ranks: a rank that carried on
                if multi:                # alone would leave the others inside a collective -- fail the job instead
                    raise
                return {"error": "%s: %s" % (type(e).__name__, e)}
        also = {}
        if multi and args.scaling == "weak":
            # BASELINE.json's ">= 6x further at 8 GPUs" read literally: the ONE 10k-document corpus, sharded
            r = attempt(lambda: run_config(env, 2, "strong", args.steps, args.warmup, args.repeats, 0, False, every_transport=True))
            if env.rank == 0:
                res["strong"] = compact(r)
        for c in (4, 5):
            t0 = time.perf_counter()
            if multi:
                # BASELINE configs[3] / configs[4] ARE sharded corpora ("50k docs ... sharded 8x", "100k docs ... 8x"): strong scaling is the
                # entry, the weak variant (every rank its own corpus of that size) rides beside it
                r = attempt(lambda: run_config(env, c, "strong", 10, 2, 5, 0, False, every_transport=True))
                rw = attempt(lambda: run_config(env, c, "weak", 10, 2, 3, 0, False))
                if env.rank == 0:
                    r["weak"] = compact(rw)
            else:
                r = attempt(lambda: run_config(env, c, args.scaling, 10, 2, 5, 0, not args.no_cpu_baseline, cpu_target_s=8.0))
            if env.rank == 0:
                r["wall_s_including_corpus_generation_and_cpu_baseline"] = time.perf_counter() - t0
                also["cfg%d" % c] = r
        # ... and the headline model at a size where the kernels reach their steady state (640k documents per GPU: the roofline figures of
        # the dense-row E-step build); bounded: ~10 s including corpus generation
        t0 = time.perf_counter()
        r = attempt(lambda: run_config(env, 2, "weak", 30, 5, 5, 640000, False))
        if env.rank == 0:
            r["wall_s_including_corpus_generation"] = time.perf_counter() - t0
            also["lda_640k_docs"] = r
        if not multi:      # the package's real workload: the restart sweep of the reference's script on the shipped tables (independent fits: no collective)
            t0 = time.perf_counter()
            r = attempt(lambda: restart_sweep_brca(env))
            r["wall_s_including_cpu_baseline"] = time.perf_counter() - t0
            also["restart_sweep_brca"] = r
            full = {2: res, 4: also.get("cfg4"), 5: also.get("cfg5")}
            t0 = time.perf_counter()
            r = attempt(lambda: shard_proxy(env, {k: v for k, v in full.items() if v and "error" not in v}))
            r["wall_s"] = time.perf_counter() - t0
            also["shard_proxy"] = r
        if env.rank == 0:
            res["also"] = also
    if env.rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
    env.close()


if __name__ == "__main__":
    main()
