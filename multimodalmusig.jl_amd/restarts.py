"""The restart driver of scripts/run_mmctm.jl (`fit_model`, :163-182) over the batched HIP fit.

The reference fits `restarts` randomly initialised models in worker processes (`pmap(fit_restart, seeds)`, :97-109), keeps
the best model per modality (`pick_optimal_modality_models`, :86-95), builds a model from those per-modality topics
(`seed_and_fit_restart`, :113-134), fits it to tol = 1e-5 and returns it.  Here the restarts of stage 1 are the replicas of
one batch handle: one resident corpus, every kernel launched once per pass for all restarts.

Differences that do not change results:
  * Stage 2 of the reference runs `seed_and_fit_restart` once per seed and ranks the outcomes (`pick_optimal_model`,
    :136-147).  Every one of those fits starts from the same γ/Elnϕ/ϕ (the seed only feeds the constructor's random γ, which
    is overwritten, and `NLopt.srand`, which the deterministic LD_MMA never reads), so they are identical; it is fitted once.
  * Random γ₀ comes from numpy's generator, not Julia's: the restarts differ from a Julia run's the way two Julia runs with
    different seeds differ.
"""
import numpy as np

from .ctm import MMCTM, fit_restarts, pick_optimal_modality_models
from .models import fit


def dense_rank(x):
    """StatsBase.denserank: 1-based rank, equal values share a rank, no gaps."""
    _, inv = np.unique(np.asarray(x), return_inverse=True)
    return inv + 1


def pick_optimal_model(ll):
    """Index of the model with the lowest mean dense rank of |ll| over the modalities -- run_mmctm.jl:136-147."""
    ll = np.atleast_2d(np.asarray(ll, dtype=np.float64))
    ranks = np.stack([dense_rank(np.abs(ll[:, m])) for m in range(ll.shape[1])], axis=1).astype(np.float64)
    return int(np.argmin(ranks.mean(axis=1)))


def fit_seed_models(counts, K, α, V, seeds, batch_size=None, ctx=None, maxiter=1000, tol=1e-4, rank=0, nranks=1, allgather=None, **kw):
    """Stage 1 (run_mmctm.jl:97-109): one restart per seed.  Returns (γ of the best restart per modality -- a list over m of
    [K_m, V_m] arrays --, their final log-likelihoods, the [len(seeds), M] matrix of all final log-likelihoods).

    Several GPUs: restarts are independent, so they are simply dealt out -- rank r of nranks fits seeds[r::nranks] on its own
    context (one WITHOUT a communicator: every rank holds the whole corpus) and the per-modality winners are merged with the
    host's `allgather(obj) -> list of every rank's obj` (torch.distributed.all_gather_object, MPI, ...).  No device collective."""
    seeds = [int(s) for s in seeds]
    if nranks > 1:
        if allgather is None:
            raise ValueError("nranks > 1 needs an allgather callable")
        mine = seeds[rank::nranks]
        g, ll, all_ll = fit_seed_models(counts, K, α, V, mine, batch_size=batch_size, ctx=ctx, maxiter=maxiter, tol=tol, **kw) if mine else (
            [None] * len(K), np.full(len(K), -np.inf), np.zeros((0, len(K))))
        parts = allgather((g, ll, all_ll))
        M = len(K)
        best_gamma, best_ll = [None] * M, np.full(M, -np.inf)
        for pg, pl, _ in parts:                      # rank order: ties go to the lowest rank on every rank alike
            for m in range(M):
                if pg[m] is not None and pl[m] > best_ll[m]:
                    best_ll[m], best_gamma[m] = pl[m], pg[m]
        merged = np.zeros((len(seeds), M))
        for r, (_, _, pa) in enumerate(parts):
            merged[r::nranks][:len(pa)] = pa
        return best_gamma, best_ll, merged
    R = len(seeds)
    bs = R if not batch_size else int(batch_size)
    M = len(K)
    best_ll = np.full(M, -np.inf)
    best_gamma = [None] * M
    all_ll = np.zeros((R, M))
    for b0 in range(0, R, bs):
        chunk = seeds[b0:b0 + bs]
        g0 = []
        for s in chunk:
            rng = np.random.default_rng(s)
            g0.append([rng.integers(1, 101, size=(K[m], V[m])).astype(np.float64) for m in range(M)])   # MMCTM.jl:60-63
        model = MMCTM(K, α, V, counts, γ0=g0, restarts=len(chunk), ctx=ctx, **kw)
        fit_restarts(model, maxiter=maxiter, tol=tol)
        all_ll[b0:b0 + len(chunk)] = model.restart_ll
        opt = pick_optimal_modality_models(model)
        for m in range(M):
            if model.restart_ll[opt[m], m] > best_ll[m]:
                best_ll[m] = model.restart_ll[opt[m], m]
                model.select(opt[m])
                best_gamma[m] = np.stack([model.γ[m][k] for k in range(K[m])])
        model.close()
    return best_gamma, best_ll, all_ll


def seed_and_fit_restart(counts, K, α, V, opt_gamma, ctx=None, maxiter=1000, tol=1e-5, **kw):
    """Stage 2 (run_mmctm.jl:113-134): a model whose topics of modality m are those of the best stage-1 model for m.
    Constructing from γ₀ = γ_opt gives the same Elnϕ the reference copies over (update_Elnϕ! of the same γ)."""
    model = MMCTM(K, α, V, counts, γ0=opt_gamma, ctx=ctx, **kw)
    fit(model, maxiter=maxiter, tol=tol, verbose=False)
    return model


def fit_model(counts, K, α, V, restarts, seed=0, verbose=False, batch_size=None, ctx=None, rank=0, nranks=1, allgather=None, **kw):
    """`fit_model` of run_mmctm.jl:163-182.  Returns the fitted stage-2 model (fields as MMCTM: ϕ, props, Σ, ll, elbo...).
    With nranks > 1 stage 1 is dealt over the ranks (see fit_seed_models); every rank then fits the same seeded stage 2."""
    seeds = np.random.default_rng(seed).integers(1, 2 ** 62, size=int(restarts))
    opt_gamma, opt_ll, all_ll = fit_seed_models(counts, K, α, V, seeds, batch_size=batch_size, ctx=ctx, rank=rank, nranks=nranks,
                                                allgather=allgather, **kw)
    if verbose:
        print("Modality optimal model log-likelihoods:")
        for m in range(len(K)):
            print("%d: %r" % (m + 1, float(opt_ll[m])))
    model = seed_and_fit_restart(counts, K, α, V, opt_gamma, ctx=ctx, **kw)
    if verbose:
        print("Seeded model log-likelihoods:")
        print(model.ll)
    model.stage1_ll = all_ll
    return model
