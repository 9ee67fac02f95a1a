"""Host logic of the restart driver (scripts/run_mmctm.jl:86-95,136-147): model selection rules (no GPU needed)."""
import numpy as np

import mmm_pkg

mmm_pkg.load()
from multimodalmusig_jl_amd import restarts as rs  # noqa: E402


def test_dense_rank_matches_statsbase_semantics():
    # StatsBase.denserank([10, 20, 10, 30]) == [1, 2, 1, 3]
    assert rs.dense_rank([10, 20, 10, 30]).tolist() == [1, 2, 1, 3]
    assert rs.dense_rank([3.5]).tolist() == [1]
    assert rs.dense_rank([-1.0, -1.0, -2.0]).tolist() == [2, 2, 1]


def test_pick_optimal_model_ranks_abs_loglik_per_modality():
    # run_mmctm.jl:136-147: rank |ll| per modality (smaller is better), pick the lowest mean rank; first wins ties (findmin)
    ll = np.array([[-3.0, -2.5], [-2.9, -2.6], [-3.1, -2.4]])
    # ranks: m1: [2,1,3]; m2: [2,3,1] -> means [2, 2, 2] -> first
    assert rs.pick_optimal_model(ll) == 0
    ll = np.array([[-3.0, -2.5], [-2.9, -2.45], [-3.1, -2.4]])
    # m1: [2,1,3]; m2: [3,2,1] -> means [2.5, 1.5, 2]
    assert rs.pick_optimal_model(ll) == 1


def test_pick_optimal_modality_models_is_argmax_per_modality():
    class Stub:
        restart_ll = np.array([[-3.0, -2.5], [-2.9, -2.6], [-3.1, -2.4]])
    from multimodalmusig_jl_amd.ctm import pick_optimal_modality_models
    assert pick_optimal_modality_models(Stub()) == [1, 2]      # findmax(ll; dims=1), run_mmctm.jl:94


def test_restarts_dealt_over_ranks_merge(monkeypatch):
    """fit_seed_models with nranks > 1: every rank fits seeds[r::nranks]; winners per modality and the ll matrix are merged
    identically on all ranks.  The single-rank fit is stubbed (no GPU here): ll of seed s = (-s, -100 + s)."""
    calls = {}

    def fake_single(counts, K, α, V, seeds, **kw):
        seeds = list(seeds)
        ll = np.array([[-float(s), -100.0 + s] for s in seeds])
        g = [np.full((K[0], V[0]), float(seeds[int(np.argmax(ll[:, 0]))])), np.full((K[1], V[1]), float(seeds[int(np.argmax(ll[:, 1]))]))]
        return g, ll.max(axis=0), ll

    real = rs.fit_seed_models

    def dispatch(counts, K, α, V, seeds, rank=0, nranks=1, allgather=None, **kw):
        if nranks == 1:
            return fake_single(counts, K, α, V, seeds, **kw)
        return real(counts, K, α, V, seeds, rank=rank, nranks=nranks, allgather=allgather, **kw)

    monkeypatch.setattr(rs, "fit_seed_models", dispatch)
    seeds = [3, 9, 4, 7, 5]
    K, V = [2, 1], [3, 2]
    parts = [fake_single(None, K, None, V, seeds[r::2]) for r in range(2)]
    for rank in range(2):
        g, ll, allm = real(None, K, None, V, seeds, rank=rank, nranks=2, allgather=lambda obj: parts)
        assert ll.tolist() == [-3.0, -91.0]                      # best of modality 0: seed 3; of modality 1: seed 9
        assert g[0][0, 0] == 3.0 and g[1][0, 0] == 9.0
        np.testing.assert_array_equal(allm, [[-float(s), -100.0 + s] for s in seeds])
