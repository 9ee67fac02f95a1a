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
