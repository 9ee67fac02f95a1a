"""multimodalmusig.jl_amd -- MI355X (gfx950) backend for the variational-EM hot path of MultiModalMuSig.jl.

Exports mirror the reference module (src/MultiModalMuSig.jl:9): IMMCTM, MMCTM, LDA, fit (`fit!`),
format_counts_lda / _ctm / _mmctm.  ILDA rides on the LDA kernels with feature-folded topic tables.
"""
from . import _lib
from ._lib import Context, MmmError, build, comm_unique_id, default_context, lib
from .models import (ILDA, LDA, calculate_elbo, calculate_loglikelihood, fit, fit_bang, update_Elnβ, update_Elnθ, update_β, update_γ, update_θ,
                     update_λ, update_ϕ)
from .ctm import (calculate_docmodality_loglikelihood, calculate_modality_loglikelihood, calculate_Ndivζ, calculate_sumθ, α_objective, λ_objective, ν_objective,
                  IMMCTM, MMCTM, calculate_loglikelihoods, fit_restarts, fitdoc, pick_optimal_modality_models, update_Elnϕ, update_props, update_α, update_Σ, update_ζ, update_μ,
                  update_ν)
from .inference import fit_heldout, predict_modality_η, transform
from .utils import (format_counts_ctm, format_counts_lda, format_counts_mmctm, make_count_matrix, pack_lda,
                    pack_mm, read_counts_tsv, shard_documents)

__all__ = ["ILDA", "IMMCTM", "MMCTM", "LDA", "fit", "fit_bang", "format_counts_lda", "format_counts_ctm", "format_counts_mmctm", "Context",
           "MmmError", "build", "fit_restarts", "transform", "fit_heldout", "predict_modality_η"]
