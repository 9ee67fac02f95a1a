"""Frozen-topic inference of the reference over the HIP backend: `transform`, `fit_heldout`, `predict_modality_η`
(LDA.jl:226-295, MMCTM.jl:496-634, IMMCTM.jl:468-545).  Each builds a fresh model on the new documents, copies the trained
globals the reference copies (and only those), and runs the frozen-topic passes on the GPU (`mmm_lda_infer` / `mmm_ctm_infer`).

Reference behaviour kept as is:
  * `transform(::MMCTM)` has `tol = 1e4` as its default (MMCTM.jl:512), so with defaults it stops at pass 11; and with
    `fit_gaussian=false` it copies μ and Σ but NOT invΣ, which stays the identity of the constructor (MMCTM.jl:518-521).
  * a failed convergence calls `warn(...)`, which does not exist in Julia >= 1.0 (LDA.jl:259, MMCTM.jl:624): here it is a
    Python warning.
One deviation: `predict_modality_η(::MMCTM)` evaluates its convergence log-likelihood on `props`/`ϕ` that were never
initialised (MMCTM.jl:47-50,80,612); the IMMCTM twin recomputes them (IMMCTM.jl:408-428), and that is what both do here.
"""
import ctypes as C
import warnings

import numpy as np

from ._lib import check, lib
from .ctm import IMMCTM, MMCTM
from .models import ILDA, LDA, calculate_elbo

UNSMOOTHED, FIT_GAUSSIAN = 1, 2      # MMM_INFER_* of include/mmmusig.h


def _lda_infer(model, unsmoothed, maxiter, tol, verbose):
    ll = np.zeros(maxiter); ni = C.c_int(); cv = C.c_int()
    check(lib().mmm_lda_infer(model._h, int(unsmoothed), int(maxiter), float(tol), ll.ctypes.data, C.byref(ni), C.byref(cv)),
          model.ctx.h, "mmm_lda_infer")
    hist = ll[:ni.value].copy()
    if verbose:
        for i, v in enumerate(hist):
            print("%d\tLog-likelihood: %r" % (i + 1, v))
    model.converged = bool(cv.value)
    return hist


def _ctm_infer(model, flags, maxiter, tol, verbose):
    M = model.M
    ll = np.zeros(maxiter * M); ni = C.c_int(); cv = C.c_int()
    check(lib().mmm_ctm_infer(model._h, int(flags), int(maxiter), float(tol), ll.ctypes.data, C.byref(ni), C.byref(cv)),
          model.ctx.h, "mmm_ctm_infer")
    hist = ll[:ni.value * M].reshape(ni.value, M).copy()
    if verbose:
        for i, v in enumerate(hist):
            print("%d\tLog-likelihoods: %s" % (i + 1, ", ".join(repr(float(x)) for x in v)))
    model.converged = bool(cv.value)
    return hist


def _new_like(model, X, K=None, α=None, mods=None, seed=None):
    """A constructor-state model of the same family on documents X (optionally restricted to the modalities `mods`)."""
    if isinstance(model, IMMCTM):
        mods = range(model.M) if mods is None else mods
        return IMMCTM([model.K[m] for m in mods], [model.α[m] for m in mods], [model.features[m] for m in mods], X, seed=seed, ctx=model.ctx,
                      xtol_rule=model._opts.xtol_rule)
    mods = range(model.M) if mods is None else mods
    return MMCTM([model.K[m] for m in mods], [float(model.α[m]) for m in mods], [model.V[m] for m in mods], X, seed=seed, ctx=model.ctx,
                 xtol_rule=model._opts.xtol_rule)


def transform(model, X, maxiter=1000, tol=None, fit_gaussian=False, verbose=False, seed=None):
    """`transform(model, X)`.  LDA (LDA.jl:233-263): returns θ (K x D) of the new documents under the trained β.
    MMCTM (MMCTM.jl:511-552): returns the new model (λ, ν, θ, props, ll of the new documents under the trained ϕ)."""
    if isinstance(model, ILDA):
        # ILDA.jl:289-318 builds `LDA(model.K, model.α, model.η, X)` with a vector η: a MethodError upstream
        raise TypeError("transform(::ILDA) is a MethodError in the reference (ILDA.jl:293); use fit_heldout")
    if isinstance(model, LDA):
        tol = 1e-4 if tol is None else tol
        new = LDA(model.K, model.α, model.η, model.V, X, seed=seed, ctx=model.ctx)
        new.β = model.β                                              # LDA.jl:237
        _lda_infer(new, True, maxiter, tol, verbose)
        if not new.converged:
            warnings.warn("transform did not converge")              # LDA.jl:258-260
        θ = new.θ
        new.close()
        return θ
    if isinstance(model, IMMCTM):
        raise TypeError("transform is not defined for IMMCTM (IMMCTM.jl has no such method)")
    tol = 1e4 if tol is None else tol                                # MMCTM.jl:512 (sic)
    new = _new_like(model, X, seed=seed)
    new._set("phi", model._get("phi"))                               # MMCTM.jl:516
    if not fit_gaussian:
        new.μ = model.μ; new.Σ = model.Σ                             # MMCTM.jl:518-521 (invΣ stays I)
    hist = _ctm_infer(new, UNSMOOTHED | (FIT_GAUSSIAN if fit_gaussian else 0), maxiter, tol, verbose)
    new.ll = hist[-1].copy()                                         # MMCTM.jl:549
    new.ll_history = hist
    return new


def fit_heldout(Xheldout, model, maxiter=100, verbose=False, seed=None):
    """`fit_heldout(Xheldout, model)` -- LDA.jl:265-295, MMCTM.jl:554-586, IMMCTM.jl:468-497: the variational document
    parameters of held-out documents under the trained topics (smoothed update_ϕ!/update_θ!), tol = 1e-4."""
    if isinstance(model, ILDA):                                       # ILDA.jl:320-353
        new = ILDA(model.K, model.α, model.η, model.features, Xheldout, seed=seed, ctx=model.ctx)
        new._set("ilambda", model._get("ilambda")); new._set("ibeta", model._get("ibeta")); new._set("iElnbeta", model._get("iElnbeta"))
        hist = _lda_infer(new, False, maxiter, 1e-4, verbose)
        new.elbo = calculate_elbo(new)
        new.ll = float(hist[-1])
        new.ll_history = hist
        return new
    if isinstance(model, LDA):
        new = LDA(model.K, model.α, model.η, model.V, Xheldout, seed=seed, ctx=model.ctx)
        new.λ = model.λ; new.β = model.β; new.Elnβ = model.Elnβ      # LDA.jl:269-271
        hist = _lda_infer(new, False, maxiter, 1e-4, verbose)
        new.elbo = calculate_elbo(new)                               # LDA.jl:291
        new.ll = float(hist[-1])
        new.ll_history = hist
        return new
    new = _new_like(model, Xheldout, seed=seed)
    new.μ = model.μ; new.Σ = model.Σ; new.invΣ = model.invΣ          # MMCTM.jl:558-560
    new._set("gamma", model._get("gamma")); new._set("Elnphi", model._get("Elnphi"))   # :561-562
    if not isinstance(model, IMMCTM):
        new._set("phi", model._get("phi"))                           # :563
    new.ll_history = _ctm_infer(new, 0, maxiter, 1e-4, verbose)
    return new


def predict_modality_η(Xobs, m, model, maxiter=100, verbose=False, seed=None):
    """`predict_modality_η(Xobs, m, model)` -- MMCTM.jl:588-634 / IMMCTM.jl:499-545: conditional mean of the unobserved
    modality m's η given the documents' other modalities.  `m` is 0-based here; Xobs[d] lists the observed modalities in order.
    Returns a list over documents of length-K[m] vectors."""
    obsM = [i for i in range(model.M) if i != m]
    koff = np.concatenate([[0], np.cumsum(model.K)])
    unobs = np.arange(koff[m], koff[m + 1])
    obs = np.setdiff1d(np.arange(koff[-1]), unobs)
    μ, Σ, invΣ = model.μ, model.Σ, model.invΣ
    new = _new_like(model, Xobs, mods=obsM, seed=seed)
    new.μ = μ[obs]; new.Σ = Σ[np.ix_(obs, obs)]; new.invΣ = invΣ[np.ix_(obs, obs)]     # MMCTM.jl:597-599 (sub-block of invΣ, sic)
    goff = model._mgoff if isinstance(model, IMMCTM) else model._goff                     # flat [m][k][...] topic layout

    def sub(name):
        flat = model._get(name)
        return np.concatenate([flat[goff[i]:goff[i + 1]] for i in obsM])
    new._set("gamma", sub("gamma")); new._set("Elnphi", sub("Elnphi"))                    # :600-601
    if not isinstance(model, IMMCTM):
        new._set("phi", sub("phi"))      # the reference leaves the constructor's ϕ here (module docstring)
    _ctm_infer(new, 0, maxiter, 1e-4, verbose)
    if not new.converged:
        warnings.warn("model not converged.")                                              # :623-625
    A = Σ[np.ix_(unobs, obs)] @ invΣ[np.ix_(obs, obs)]                                   # :627-633
    lam = new.lam_matrix()
    η = [μ[unobs] + A @ (lam[d] - μ[obs]) for d in range(new.D)]
    new.close()
    return η
