"""Host-side mirror of the MultiModalMuSig.jl model API over the HIP backend.

Names, argument meaning and field names follow the reference (`LDA`, `MMCTM`, `IMMCTM`, `fit!`, `update_*!`,
`calculate_*`; Julia's `!` suffix is dropped).  Model state lives in HBM inside the C-ABI handle; the
reference's fields (`model.λ`, `model.ϕ`, `model.θ` ...) are properties that download on read and upload on
assignment, so the parity tests read like the reference's own tests.  There is no CPU implementation here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MmmError, check, lib
from .utils import pack_lda, pack_mm

# field ids of include/mmmusig.h (ASCII keys: Python NFKC-normalises identifiers, so `ϕ` typed as a keyword
# would not equal the string "ϕ")
_F = {"lambda": 0, "Elnbeta": 1, "beta": 2, "gamma": 3, "Elntheta": 4, "theta": 5, "phi": 6, "ilambda": 7, "iElnbeta": 8, "ibeta": 9}


class _PerDocView:
    """List-like view of a per-document field (model.ϕ[d] is K x W_d as in LDA.jl:16); item assignment writes
    through to the device."""

    def __init__(self, getter, setter, D):
        self._get, self._set, self._D = getter, setter, D

    def __len__(self):
        return self._D

    def __getitem__(self, d):
        return self._get(d)

    def __setitem__(self, d, value):
        self._set(d, value)

    def __iter__(self):
        return (self._get(d) for d in range(self._D))


class LDA:
    """`LDA(k, α, η, X)` / `LDA(k, α, η, V, X)` -- LDA.jl:24-67.

    X: list of (W_d x 2) integer matrices `[term(1-based) count]` as produced by `format_counts_lda`.
    The random init `λ = rand(1:100, V, K)` (LDA.jl:36) is drawn with numpy (`seed`) unless `λ0` is given.
    With a multi-rank Context, X is this rank's shard of the documents.
    """

    def __init__(self, k, α, η, *args, λ0=None, seed=None, ctx=None):
        if len(args) == 1:
            V, X = None, args[0]
        elif len(args) == 2:
            V, X = args
        else:
            raise TypeError("LDA(k, α, η, [V,] X)")
        self.K, self.α, self.η, self.X = int(k), float(α), float(η), X
        self.D = len(X)
        self._doc_ptr, self._term, self._count = pack_lda(X)
        if V is None:
            V = int(self._term.max()) + 1 if self._term.size else 0     # LDA.jl:57-66
        self.V = int(V)
        self.N = np.array([int(self._count[self._doc_ptr[d]:self._doc_ptr[d + 1]].sum()) for d in range(self.D)], dtype=np.int64)
        if λ0 is None:
            λ0 = np.random.default_rng(seed).integers(1, 101, size=(self.V, self.K)).astype(np.float64)
        λ0 = np.asarray(λ0, dtype=np.float64)
        if λ0.shape != (self.V, self.K):
            raise ValueError("λ0 must be V x K")
        self.ctx = ctx or _lib.default_context()
        self._h = C.c_void_p()
        tp = self._term.ctypes.data if self._term.size else None
        cp = self._count.ctypes.data if self._count.size else None
        check(lib().mmm_lda_create(self.ctx.h, self.D, self.V, self.K, self.α, self.η, self._doc_ptr, tp, cp,
                                   np.ascontiguousarray(λ0.ravel(order="F")), C.byref(self._h)), self.ctx.h, "mmm_lda_create")
        _lib.track(self)
        self.converged = False
        self.elbo = float("nan")
        self.ll = float("nan")

    # ---- raw field transfer -------------------------------------------------------------------------------
    def _size(self, name):
        VK, KD = self.V * self.K, self.K * self.D
        return {"lambda": VK, "Elnbeta": VK, "beta": VK, "gamma": KD, "Elntheta": KD, "theta": KD,
                "phi": self.K * int(self._doc_ptr[-1])}[name]

    def _get(self, name):
        out = np.empty(self._size(name), dtype=np.float64)
        check(lib().mmm_lda_get(self._h, _F[name], out, out.size), self.ctx.h, "mmm_lda_get(%s)" % name)
        return out

    def _set(self, name, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64)
        check(lib().mmm_lda_set(self._h, _F[name], flat, flat.size), self.ctx.h, "mmm_lda_set(%s)" % name)

    def _mat(self, name, rows, cols):
        return self._get(name).reshape(rows, cols, order="F")

    def _push_hyper(self):
        """`model.α` / `model.η` are plain mutable fields upstream (LDA.jl:7,11; ILDA.jl:8,12): what the caller has assigned goes to the
        device before any function that reads them (mmm_lda_set_hyper does nothing when they are unchanged)."""
        eta = np.ascontiguousarray(np.atleast_1d(np.asarray(self.η, dtype=np.float64)))
        check(lib().mmm_lda_set_hyper(self._h, float(self.α), eta, int(eta.size)), self.ctx.h, "mmm_lda_set_hyper")

    # V x K fields
    λ = property(lambda s: s._mat("lambda", s.V, s.K), lambda s, v: s._set("lambda", np.asarray(v, float).ravel(order="F")))
    Elnβ = property(lambda s: s._mat("Elnbeta", s.V, s.K), lambda s, v: s._set("Elnbeta", np.asarray(v, float).ravel(order="F")))
    β = property(lambda s: s._mat("beta", s.V, s.K), lambda s, v: s._set("beta", np.asarray(v, float).ravel(order="F")))
    # K x D fields
    γ = property(lambda s: s._mat("gamma", s.K, s.D), lambda s, v: s._set("gamma", np.asarray(v, float).ravel(order="F")))
    Elnθ = property(lambda s: s._mat("Elntheta", s.K, s.D), lambda s, v: s._set("Elntheta", np.asarray(v, float).ravel(order="F")))
    θ = property(lambda s: s._mat("theta", s.K, s.D), lambda s, v: s._set("theta", np.asarray(v, float).ravel(order="F")))

    @property
    def ϕ(self):
        def get(d):
            flat = self._get("phi")
            a, b = int(self._doc_ptr[d]), int(self._doc_ptr[d + 1])
            return flat[self.K * a:self.K * b].reshape(b - a, self.K).T.copy()

        def put(d, value):
            flat = self._get("phi")
            a, b = int(self._doc_ptr[d]), int(self._doc_ptr[d + 1])
            value = np.asarray(value, dtype=np.float64)
            if value.shape != (self.K, b - a):
                raise ValueError("ϕ[d] must be K x W_d")
            flat[self.K * a:self.K * b] = value.T.ravel()
            self._set("phi", flat)
        return _PerDocView(get, put, self.D)

    @ϕ.setter
    def ϕ(self, docs):
        flat = np.concatenate([np.asarray(p, dtype=np.float64).T.ravel() for p in docs]) if len(docs) else np.zeros(0)
        self._set("phi", flat)

    def phi_flat(self):
        """ϕ as one [nnz, K] array (document blocks concatenated)."""
        return self._get("phi").reshape(-1, self.K)

    def events(self):
        """Non-fatal events (mmm_lda_events): non-finite values in the log-likelihood history (LDA has no LD_MMA solves)."""
        out = (C.c_int64 * 4)()
        check(lib().mmm_lda_events(self._h, out), self.ctx.h, "events")
        return {"n_capped": int(out[0]), "n_nonfinite": int(out[1]), "n_nonfinite_ll": int(out[2])}

    def geometry(self):
        """E-step build and launch geometry of the handle (mmm_lda_geometry)."""
        g = (C.c_int * 8)()
        check(lib().mmm_lda_geometry(self._h, g), self.ctx.h, "geometry")
        return {"L": g[0], "grid_e": g[1], "waves_e": g[2], "single_step": g[3], "wide": g[4], "dense": g[5], "SL": g[6], "KP": g[7],
                "row_bytes": int(lib().mmm_lda_row_bytes(self._h)), "prologue_moved": int(lib().mmm_lda_prologue_moved(self._h))}

    def close(self):
        if self._h:
            lib().mmm_lda_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _FactorList:
    """`model.λ` / `model.Elnβ` / `model.β` of an ILDA: list over features of J_i x K matrices; item assignment writes through."""

    def __init__(self, model, name):
        self._m, self._name = model, name

    def __len__(self):
        return self._m.I

    def __getitem__(self, i):
        m = self._m
        flat = m._get(self._name)
        return flat[m._ioff[i]:m._ioff[i + 1]].reshape(m.J[i], m.K, order="F").copy()

    def __setitem__(self, i, value):
        m = self._m
        flat = m._get(self._name)
        value = np.asarray(value, dtype=np.float64)
        if value.shape != (m.J[i], m.K):
            raise ValueError("%s[%d] must be J_i x K" % (self._name, i))
        flat[m._ioff[i]:m._ioff[i + 1]] = value.ravel(order="F")
        m._set(self._name, flat)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class ILDA(LDA):
    """`ILDA(k, α, η, features, X)` -- ILDA.jl:25-63: LDA whose topic-term distribution factorises over the I features of a
    term.  features: V x I matrix of 1-based feature values; η: scalar or length-I vector.  `λ0`: list over features of
    J_i x K matrices (default `rand(1:100, J_i, K)`, ILDA.jl:36).  The per-function API and `fit` are LDA's (same C handle)."""

    def __init__(self, k, α, η, features, X, λ0=None, seed=None, ctx=None):
        f = np.asarray(features, dtype=np.int64)
        self.features = f
        self.K, self.α, self.X = int(k), float(α), X
        self.I = int(f.shape[1]); self.V = int(f.shape[0])
        self.J = [int(x) for x in f.max(axis=0)]                          # ILDA.jl:35
        self.η = np.full(self.I, float(η)) if np.ndim(η) == 0 else np.asarray(η, dtype=np.float64).copy()
        self.D = len(X)
        self._doc_ptr, self._term, self._count = pack_lda(X)
        self.N = np.array([int(self._count[self._doc_ptr[d]:self._doc_ptr[d + 1]].sum()) for d in range(self.D)], dtype=np.int64)
        self._ioff = np.concatenate([[0], np.cumsum([j * self.K for j in self.J])]).astype(np.int64)
        if λ0 is None:
            rng = np.random.default_rng(seed)
            λ0 = [rng.integers(1, 101, size=(j, self.K)).astype(np.float64) for j in self.J]
        lam0 = np.concatenate([np.asarray(λ0[i], dtype=np.float64).reshape(self.J[i], self.K).ravel(order="F") for i in range(self.I)])
        self.ctx = ctx or _lib.default_context()
        self._h = C.c_void_p()
        tp = self._term.ctypes.data if self._term.size else None
        cp = self._count.ctypes.data if self._count.size else None
        feat = np.ascontiguousarray((f - 1).T.ravel(), dtype=np.int32)                         # [i*V + v], 0-based
        check(lib().mmm_ilda_create(self.ctx.h, self.D, self.V, self.K, self.α, self.I, np.ascontiguousarray(self.J, dtype=np.int32),
                                    np.ascontiguousarray(self.η), feat, self._doc_ptr, tp, cp, lam0, C.byref(self._h)), self.ctx.h, "mmm_ilda_create")
        _lib.track(self)
        self.converged = False
        self.elbo = float("nan")
        self.ll = float("nan")

    def _size(self, name):
        if name in ("ilambda", "iElnbeta", "ibeta"):
            return int(self._ioff[-1])
        return LDA._size(self, name)

    λ = property(lambda s: _FactorList(s, "ilambda"), lambda s, v: [_FactorList(s, "ilambda").__setitem__(i, x) for i, x in enumerate(v)] and None)
    Elnβ = property(lambda s: _FactorList(s, "iElnbeta"), lambda s, v: [_FactorList(s, "iElnbeta").__setitem__(i, x) for i, x in enumerate(v)] and None)
    β = property(lambda s: _FactorList(s, "ibeta"), lambda s, v: [_FactorList(s, "ibeta").__setitem__(i, x) for i, x in enumerate(v)] and None)


# ---- function API (the reference's free functions on a model; dispatch on the model type like Julia methods) ------------
def _call(model, fn, what):
    check(getattr(lib(), fn)(model._h), model.ctx.h, what)


def update_γ(model):   # LDA.jl:82-90 ; MMCTM.jl:224-242 ; IMMCTM.jl:199-223
    if isinstance(model, LDA):
        model._push_hyper()
        _call(model, "mmm_lda_update_gamma", "update_γ!")
    else:
        from . import ctm
        ctm.update_γ_ctm(model)


def update_ϕ(model):   # LDA.jl:69-76 ; MMCTM.jl:244-250
    if isinstance(model, LDA):
        _call(model, "mmm_lda_update_phi", "update_ϕ!")
    else:
        from . import ctm
        ctm.update_ϕ_ctm(model)


def update_λ(model, d=None):   # LDA.jl:100-108 ; MMCTM.jl:127-143
    if isinstance(model, LDA):
        model._push_hyper()
        _call(model, "mmm_lda_update_lambda", "update_λ!")
    else:
        from . import ctm
        ctm.update_λ_ctm(model, d)


def update_β(model):   # LDA.jl:110-112
    _call(model, "mmm_lda_update_beta", "update_β!")


def update_Elnθ(model):   # LDA.jl:78-80
    _call(model, "mmm_lda_update_Elntheta", "update_Elnθ!")


def update_Elnβ(model):   # LDA.jl:96-98 ; ILDA.jl:96-101
    _call(model, "mmm_lda_update_Elnbeta", "update_Elnβ!")


def update_θ(model, d=None):   # LDA.jl:92-94 ; MMCTM.jl:183-198 ; IMMCTM.jl:152-172
    if isinstance(model, LDA):
        _call(model, "mmm_lda_update_theta", "update_θ!")
    else:
        from . import ctm
        ctm.update_θ_ctm(model, d)


def calculate_loglikelihood(model, θ=None, β=None, ctx=None):
    """calculate_loglikelihood(model) -- LDA.jl:194-196, or the free form calculate_loglikelihood(X, θ, β) -- LDA.jl:174-188 (θ K x D, β V x K)."""
    if θ is not None:
        doc_ptr, term, count = pack_lda(model)
        ctx = ctx or _lib.default_context()
        th = np.ascontiguousarray(np.asarray(θ, dtype=np.float64).T); be = np.ascontiguousarray(np.asarray(β, dtype=np.float64).T)   # [k + K d], [k V + v]
        v = C.c_double()
        check(lib().mmm_mixture_loglik(ctx.h, len(model), th.shape[1], be.shape[1], doc_ptr, term.ctypes.data if term.size else None,
                                       count.ctypes.data if count.size else None, th.ravel(), be.ravel(), C.byref(v)), ctx.h, "calculate_loglikelihood")
        return v.value
    v = C.c_double()
    check(lib().mmm_lda_loglik(model._h, C.byref(v)), model.ctx.h, "calculate_loglikelihood")
    return v.value


def calculate_elbo(model, terms=False):   # LDA.jl:162-172 / MMCTM.jl:372-382
    v = C.c_double(); t = np.zeros(7)
    if isinstance(model, LDA):
        model._push_hyper()
    else:
        model._push_alpha()
    fn = lib().mmm_lda_elbo if isinstance(model, LDA) else lib().mmm_ctm_elbo
    check(fn(model._h, C.byref(v), t.ctypes.data), model.ctx.h, "calculate_elbo")
    return (v.value, t) if terms else v.value


def fit(model, maxiter=None, tol=1e-4, verbose=True, **kw):
    """`fit!(model; maxiter, tol, verbose)` -- LDA.jl:198-224 (maxiter default 1000), MMCTM.jl:457-494 and
    IMMCTM.jl:437-466 (default 100).  Returns the log-likelihood history and sets converged/elbo/ll."""
    if isinstance(model, LDA):
        maxiter = 1000 if maxiter is None else int(maxiter)
        model._push_hyper()
        ll = np.zeros(maxiter); ni = C.c_int(); cv = C.c_int(); el = C.c_double()
        check(lib().mmm_lda_fit(model._h, maxiter, float(tol), ll.ctypes.data, C.byref(ni), C.byref(cv), C.byref(el)),
              model.ctx.h, "fit!(::LDA)")
        hist = ll[:ni.value].copy()
        if verbose:
            for i, v in enumerate(hist):
                print("%d\tLog-likelihood: %r" % (i + 1, v))
        model.converged = bool(cv.value); model.elbo = el.value; model.ll = float(hist[-1])
        return hist
    from .ctm import _fit_ctm
    return _fit_ctm(model, maxiter, tol, verbose, **kw)


fit_bang = fit   # `fit!`
