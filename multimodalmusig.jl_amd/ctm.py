"""Host-side mirror of `MMCTM` (src/MMCTM.jl) and `IMMCTM` (src/IMMCTM.jl) over the HIP backend.

Field names follow the reference structs (MMCTM.jl:1-27, IMMCTM.jl:1-27).  State lives in HBM inside the C-ABI handle;
nested fields (`model.θ[d][m]`, `model.γ[m][k]`, ...) are write-through views.  The per-document functions of the
reference (`update_ζ!(model, d)`, ...) change document d only (0-based here; mmm_ctm_update_doc); without `d` they process
every document in one launch, which is what `fitdoc!` over all documents does.  The reference's free functions (λ_objective,
ν_objective, α_objective, calculate_modality_loglikelihood: common.jl:11-46, MMCTM.jl:384-418, IMMCTM.jl:362-407) are
here as functions over plain arrays.  There is no CPU implementation in this package.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SolverOpts, check, lib
from .utils import pack_mm

_FID = {"mu": 0, "Sigma": 1, "invSigma": 2, "gamma": 3, "Elnphi": 4, "phi": 5, "lambda": 6, "nu": 7, "zeta": 8,
        "props": 9, "theta": 10, "alpha": 11}


class _View:
    """Nested list-like view over a flat device field; leaves are numpy copies, item assignment writes through."""

    def __init__(self, model, name, lens, prefix=()):
        self._m, self._name, self._lens, self._prefix = model, name, lens, prefix

    def _len(self):
        ln = self._lens[len(self._prefix)]
        return ln(self._prefix) if callable(ln) else ln

    def __len__(self):
        return self._len()

    def __getitem__(self, i):
        if i < 0:
            i += self._len()
        path = self._prefix + (i,)
        if len(path) == len(self._lens):
            return self._m._leaf_get(self._name, path)
        return _View(self._m, self._name, self._lens, path)

    def __setitem__(self, i, value):
        if i < 0:
            i += self._len()
        path = self._prefix + (i,)
        if len(path) == len(self._lens):
            self._m._leaf_set(self._name, path, value)
        else:
            sub = _View(self._m, self._name, self._lens, path)
            for j in range(len(sub)):
                sub[j] = value[j]

    def __iter__(self):
        return (self[i] for i in range(self._len()))

    def tolist(self):
        return [x.tolist() if isinstance(x, _View) else x for x in self]


class _CTM:
    def _init_common(self, k, X, ctx, xtol_rule, max_eval):
        self.K = [int(x) for x in k]
        self.M = len(self.K)
        self.MK = int(sum(self.K))
        self.X = X
        self.D = len(X)
        self._doc_ptr, self._term, self._count = pack_mm(X, self.M)
        D = self.D
        self.N = [[int(self._count[self._doc_ptr[m * (D + 1) + d]:self._doc_ptr[m * (D + 1) + d + 1]].sum()) for m in range(self.M)]
                  for d in range(D)]
        self._estart = [int(self._doc_ptr[m * (D + 1)]) for m in range(self.M)]
        self._nnz = [int(self._doc_ptr[m * (D + 1) + D]) - self._estart[m] for m in range(self.M)]
        self._toff = np.concatenate([[0], np.cumsum([self._nnz[m] * self.K[m] for m in range(self.M)])]).astype(np.int64)
        self._koff = np.concatenate([[0], np.cumsum(self.K)]).astype(np.int64)
        self.ctx = ctx or _lib.default_context()
        self._opts = SolverOpts()
        lib().mmm_solver_opts_default(C.byref(self._opts))
        self._opts.xtol_rule = int(xtol_rule)
        if max_eval:
            self._opts.max_eval = int(max_eval)
        self.converged = False
        self.elbo = float("nan")
        self.ll = None

    def _create(self, V, alpha_flat, gamma0, nfeat=None, J=None, features=None):
        """gamma0: flat init of one model, or R of them stacked ([R, GM]) for a restart batch."""
        self._h = C.c_void_p()
        gamma0 = np.ascontiguousarray(gamma0, dtype=np.float64)
        self.R = 1 if gamma0.ndim == 1 else int(gamma0.shape[0])
        if gamma0.size != self.R * self._GM:
            raise ValueError("γ0 has %d values, expected %d per model" % (gamma0.size, self._GM))
        self._sel = 0
        Kc = np.ascontiguousarray(self.K, dtype=np.int32); Vc = np.ascontiguousarray(V, dtype=np.int32)
        tp = self._term.ctypes.data if self._term.size else None
        cp = self._count.ctypes.data if self._count.size else None
        keep = [np.ascontiguousarray(x, dtype=np.int32) if x is not None else None for x in (nfeat, J, features)]
        ptr = [x.ctypes.data if x is not None else None for x in keep]
        check(lib().mmm_ctm_create_batch(self.ctx.h, self.R, self.D, self.M, Kc, Vc, np.ascontiguousarray(alpha_flat, dtype=np.float64), self._doc_ptr,
                                         tp, cp, ptr[0], ptr[1], ptr[2], gamma0.ravel(), C.byref(self._opts), C.byref(self._h)),
              self.ctx.h, "mmm_ctm_create_batch")
        _lib.track(self)
        self._alpha_dev = np.ascontiguousarray(alpha_flat, dtype=np.float64).ravel().copy()
        self.restart_ll = None; self.restart_elbo = None; self.restart_converged = None; self.restart_iters = None

    # ---- restart batch (scripts/run_mmctm.jl:77-134) -------------------------------------------------------------------
    def select(self, r):
        """Make restart `r` the model the fields and the per-model functions act on."""
        check(lib().mmm_ctm_select(self._h, int(r)), self.ctx.h, "select")
        self._sel = int(r)
        self._refresh_alpha()
        return self

    def _push_alpha(self):
        """`model.α` is a plain mutable field upstream (MMCTM.jl:12, IMMCTM.jl:11): what the caller has assigned since the last refresh goes
        to the device before any function that reads it (fit!, update_γ!, update_α!, the ELBO) -- of the selected restart in a batch."""
        flat = np.ascontiguousarray(np.concatenate([np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in self.α]) if self._immctm
                                    else np.asarray(self.α, dtype=np.float64)).ravel()
        if flat.size != self._nalpha:
            raise ValueError("model.α holds %d values, expected %d" % (flat.size, self._nalpha))
        if not np.array_equal(flat, getattr(self, "_alpha_dev", None)):
            self._set("alpha", flat)
            self._alpha_dev = flat.copy()

    def _refresh_alpha(self):
        """model.α mirrors the device value of the selected restart (it changes under update_α! / autoα)."""
        a = self._get("alpha")
        self._alpha_dev = a.copy()
        if self._immctm:
            off = np.concatenate([[0], np.cumsum(self.I)])
            self.α = [a[off[m]:off[m + 1]].copy() for m in range(self.M)]
        else:
            self.α = a.copy()

    @property
    def selected(self):
        return self._sel

    # ---- flat field transfer --------------------------------------------------------------------------------------
    def _fsize(self, name):
        D, MK, M = self.D, self.MK, self.M
        return {"mu": MK, "Sigma": MK * MK, "invSigma": MK * MK, "gamma": self._GM, "Elnphi": self._GM, "phi": self._GT if self._immctm else self._GM,
                "lambda": D * MK, "nu": D * MK, "zeta": D * M, "props": D * MK, "theta": int(self._toff[-1]), "alpha": self._nalpha}[name]

    def _get(self, name):
        out = np.empty(self._fsize(name), dtype=np.float64)
        check(lib().mmm_ctm_get(self._h, _FID[name], out, out.size), self.ctx.h, "mmm_ctm_get(%s)" % name)
        return out

    def _set(self, name, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64)
        check(lib().mmm_ctm_set(self._h, _FID[name], flat, flat.size), self.ctx.h, "mmm_ctm_set(%s)" % name)

    # ---- leaves ---------------------------------------------------------------------------------------------------
    def _span(self, name, path):
        """(start, stop, shape-for-reshape, transpose?) of a leaf inside the flat field."""
        D, MK, M = self.D, self.MK, self.M
        if name in ("lambda", "nu"):
            d, = path; return d * MK, (d + 1) * MK, None, False
        if name == "zeta":
            d, = path; return d * M, (d + 1) * M, None, False
        if name == "props":
            d, m = path; a = d * MK + int(self._koff[m]); return a, a + self.K[m], None, False
        if name == "theta":
            d, m = path
            e0 = int(self._doc_ptr[m * (D + 1) + d]) - self._estart[m]; e1 = int(self._doc_ptr[m * (D + 1) + d + 1]) - self._estart[m]
            Km = self.K[m]
            return int(self._toff[m]) + e0 * Km, int(self._toff[m]) + e1 * Km, (e1 - e0, Km), True
        if name in ("gamma", "Elnphi", "phi"):
            return self._topic_span(name, path)
        raise KeyError(name)

    def _leaf_get(self, name, path):
        a, b, shape, tr = self._span(name, path)
        v = self._get(name)[a:b].copy()
        if shape is not None:
            v = v.reshape(shape)
            if tr:
                v = v.T.copy()
        return v

    def _leaf_set(self, name, path, value):
        a, b, shape, tr = self._span(name, path)
        value = np.asarray(value, dtype=np.float64)
        if tr:
            value = value.T
        flat = self._get(name)
        if value.size != b - a:
            raise ValueError("%s%s expects %d values, got %d" % (name, list(path), b - a, value.size))
        flat[a:b] = value.ravel()
        self._set(name, flat)

    # ---- reference fields ---------------------------------------------------------------------------------------------
    def _vec_prop(name):
        return property(lambda s: s._get(name), lambda s, v: s._set(name, np.asarray(v, float).ravel()))

    def _mat_prop(name):
        return property(lambda s: s._get(name).reshape(s.MK, s.MK, order="F"), lambda s, v: s._set(name, np.asarray(v, float).ravel(order="F")))

    μ = _vec_prop("mu")
    Σ = _mat_prop("Sigma")
    invΣ = _mat_prop("invSigma")

    def _nested_prop(name, lens_fn):
        def getter(s):
            return _View(s, name, lens_fn(s))

        def setter(s, value):
            v = _View(s, name, lens_fn(s))
            for i in range(len(v)):
                v[i] = value[i]
        return property(getter, setter)

    λ = _nested_prop("lambda", lambda s: (s.D,))
    ν = _nested_prop("nu", lambda s: (s.D,))
    ζ = _nested_prop("zeta", lambda s: (s.D,))
    θ = _nested_prop("theta", lambda s: (s.D, s.M))

    def lam_matrix(self):
        return self._get("lambda").reshape(self.D, self.MK)

    def nu_matrix(self):
        return self._get("nu").reshape(self.D, self.MK)

    def close(self):
        if getattr(self, "_h", None):
            lib().mmm_ctm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solver_stats(self, per_doc=False):
        a = C.c_int64(); b = C.c_int64(); c = C.c_int64()
        pn = np.zeros(self.D, dtype=np.int32); pl = np.zeros(self.D, dtype=np.int32)
        check(lib().mmm_ctm_solver_stats(self._h, C.byref(a), C.byref(b), C.byref(c), pn.ctypes.data, pl.ctypes.data), self.ctx.h, "solver_stats")
        r = {"n_eval_nu": a.value, "n_eval_lambda": b.value, "n_capped": c.value}
        if per_doc:
            r["per_doc_nu"], r["per_doc_lambda"] = pn, pl
        return r

    def events(self):
        """Non-fatal events, counted as the reference would carry on (mmm_ctm_events): LD_MMA solves of the last E-step that hit the evaluation
        cap / met a non-finite objective value, and non-finite values in the log-likelihood history."""
        out = (C.c_int64 * 4)()
        check(lib().mmm_ctm_events(self._h, out), self.ctx.h, "events")
        return {"n_capped": int(out[0]), "n_nonfinite": int(out[1]), "n_nonfinite_ll": int(out[2])}

    def geometry(self):
        """Launch geometry that fixes the order of the sums across documents (mmm_ctm_geometry)."""
        g = (C.c_int * 8)()
        check(lib().mmm_ctm_geometry(self._h, g), self.ctx.h, "geometry")
        return {"L": g[0], "grid_e": g[1], "waves_e": g[2], "grid_m": g[3], "wide": int(g[4] == 1), "tdense": int(g[4] == 2), "Ls": g[5], "cpl": g[6], "solve_waves": g[7]}

    def objectives(self, d):
        """(λ_objective value, ∇λ, ν_objective value, ∇ν) of document d at its stored λ, ν, ζ, θ (common.jl:11-36)."""
        lv = C.c_double(); nv = C.c_double(); lg = np.zeros(self.MK); ng = np.zeros(self.MK)
        check(lib().mmm_ctm_objectives(self._h, int(d), C.byref(lv), lg, C.byref(nv), ng), self.ctx.h, "objectives")
        return lv.value, lg, nv.value, ng


class MMCTM(_CTM):
    """`MMCTM(k, α, X)` / `MMCTM(k, α, V, X)` -- MMCTM.jl:29-108 (init = :random only; `:document` is unusable in the
    reference, MMCTM.jl:64-74).  The random init γ[m][k] = rand(1:100, V[m]) (MMCTM.jl:60-63) is drawn with numpy (`seed`)
    unless `γ0` (list over m of K_m x V_m arrays) is given."""
    _immctm = False

    def __init__(self, k, α, *args, γ0=None, seed=None, init="random", ctx=None, xtol_rule=0, max_eval=0, restarts=None):
        if init != "random":
            raise ValueError("init must be either :random or :document")       # MMCTM.jl:76 (only :random works upstream)
        if len(args) == 1:
            V, X = None, args[0]
        elif len(args) == 2:
            V, X = args
        else:
            raise TypeError("MMCTM(k, α, [V,] X)")
        self._init_common(k, X, ctx, xtol_rule, max_eval)
        D = self.D
        if V is None:                                        # MMCTM.jl:94-108
            V = []
            for m in range(self.M):
                t = self._term[self._estart[m]:self._estart[m] + self._nnz[m]]
                V.append(int(t.max()) + 1 if t.size else 0)
        self.V = [int(v) for v in V]
        self.α = np.asarray(α, dtype=np.float64).copy()
        self._goff = np.concatenate([[0], np.cumsum([self.K[m] * self.V[m] for m in range(self.M)])]).astype(np.int64)
        self._GM = self._GT = int(self._goff[-1]); self._nalpha = self.M
        def draw(sd):
            rng = np.random.default_rng(sd)
            return [rng.integers(1, 101, size=(self.K[m], self.V[m])).astype(np.float64) for m in range(self.M)]

        def flat(g0):
            return np.concatenate([np.asarray(g0[m], dtype=np.float64).reshape(self.K[m], self.V[m]).ravel() for m in range(self.M)])
        if restarts is None:
            g = flat(draw(seed) if γ0 is None else γ0)
        else:
            # one model per restart: γ0 = list of R inits, or seeds seed, seed+1, ... (fit_restart, scripts/run_mmctm.jl:77-84)
            R = int(restarts)
            if γ0 is None:
                γ0 = [draw(None if seed is None else seed + r) for r in range(R)]
            if len(γ0) != R:
                raise ValueError("restarts=%d but %d initialisations given" % (R, len(γ0)))
            g = np.stack([flat(x) for x in γ0])
        self._create(self.V, self.α, g)

    def _topic_span(self, name, path):
        m, k = path
        a = int(self._goff[m]) + k * self.V[m]
        return a, a + self.V[m], None, False

    γ = _CTM._nested_prop("gamma", lambda s: (s.M, lambda p: s.K[p[0]]))
    Elnϕ = _CTM._nested_prop("Elnphi", lambda s: (s.M, lambda p: s.K[p[0]]))
    ϕ = _CTM._nested_prop("phi", lambda s: (s.M, lambda p: s.K[p[0]]))
    props = _CTM._nested_prop("props", lambda s: (s.D, s.M))


class IMMCTM(_CTM):
    """`IMMCTM(k, α, features, X)` -- IMMCTM.jl:29-88.  features[m]: V_m x I_m matrix of 1-based feature values;
    α: per modality scalar (IMMCTM.jl:81-88) or per-feature vector.  γ0: flat init in the layout [m][k][i][j]."""
    _immctm = True

    def __init__(self, k, α, features, X, γ0=None, seed=None, ctx=None, xtol_rule=0, max_eval=0, restarts=None):
        self._init_common(k, X, ctx, xtol_rule, max_eval)
        feats = [np.asarray(f, dtype=np.int64) for f in features]
        self.features = feats
        self.I = [int(f.shape[1]) for f in feats]                         # IMMCTM.jl:41
        self.J = [[int(x) for x in f.max(axis=0)] for f in feats]         # IMMCTM.jl:42
        self.V = [int(f.shape[0]) for f in feats]                         # IMMCTM.jl:43
        if np.ndim(α[0]) == 0:
            self.α = [np.full(self.I[m], float(α[m])) for m in range(self.M)]
        else:
            self.α = [np.asarray(α[m], dtype=np.float64).copy() for m in range(self.M)]
        self._SJ = [int(sum(self.J[m])) for m in range(self.M)]
        self._mgoff = np.concatenate([[0], np.cumsum([self.K[m] * self._SJ[m] for m in range(self.M)])]).astype(np.int64)
        self._GM = int(self._mgoff[-1]); self._GT = int(sum(self.K[m] * self.V[m] for m in range(self.M)))
        self._nalpha = int(sum(self.I))
        if restarts is None:
            if γ0 is None:
                γ0 = np.random.default_rng(seed).integers(1, 101, size=self._GM).astype(np.float64)
            γ0 = np.asarray(γ0, dtype=np.float64).ravel()
        else:
            R = int(restarts)
            if γ0 is None:
                γ0 = [np.random.default_rng(None if seed is None else seed + r).integers(1, 101, size=self._GM) for r in range(R)]
            γ0 = np.stack([np.asarray(x, dtype=np.float64).ravel() for x in γ0])
            if γ0.shape[0] != R:
                raise ValueError("restarts=%d but %d initialisations given" % (R, γ0.shape[0]))
        featflat = np.concatenate([(f - 1).T.ravel() for f in feats]).astype(np.int32)
        self._create(self.V, np.concatenate(self.α), γ0,
                     nfeat=np.asarray(self.I), J=np.concatenate([np.asarray(j) for j in self.J]), features=featflat)

    def _topic_span(self, name, path):
        if name == "phi":
            raise AttributeError("IMMCTM has no ϕ field (IMMCTM.jl:1-27)")
        m, k, i = path
        a = int(self._mgoff[m]) + k * self._SJ[m] + int(sum(self.J[m][:i]))
        return a, a + self.J[m][i], None, False

    γ = _CTM._nested_prop("gamma", lambda s: (s.M, lambda p: s.K[p[0]], lambda p: s.I[p[0]]))
    Elnϕ = _CTM._nested_prop("Elnphi", lambda s: (s.M, lambda p: s.K[p[0]], lambda p: s.I[p[0]]))


# ---- function API --------------------------------------------------------------------------------------------------------
def _call(model, fn, what):
    check(getattr(lib(), fn)(model._h), model.ctx.h, what)


def _doc_stage(model, fn, stage, what, d):
    if d is None:
        _call(model, fn, what)
    else:
        check(lib().mmm_ctm_update_doc(model._h, stage, int(d)), model.ctx.h, "%s(model, %d)" % (what, d))


def update_ζ(model, d=None):      # MMCTM.jl:172-181
    _doc_stage(model, "mmm_ctm_update_zeta", 0, "update_ζ!", d)


def update_θ_ctm(model, d=None):  # MMCTM.jl:183-198 / IMMCTM.jl:152-172
    _doc_stage(model, "mmm_ctm_update_theta", 1, "update_θ!", d)


def update_ν(model, d=None):      # MMCTM.jl:156-170
    _doc_stage(model, "mmm_ctm_update_nu", 2, "update_ν!", d)


def update_λ_ctm(model, d=None):  # MMCTM.jl:127-143
    _doc_stage(model, "mmm_ctm_update_lambda", 3, "update_λ!", d)


def calculate_sumθ(model, d):     # MMCTM.jl:110-117
    out = np.zeros(model.MK)
    check(lib().mmm_ctm_doc_sums(model._h, int(d), out.ctypes.data, None), model.ctx.h, "calculate_sumθ")
    return out


def calculate_Ndivζ(model, d):    # MMCTM.jl:119-125
    out = np.zeros(model.MK)
    check(lib().mmm_ctm_doc_sums(model._h, int(d), None, out.ctypes.data), model.ctx.h, "calculate_Ndivζ")
    return out


# ---- free functions (src/common.jl; the log-likelihood helpers) --------------------------------------------------------------
def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def λ_objective(λ, grad, ν, Ndivζ, sumθ, μ, invΣ, ctx=None):      # common.jl:11-23; `grad` (array or None) is filled in place like ∇λ
    ctx = ctx or _lib.default_context()
    n = len(λ); v = C.c_double()
    g = np.zeros(n) if grad is not None else None
    check(lib().mmm_lambda_objective(ctx.h, n, _f64(λ), _f64(ν), _f64(Ndivζ), _f64(sumθ), _f64(μ), _f64(np.asarray(invΣ, dtype=np.float64).T.ravel()),
                                     C.byref(v), g.ctypes.data if g is not None else None), ctx.h, "λ_objective")
    if grad is not None:
        grad[:] = g
    return v.value


def ν_objective(ν, grad, λ, Ndivζ, μ, invΣ, ctx=None):            # common.jl:25-36
    ctx = ctx or _lib.default_context()
    n = len(ν); v = C.c_double()
    g = np.zeros(n) if grad is not None else None
    mu = _f64(μ) if μ is not None else None
    check(lib().mmm_nu_objective(ctx.h, n, _f64(ν), _f64(λ), _f64(Ndivζ), mu.ctypes.data if mu is not None else None,
                                 _f64(np.asarray(invΣ, dtype=np.float64).T.ravel()), C.byref(v), g.ctypes.data if g is not None else None), ctx.h, "ν_objective")
    if grad is not None:
        grad[:] = g
    return v.value


def α_objective(α, grad, sum_Elnϕ, K, V, ctx=None):               # common.jl:38-46 (α a 1-vector, as NLopt hands it over)
    ctx = ctx or _lib.default_context()
    v = C.c_double(); g = C.c_double()
    check(lib().mmm_alpha_objective(ctx.h, float(α[0]), float(sum_Elnϕ), int(K), int(V), C.byref(v), C.byref(g)), ctx.h, "α_objective")
    if grad is not None and len(grad) > 0:
        grad[0] = g.value
    return v.value


def calculate_modality_loglikelihood(X, props, ϕ, features=None, ctx=None, softmax=True):
    """MMCTM.jl:402-418 (X: documents of ONE modality, props[d]: K proportions, ϕ[k]: V term probabilities) or, with `features` (V x I,
    1-based values), IMMCTM.jl:387-407 (props is then η[d], ϕ[k][i]: J_i feature-value probabilities)."""
    from .utils import pack_lda
    ctx = ctx or _lib.default_context()
    doc_ptr, term, count = pack_lda(X)
    D, K = len(X), len(ϕ)
    P = _f64(np.asarray([np.asarray(p, dtype=np.float64) for p in props]).reshape(D, K))
    tp = term.ctypes.data if term.size else None; cp = count.ctypes.data if count.size else None
    v = C.c_double()
    if features is None:
        F = _f64(np.asarray([np.asarray(x, dtype=np.float64) for x in ϕ]))
        check(lib().mmm_mixture_loglik(ctx.h, D, K, F.shape[1], doc_ptr, tp, cp, P.ravel(), F.ravel(), C.byref(v)), ctx.h, "calculate_modality_loglikelihood")
    else:
        f = np.asarray(features, dtype=np.int64)
        # J from the lengths of ϕ[k][i] (what the packed layout below follows -- IMMCTM.jl:42 takes it from the feature table, and a top value
        # that no term uses would make the two disagree): every topic the same lengths, every feature value inside them
        I = f.shape[1]
        J = np.ascontiguousarray([len(ϕ[0][i]) for i in range(I)], dtype=np.int32)
        for k in range(K):
            if len(ϕ[k]) != I or any(len(ϕ[k][i]) != J[i] for i in range(I)):
                raise ValueError("ϕ[%d] does not hold %d feature distributions of lengths %s" % (k, I, J.tolist()))
        if f.size and (f.min() < 1 or (f.max(axis=0) > J).any()):
            raise ValueError("feature values outside 1..J = %s" % J.tolist())
        F = _f64(np.concatenate([np.concatenate([np.asarray(ϕ[k][i], dtype=np.float64) for i in range(f.shape[1])]) for k in range(K)]))
        check(lib().mmm_mixture_loglik_features(ctx.h, D, K, f.shape[0], f.shape[1], J, np.ascontiguousarray((f - 1).T.ravel(), dtype=np.int32), doc_ptr, tp, cp,
                                                P.ravel(), 1 if softmax else 0, F, C.byref(v)), ctx.h, "calculate_modality_loglikelihood")
    return v.value


def calculate_docmodality_loglikelihood(Xd, props, ϕ, features=None, ctx=None):      # MMCTM.jl:384-400 / IMMCTM.jl:362-385
    return calculate_modality_loglikelihood([Xd], [props], ϕ, features=features, ctx=ctx)


def update_μ(model):              # MMCTM.jl:200-202
    _call(model, "mmm_ctm_update_mu", "update_μ!")


def update_Σ(model):              # MMCTM.jl:204-212
    _call(model, "mmm_ctm_update_Sigma", "update_Σ!")


def update_γ_ctm(model):          # MMCTM.jl:224-242 / IMMCTM.jl:199-223
    model._push_alpha()
    _call(model, "mmm_ctm_update_gamma", "update_γ!")


def update_Elnϕ(model):           # MMCTM.jl:214-222 / IMMCTM.jl:188-197
    _call(model, "mmm_ctm_update_Elnphi", "update_Elnϕ!")


def update_α(model):              # MMCTM.jl:252-269 / IMMCTM.jl:225-244
    model._push_alpha()
    _call(model, "mmm_ctm_update_alpha", "update_α!")
    model._refresh_alpha()


def update_props(model):          # MMCTM.jl:145-154
    _call(model, "mmm_ctm_update_props", "update_props!")


def update_ϕ_ctm(model):          # MMCTM.jl:244-250
    _call(model, "mmm_ctm_update_phi", "update_ϕ!")


def fitdoc(model, d=None):        # MMCTM.jl:450-455
    update_ζ(model); update_θ_ctm(model); update_ν(model); update_λ_ctm(model)


def calculate_loglikelihoods(model):   # MMCTM.jl:446-448 / IMMCTM.jl:408-428
    out = np.zeros(model.M)
    check(lib().mmm_ctm_loglik(model._h, out), model.ctx.h, "calculate_loglikelihoods")
    return out


def _fit_flags(autoα, updateΣ):
    return (1 if updateΣ else 0) | (2 if autoα else 0)      # MMM_FIT_UPDATE_SIGMA | MMM_FIT_AUTO_ALPHA


def _fit_ctm(model, maxiter, tol, verbose, autoα=False, updateΣ=True):
    maxiter = 100 if maxiter is None else int(maxiter)
    model._push_alpha()
    ll = np.zeros(maxiter * model.M); ni = C.c_int(); cv = C.c_int(); el = C.c_double()
    check(lib().mmm_ctm_fit(model._h, maxiter, float(tol), _fit_flags(autoα, updateΣ), ll.ctypes.data, C.byref(ni), C.byref(cv), C.byref(el)),
          model.ctx.h, "fit!(::%s)" % type(model).__name__)
    hist = ll[:ni.value * model.M].reshape(ni.value, model.M).copy()
    if verbose:
        for i, v in enumerate(hist):
            print("%d\tLog-likelihoods: %s" % (i + 1, ", ".join(repr(float(x)) for x in v)))
    model.converged = bool(cv.value); model.elbo = el.value; model.ll = hist[-1].copy()
    if autoα:
        model._refresh_alpha()
    return hist


def fit_restarts(model, maxiter=100, tol=1e-4, verbose=False, updateΣ=True, autoα=False):
    """`fit!` of every restart of a batch model (constructed with `restarts=R` or R stacked γ0), all restarts advancing
    together on the GPU -- what `fit_seed_models` (scripts/run_mmctm.jl:97-109) gets from `pmap(fit_restart, seeds)`.
    Returns the list of per-restart ll histories ([n_iter_r, M] each); per-restart results are kept on the model as
    `restart_ll` ([R, M] final ll), `restart_elbo`, `restart_converged`, `restart_iters`."""
    R, M = model.R, model.M
    maxiter = int(maxiter)
    ll = np.zeros(R * maxiter * M); ni = np.zeros(R, dtype=np.int32); cv = np.zeros(R, dtype=np.int32); el = np.zeros(R)
    check(lib().mmm_ctm_fit_batch(model._h, maxiter, float(tol), _fit_flags(autoα, updateΣ), ll.ctypes.data, ni.ctypes.data, cv.ctypes.data, el.ctypes.data),
          model.ctx.h, "fit_restarts(::%s)" % type(model).__name__)
    ll = ll.reshape(R, maxiter, M)
    hists = [ll[r, :ni[r]].copy() for r in range(R)]
    model.restart_ll = np.stack([h[-1] for h in hists])
    model.restart_elbo = el.copy(); model.restart_converged = cv.astype(bool); model.restart_iters = ni.copy()
    if verbose:
        for r in range(R):
            print("restart %d	%d iterations	Log-likelihoods: %s" % (r, ni[r], ", ".join(repr(float(x)) for x in hists[r][-1])))
    if autoα:
        model._refresh_alpha()
    s = model.selected
    model.converged = bool(cv[s]); model.elbo = float(el[s]); model.ll = hists[s][-1].copy()
    return hists


def pick_optimal_modality_models(model):
    """Index of the best restart per modality by final log-likelihood -- scripts/run_mmctm.jl:86-95."""
    return [int(i) for i in np.argmax(model.restart_ll, axis=0)]
