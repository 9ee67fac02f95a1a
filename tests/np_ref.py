"""Independent numpy/scipy restatement of the closed-form parts of the hot path (and of the ELBO terms),
written directly from SURVEY.md Appendix A / the Julia source in nested reference layout (X[d][m], theta[d][m]
K x W ...).  Used ONLY to cross-validate the C oracle (tests/test_oracle_crosscheck.py): two independent
restatements agreeing to ~1e-12 is what stands in for the reference run we cannot make here."""
import numpy as np
from scipy.special import digamma as psi, gammaln


# ---------------------------------------------------------------------------------------------- LDA
def lda_iteration(X, K, alpha, eta, lam, phi):
    """One pass of LDA.jl:202-209. X[d]: (W,2) 1-based. lam: V x K. phi[d]: K x W. Returns dict."""
    D = len(X); V = lam.shape[0]
    Elnbeta = psi(lam) - psi(lam.sum(axis=0, keepdims=True))
    gamma = np.full((K, D), alpha)
    for d in range(D):
        gamma[:, d] += phi[d] @ X[d][:, 1]
    Elntheta = psi(gamma) - psi(gamma.sum(axis=0, keepdims=True))
    newphi = []
    for d in range(D):
        p = np.exp(Elntheta[:, d][:, None] + Elnbeta[X[d][:, 0] - 1, :].T)
        newphi.append(p / p.sum(axis=0, keepdims=True))
    lam2 = np.full((V, K), eta)
    for d in range(D):
        np.add.at(lam2, (X[d][:, 0] - 1), newphi[d].T * X[d][:, 1][:, None])
    Elnbeta2 = psi(lam2) - psi(lam2.sum(axis=0, keepdims=True))
    beta = lam2 / lam2.sum(axis=0, keepdims=True)
    theta = gamma / gamma.sum(axis=0, keepdims=True)
    ll = 0.0; N = 0
    for d in range(D):
        N += X[d][:, 1].sum()
        ll += (X[d][:, 1] * np.log(beta[X[d][:, 0] - 1, :] @ theta[:, d])).sum()
    return dict(gamma=gamma, Elntheta=Elntheta, phi=newphi, lam=lam2, Elnbeta=Elnbeta2, beta=beta, theta=theta, ll=ll / N)


def lda_elbo(X, K, alpha, eta, lam, Elnbeta, gamma, Elntheta, phi):
    D = len(X); V = lam.shape[0]
    t = np.zeros(7)
    t[0] = K * (gammaln(V * eta) - V * gammaln(eta)) + (eta - 1) * Elnbeta.sum()
    t[1] = D * (gammaln(K * alpha) - K * gammaln(alpha)) + (alpha - 1) * Elntheta.sum()
    for d in range(D):
        n = X[d][:, 1]
        t[2] += (phi[d] * Elntheta[:, d][:, None] * n[None, :]).sum()
        t[3] += (phi[d].T * Elnbeta[X[d][:, 0] - 1, :] * n[:, None]).sum()
        p = phi[d]
        t[6] += np.where(p > 0, p * np.log(np.where(p > 0, p, 1.0)), 0.0).sum()
    t[4] = gammaln(lam).sum() - gammaln(lam.sum(axis=0)).sum() - ((lam - 1) * Elnbeta).sum()
    t[5] = gammaln(gamma).sum() - gammaln(gamma.sum(axis=0)).sum() - ((gamma - 1) * Elntheta).sum()
    return t[0] + t[1] + t[2] + t[3] - t[4] - t[5] - t[6], t


# -------------------------------------------------------------------------------------------- MMCTM
def softmax_cols(a):
    e = np.exp(a)
    return e / e.sum(axis=0, keepdims=True)


def mmctm_zeta_theta(X, K, lam_d, nu_d, Elnphi, d):
    """zeta and theta of one doc (MMCTM.jl:172-198). Elnphi[m]: K_m x V_m."""
    M = len(K); off = 0; zeta = np.zeros(M); theta = []
    for m in range(M):
        sl = slice(off, off + K[m])
        zeta[m] = np.exp(lam_d[sl] + 0.5 * nu_d[sl]).sum()
        v = X[d][m][:, 0] - 1
        theta.append(softmax_cols(lam_d[sl][:, None] + Elnphi[m][:, v]))
        off += K[m]
    return zeta, theta


def mmctm_objs(X, K, d, zeta, theta, mu, invS):
    M = len(K)
    sumth = np.concatenate([theta[m] @ X[d][m][:, 1] for m in range(M)])
    Ndz = np.concatenate([np.full(K[m], X[d][m][:, 1].sum() / zeta[m]) for m in range(M)])

    def f_lam(lam, nu):
        diff = lam - mu; Ee = np.exp(lam + 0.5 * nu)
        val = -0.5 * diff @ invS @ diff + lam @ sumth - Ndz @ Ee
        return val, -invS @ diff + sumth - Ndz * Ee

    def f_nu(nu, lam):
        Ee = np.exp(lam + 0.5 * nu)
        val = -0.5 * (nu * np.diag(invS)).sum() - Ndz @ Ee + 0.5 * np.log(nu).sum()
        return val, -0.5 * np.diag(invS) - 0.5 * Ndz * Ee + 0.5 / nu
    return sumth, Ndz, f_lam, f_nu


def mmctm_mstep(X, K, V, alpha, lam, nu, theta):
    """lam, nu: D x MK; theta[d][m]. Returns mu, Sigma, invSigma, gamma[m] (K_m x V_m), Elnphi, phi, props, ll."""
    D = len(X); M = len(K)
    mu = lam.mean(axis=0)
    S = np.diag(nu.sum(axis=0))
    for d in range(D):
        df = lam[d] - mu; S = S + np.outer(df, df)
    S = S / D
    invS = np.linalg.inv(S)
    gamma = [np.full((K[m], V[m]), alpha[m]) for m in range(M)]
    for d in range(D):
        for m in range(M):
            np.add.at(gamma[m], (slice(None), X[d][m][:, 0] - 1), theta[d][m] * X[d][m][:, 1][None, :])
    Elnphi = [psi(g) - psi(g.sum(axis=1, keepdims=True)) for g in gamma]
    phi = [g / g.sum(axis=1, keepdims=True) for g in gamma]
    props = np.zeros_like(lam); off = 0
    for m in range(M):
        e = np.exp(lam[:, off:off + K[m]]); props[:, off:off + K[m]] = e / e.sum(axis=1, keepdims=True); off += K[m]
    ll = np.zeros(M); off = 0
    for m in range(M):
        tot = 0.0; N = 0
        for d in range(D):
            n = X[d][m][:, 1]
            if n.sum() > 0:
                pw = props[d, off:off + K[m]] @ phi[m][:, X[d][m][:, 0] - 1]
                tot += (n * np.log(pw)).sum(); N += n.sum()
        ll[m] = tot / N; off += K[m]
    return mu, S, invS, gamma, Elnphi, phi, props, ll


def mmctm_elbo(X, K, V, alpha, mu, invS, gamma, Elnphi, lam, nu, zeta, theta):
    D = len(X); M = len(K); MK = sum(K)
    t = np.zeros(7)
    for m in range(M):
        for k in range(K[m]):
            t[0] += -(V[m] * gammaln(alpha[m]) - gammaln(V[m] * alpha[m])) + (alpha[m] - 1) * Elnphi[m][k].sum()
            t[4] += -(gammaln(gamma[m][k]).sum() - gammaln(gamma[m][k].sum())) + ((gamma[m][k] - 1) * Elnphi[m][k]).sum()
    sign, logdet = np.linalg.slogdet(invS)
    for d in range(D):
        df = lam[d] - mu
        t[1] += 0.5 * (logdet - MK * np.log(2 * np.pi) - (nu[d] * np.diag(invS)).sum() - df @ invS @ df)
        sumth = np.concatenate([theta[d][m] @ X[d][m][:, 1] for m in range(M)])
        Nd = np.array([X[d][m][:, 1].sum() for m in range(M)], dtype=float)
        Ndz = np.concatenate([np.full(K[m], Nd[m] / zeta[d][m]) for m in range(M)])
        t[2] += lam[d] @ sumth - (Ndz @ np.exp(lam[d] + 0.5 * nu[d]) - Nd.sum()) - (Nd * np.log(zeta[d])).sum()
        t[5] += -0.5 * (np.log(nu[d]).sum() + MK * (np.log(2 * np.pi) + 1))
        for m in range(M):
            n = X[d][m][:, 1]; th = theta[d][m]
            t[3] += (n[None, :] * th * Elnphi[m][:, X[d][m][:, 0] - 1]).sum()
            t[6] += (n[None, :] * np.where(th > 0, th * np.log(np.where(th > 0, th, 1.0)), 0.0)).sum()
    return t[0] + t[1] + t[2] + t[3] - t[4] - t[5] - t[6], t


# ----------------------------------------------------------------------------------- synthetic corpora
def synth_lda(D, V, K, seed, mean_n=3000, conc=0.1):
    """SURVEY.md §8(d) generator (LDA flavour). Returns X (list of (W,2) int64 arrays, 1-based) and lambda0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    beta = rng.dirichlet(np.full(V, conc), size=K)
    X = []
    for d in range(D):
        th = rng.dirichlet(np.full(K, 0.5))
        n = 200 + rng.poisson(mean_n)
        c = rng.multinomial(n, th @ beta)
        idx = np.nonzero(c)[0]
        X.append(np.stack([idx + 1, c[idx]], axis=1).astype(np.int64))
    lam0 = rng.integers(1, 101, size=(V, K)).astype(np.float64)
    return X, lam0


def synth_mm(D, V, K, seed, means=None, conc=0.1, empty_frac=0.0):
    """SURVEY.md §8(d) generator (CTM flavour). X[d][m]; gamma0[m] K_m x V_m."""
    rng = np.random.Generator(np.random.PCG64(seed))
    M = len(K)
    means = means or [3000 if V[m] >= 90 else (150 if V[m] >= 36 else 100) for m in range(M)]
    beta = [rng.dirichlet(np.full(V[m], conc), size=K[m]) for m in range(M)]
    X = []
    for d in range(D):
        eta = rng.standard_normal(sum(K)); off = 0; doc = []
        for m in range(M):
            e = np.exp(eta[off:off + K[m]]); th = e / e.sum(); off += K[m]
            n = 200 + rng.poisson(means[m])
            if empty_frac > 0 and m > 0 and rng.random() < empty_frac:
                n = 0
            c = rng.multinomial(n, th @ beta[m])
            idx = np.nonzero(c)[0]
            doc.append(np.stack([idx + 1, c[idx]], axis=1).astype(np.int64).reshape(-1, 2))
        X.append(doc)
    gamma0 = [rng.integers(1, 101, size=(K[m], V[m])).astype(np.float64) for m in range(M)]
    return X, gamma0


# ------------------------------------------------------------------------------------ LD_MMA, independent restatement
def ccsa_mma(fun, x0, lower=None, xtol_rel=1e-4, xtol_abs=1e-4, rule=0, max_eval=100000):
    """Conservative convex separable approximations with MMA-type approximating functions and NO nonlinear constraint, minimising
    fun(x) -> (value, gradient) over x >= lower.  Written from K. Svanberg, "A class of globally convergent optimization methods based on
    conservative convex separable approximations", SIAM J. Optim. 12 (2002) 555-573, in the form NLopt documents for LD_MMA
    (NLopt algorithms manual, "MMA (Method of Moving Asymptotes) and CCSA"; stopping rules: NLopt reference, xtol_rel / xtol_abs) --
    NOT from oracle/mmm_oracle.c; tests/test_mma_independent.py holds the two against each other step by step.

    Around the best point y found so far (value F, gradient g) the approximation is
        G(y + d) = F + sum_j [ g_j s_j^2 d_j + (|g_j| s_j + rho/2) d_j^2 ] / (s_j^2 - d_j^2),          |d_j| <= 0.9 s_j,
    separable and strictly convex in each d_j; s are the asymptote distances (1 when a bound is infinite), rho the conservativity
    parameter.  Inner iterations raise rho until G(candidate) >= f(candidate); outer iterations relax rho and move the asymptotes
    by the sign pattern of the last two steps.
    rule 0: NLopt >= 2.7 stop  (|x - x_old|_1 < xtol_rel |x|_1, or every |dx_j| < xtol_abs);  rule 1: NLopt <= 2.6 (per coordinate:
    |dx| < xtol_abs or |dx| < xtol_rel (|x| + |x_old|)/2 or, with xtol_rel > 0, dx == 0).
    Returns (best x, best value, number of evaluations, trace); trace = one dict per inner iteration."""
    y = np.array(x0, dtype=np.float64)                   # best point so far
    n = y.size
    lo = np.full(n, -np.inf) if lower is None else np.broadcast_to(np.asarray(lower, dtype=np.float64), (n,)).copy()
    s = np.ones(n)                                       # no finite upper bound anywhere on this path
    rho = 1.0
    F, g = fun(y.copy()); g = np.array(g, dtype=np.float64)
    evals = 1
    cand = y.copy()                                      # the latest candidate (the sequence whose steps drive s and the stop test)
    steps = []                                           # candidates at the end of the previous outer iterations
    trace = []
    while evals < max_eval:
        start = cand.copy()
        conservative = False
        while not conservative:
            # minimiser of every one-dimensional piece: root of  u d^2 + 2 v s^2 d + u s^2 = 0  inside the asymptotes
            u = g * (s * s)                                              # (the root is ill-conditioned next to a bound, |u| ~ v s: keep NLopt's operand order)
            v = np.abs(g) * s + 0.5 * rho
            with np.errstate(invalid="ignore", divide="ignore"):
                d = (u / v) / (-1.0 - np.sqrt(np.abs(1.0 - (u / (v * s)) ** 2)))
            c = y + d
            c = np.maximum(c, lo)                                        # box
            c = np.minimum(np.maximum(c, y - 0.9 * s), y + 0.9 * s)      # move limit
            d = c - y
            inv = 1.0 / (s * s - d * d)
            G = F
            W = 0.0
            for j in range(n):                                           # sums in coordinate order
                G += (g[j] * (s[j] * s[j] * d[j]) + (abs(g[j]) * s[j] + 0.5 * rho) * (d[j] * d[j])) * inv[j]
                W += 0.5 * (d[j] * d[j]) * inv[j]
            fc, gc = fun(c.copy()); evals += 1
            trace.append(dict(rho=rho, sigma=s.copy(), x=c.copy(), fcur=fc, gval=G, wval=W))
            conservative = G >= fc
            cand = c
            if fc < F:
                F, y, g = fc, c.copy(), np.array(gc, dtype=np.float64)
            if evals >= max_eval:
                return y, F, evals, trace
            if not conservative and fc > G:
                rho = min(10.0 * rho, 1.1 * (rho + (fc - G) / W))
        dxs = np.abs(cand - start)
        if rule == 0:
            done = dxs.sum() < xtol_rel * np.abs(cand).sum() or bool(np.all(dxs < xtol_abs))
        else:
            done = bool(np.all((dxs < xtol_abs) | (dxs < xtol_rel * 0.5 * (np.abs(cand) + np.abs(start))) | ((xtol_rel > 0) & (cand == start))))
        if done:
            break
        rho = max(0.1 * rho, 1e-5)
        if steps:                                                        # from the second outer iteration on
            osc = (cand - start) * (start - steps[-1])
            s = s * np.where(osc < 0, 0.7, np.where(osc > 0, 1.2, 1.0))
        steps.append(start)
    return y, F, evals, trace
