#!/usr/bin/env python3
"""Generate tests/golden/reference_kats.json.

Each entry is one known-answer test held by the reference's own test-suite for the hot path
(/root/reference/test/{lda,mmctm,immctm,common}.jl): the INPUTS are the literals of that test, the
EXPECTED values are that test's closed-form expectation evaluated here in 50-digit arithmetic with mpmath
(independent of the oracle and of the HIP kernels).  The file is data only; the reference cannot be
executed in the build container (no julia, no NLopt), so these formulas -- not a reference run -- are
what pins the oracle.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os

import mpmath as mp

mp.mp.dps = 50
psi = mp.digamma
E = mp.e


def f(x):
    return float(x)


def fl(v):
    return [f(x) for x in v]


out = {}

# toy corpora ------------------------------------------------------------------------------------------
X_LDA = [[[1, 5], [2, 8]], [[1, 2], [2, 5]]]                       # test/lda.jl:7-16
X_MM = [[[[1, 5], [2, 8]], [[1, 2], [2, 5]]],                      # test/mmctm.jl:6-33 (= immctm.jl:24-51)
        [[[3, 4], [4, 9]], [[3, 4], [4, 6]]]]
FEATURES = [[[1, 1], [1, 2], [2, 1], [2, 2]], [[1, 1], [1, 2], [2, 1], [2, 2]]]  # test/immctm.jl:8-23
out["corpora"] = {"X_lda": X_LDA, "K_lda": 2, "alpha_lda": 0.1, "eta_lda": 0.1,
                  "X_mm": X_MM, "K_mm": [2, 3], "alpha_mm": [0.1, 0.1], "features": FEATURES}

# ---- test/lda.jl:18-36 constructor ---------------------------------------------------------------------
out["lda_ctor"] = {"ref": "test/lda.jl:18-36", "N": [13, 7], "V": 2, "V_given": 3}

# ---- test/lda.jl:38-61 update_phi ------------------------------------------------------------------------
Elnth = [[mp.mpf("0.5"), mp.mpf("-1.1")], [mp.mpf("2.3"), mp.mpf("-0.7")]]    # [k][d]
Elnb = [[mp.mpf("-0.2"), mp.mpf("-0.9")], [mp.mpf("-1.1"), mp.mpf("0.3")]]    # [v][k]
phi = [[None, None], [None, None]]                                             # [k][w] of doc 1
for k in range(2):
    for w in range(2):
        phi[k][w] = mp.exp(Elnth[k][0] + Elnb[w][k])
for w in range(2):
    s = phi[0][w] + phi[1][w]
    phi[0][w] /= s; phi[1][w] /= s
out["lda_update_phi"] = {"ref": "test/lda.jl:38-61", "Elntheta": [[0.5, -1.1], [2.3, -0.7]],
                         "Elnbeta": [[-0.2, -0.9], [-1.1, 0.3]], "phi_doc1": [fl(r) for r in phi]}

# ---- test/lda.jl:63-80 update_gamma ----------------------------------------------------------------------
al = mp.mpf("0.1")
ph = [[mp.mpf("0.4"), mp.mpf("0.2")], [mp.mpf("0.6"), mp.mpf("0.8")]]
g = [al + ph[k][0] * 5 + ph[k][1] * 8 for k in range(2)]
out["lda_update_gamma"] = {"ref": "test/lda.jl:63-80", "phi_doc1": [[0.4, 0.2], [0.6, 0.8]],
                           "gamma_doc1": fl(g), "Elntheta_doc1": fl([psi(g[k]) - psi(g[0] + g[1]) for k in range(2)])}

# ---- test/lda.jl:82-103 update_lambda --------------------------------------------------------------------
ph2 = [[[mp.mpf("0.4"), mp.mpf("0.2")], [mp.mpf("0.6"), mp.mpf("0.8")]],
       [[mp.mpf("0.1"), mp.mpf("0.6")], [mp.mpf("0.9"), mp.mpf("0.4")]]]
eta = mp.mpf("0.1")
lam = [[None, None], [None, None]]   # [v][k]
for v in range(2):
    for k in range(2):
        lam[v][k] = eta + ph2[0][k][v] * X_LDA[0][v][1] + ph2[1][k][v] * X_LDA[1][v][1]
Eb = [[psi(lam[v][k]) - psi(lam[0][k] + lam[1][k]) for k in range(2)] for v in range(2)]
out["lda_update_lambda"] = {"ref": "test/lda.jl:82-103",
                            "phi": [[[0.4, 0.2], [0.6, 0.8]], [[0.1, 0.6], [0.9, 0.4]]],
                            "lambda": [fl(r) for r in lam], "Elnbeta": [fl(r) for r in Eb]}

# ---- test/common.jl:79-97 lambda_objective ; test/mmctm.jl:135-148 nu_objective ------------------------------
mu = [1, 1, 2, 2, 1]; lamv = [1, 2, 3, 4, 1]; nuv = [1, 1, 1, 2, 1]; zeta = [2, 1]
th = [[[mp.mpf("0.4"), mp.mpf("0.1")], [mp.mpf("0.6"), mp.mpf("0.9")]],
      [[mp.mpf("0.3"), mp.mpf("0.4")], [mp.mpf("0.3"), mp.mpf("0.5")], [mp.mpf("0.4"), mp.mpf("0.1")]]]
Xd = X_MM[0]
N = [13, 7]
sumth = []
for m in range(2):
    for k in range(len(th[m])):
        sumth.append(sum(th[m][k][w] * Xd[m][w][1] for w in range(2)))
Ndz = [mp.mpf(N[0]) / zeta[0]] * 2 + [mp.mpf(N[1]) / zeta[1]] * 3
Ee = [mp.exp(lamv[i] + mp.mpf(nuv[i]) / 2) for i in range(5)]
diff = [lamv[i] - mu[i] for i in range(5)]
Lval = -mp.mpf(1) / 2 * sum(d * d for d in diff) + sum(lamv[i] * sumth[i] for i in range(5)) - sum(Ndz[i] * Ee[i] for i in range(5))
Lgrad = [-diff[i] + sumth[i] - Ndz[i] * Ee[i] for i in range(5)]
out["lambda_objective"] = {"ref": "test/common.jl:79-97", "mu": mu, "lambda": lamv, "nu": nuv, "zeta": zeta,
                           "theta": [[[0.4, 0.1], [0.6, 0.9]], [[0.3, 0.4], [0.3, 0.5], [0.4, 0.1]]],
                           "invSigma": "I", "sumtheta": fl(sumth), "Ndivzeta": fl(Ndz),
                           "value": f(Lval), "grad": fl(Lgrad)}
Nval = -mp.mpf(1) / 2 * sum(nuv) - sum(Ndz[i] * Ee[i] for i in range(5)) + sum(mp.log(n) for n in nuv) / 2
Ngrad = [-mp.mpf(1) / 2 - Ndz[i] / 2 * Ee[i] + 1 / (2 * mp.mpf(nuv[i])) for i in range(5)]
out["nu_objective"] = {"ref": "test/mmctm.jl:135-148; test/immctm.jl:122-160", "mu": mu, "lambda": lamv, "nu": nuv,
                       "zeta": zeta, "invSigma": "I", "value": f(Nval), "grad": fl(Ngrad)}

# ---- test/mmctm.jl:59-90 Ndivzeta / sumtheta -----------------------------------------------------------------
out["calc_Ndivzeta"] = {"ref": "test/mmctm.jl:59-73", "zeta": [[2, 3], [4, 5]],
                        "doc1": fl([mp.mpf(13) / 2, mp.mpf(13) / 2, mp.mpf(7) / 3, mp.mpf(7) / 3, mp.mpf(7) / 3])}
out["calc_sumtheta"] = {"ref": "test/mmctm.jl:75-90", "theta_doc1": out["lambda_objective"]["theta"], "doc1": fl(sumth)}

# ---- test/mmctm.jl:158-166 update_zeta -----------------------------------------------------------------------
out["update_zeta"] = {"ref": "test/mmctm.jl:158-166", "lambda": [[1, 2, 3, 4, 1], [2, 3, 1, 4, 2]],
                      "nu": [[1, 1, 1, 2, 1], [1, 3, 1, 2, 1]],
                      "zeta_doc1": fl([mp.exp(mp.mpf("1.5")) + mp.exp(mp.mpf("2.5")),
                                       mp.exp(mp.mpf("3.5")) + mp.exp(5) + mp.exp(mp.mpf("1.5"))])}

# ---- test/mmctm.jl:168-209 update_theta ----------------------------------------------------------------------
gam = [[[1, 2, 2, 6], [2, 3, 1, 2]], [[1, 2, 3, 4], [2, 1, 2, 6], [1, 1, 3, 1]]]
t11 = [[mp.exp(1 + psi(1) - psi(11)), mp.exp(1 + psi(2) - psi(11))],
       [mp.exp(2 + psi(2) - psi(8)), mp.exp(2 + psi(3) - psi(8))]]
for w in range(2):
    s = t11[0][w] + t11[1][w]; t11[0][w] /= s; t11[1][w] /= s
t22 = [[mp.exp(1 + psi(3) - psi(10)), mp.exp(1 + psi(4) - psi(10))],
       [mp.exp(4 + psi(2) - psi(11)), mp.exp(4 + psi(6) - psi(11))],
       [mp.exp(2 + psi(3) - psi(6)), mp.exp(2 + psi(1) - psi(6))]]
for w in range(2):
    s = t22[0][w] + t22[1][w] + t22[2][w]
    for k in range(3):
        t22[k][w] /= s
out["update_theta"] = {"ref": "test/mmctm.jl:168-209", "lambda": [[1, 2, 3, 4, 1], [2, 3, 1, 4, 2]], "gamma": gam,
                       "theta_d1_m1": [fl(r) for r in t11], "theta_d2_m2": [fl(r) for r in t22]}

# ---- test/mmctm.jl:211-236 update_mu / update_Sigma ----------------------------------------------------------
lam2 = [[1, 2, 3, 4, 1], [2, 3, 1, 4, 2]]; nu2 = [[1, 1, 1, 2, 1], [1, 3, 1, 2, 1]]; mu2 = [1, 1, 2, 2, 1]
Sig = mp.zeros(5, 5)
for d in range(2):
    for i in range(5):
        Sig[i, i] += nu2[d][i]
    df = [lam2[d][i] - mu2[i] for i in range(5)]
    for i in range(5):
        for j in range(5):
            Sig[i, j] += df[i] * df[j]
Sig = Sig / 2
iSig = Sig ** -1
out["update_mu"] = {"ref": "test/mmctm.jl:211-218", "lambda": lam2, "mu": [1.5, 2.5, 2.0, 4.0, 1.5]}
out["update_Sigma"] = {"ref": "test/mmctm.jl:220-236", "lambda": lam2, "nu": nu2, "mu": mu2,
                       "Sigma": [[f(Sig[i, j]) for j in range(5)] for i in range(5)],
                       "invSigma": [[f(iSig[i, j]) for j in range(5)] for i in range(5)]}

# ---- test/mmctm.jl:238-257 update_gamma ----------------------------------------------------------------------
a = mp.mpf("0.1")
m_ = mp.mpf
out["update_gamma"] = {
    "ref": "test/mmctm.jl:238-257",
    "theta": {"d1m1": [[0.4, 0.1], [0.6, 0.9]], "d2m1": [[0.3, 0.5], [0.7, 0.5]],
              "d1m2": [[0.2, 0.6], [0.7, 0.3], [0.1, 0.1]], "d2m2": [[0.1, 0.3], [0.7, 0.5], [0.2, 0.2]]},
    "gamma_m1": [fl([a + 5 * m_("0.4"), a + 8 * m_("0.1"), a + 4 * m_("0.3"), a + 9 * m_("0.5")]),
                 fl([a + 5 * m_("0.6"), a + 8 * m_("0.9"), a + 4 * m_("0.7"), a + 9 * m_("0.5")])],
    "gamma_m2": [fl([a + 2 * m_("0.2"), a + 5 * m_("0.6"), a + 4 * m_("0.1"), a + 6 * m_("0.3")]),
                 fl([a + 2 * m_("0.7"), a + 5 * m_("0.3"), a + 4 * m_("0.7"), a + 6 * m_("0.5")]),
                 fl([a + 2 * m_("0.1"), a + 5 * m_("0.1"), a + 4 * m_("0.2"), a + 6 * m_("0.2")])]}

# ---- test/mmctm.jl:259-266 update_Elnphi ---------------------------------------------------------------------
out["update_Elnphi"] = {"ref": "test/mmctm.jl:259-266", "gamma_m1_k1": [1, 2, 1, 3], "Elnphi_111": f(psi(1) - psi(7))}

# ---- test/mmctm.jl:349-388 loglikelihoods --------------------------------------------------------------------
eta_ = [[1, 2], [2, 3]]
props = [[mp.exp(e) / sum(mp.exp(x) for x in ed) for e in ed] for ed in eta_]
gl = [[1, 2, 1, 3], [1, 1, 2, 4]]
phl = [[mp.mpf(x) / sum(gk) for x in gk] for gk in gl]
sum_ll = [5 * mp.log(props[0][0] * phl[0][0] + props[0][1] * phl[1][0]) + 8 * mp.log(props[0][0] * phl[0][1] + props[0][1] * phl[1][1]),
          4 * mp.log(props[1][0] * phl[0][2] + props[1][1] * phl[1][2]) + 9 * mp.log(props[1][0] * phl[0][3] + props[1][1] * phl[1][3])]
out["loglik_mmctm"] = {"ref": "test/mmctm.jl:349-388", "eta": eta_, "gamma_m1": gl,
                       "props": [fl(p) for p in props], "docmodality_ll_d1": f(sum_ll[0] / 13),
                       "modality_ll_m1": f((sum_ll[0] + sum_ll[1]) / 26)}

# ---- test/immctm.jl:53-79 constructor ------------------------------------------------------------------------
out["immctm_ctor"] = {"ref": "test/immctm.jl:53-79", "I": [2, 2], "J": [[2, 2], [2, 2]], "V": [4, 4], "N": [[13, 7], [13, 10]]}

# ---- test/immctm.jl:181-222 update_theta ---------------------------------------------------------------------
gim = [[[[0.1, 0.2], [0.1, 1.0]], [[0.1, 0.1], [1.0, 1.0]]],
       [[[0.5, 0.5], [1.0, 1.5]], [[1.0, 2.0], [2.0, 3.0]], [[1.0, 5.0], [5.0, 2.0]]]]
P = lambda s: psi(mp.mpf(s))
ti = [[mp.exp(1 + P("0.1") - P("0.3") + P("0.1") - P("1.1")), mp.exp(1 + P("0.1") - P("0.3") + P("1.0") - P("1.1"))],
      [mp.exp(2 + P("0.1") - P("0.2") + P("1.0") - P("2.0")), mp.exp(2 + P("0.1") - P("0.2") + P("1.0") - P("2.0"))]]
for w in range(2):
    s = ti[0][w] + ti[1][w]; ti[0][w] /= s; ti[1][w] /= s
ti2 = [[mp.exp(1 + P("0.5") - P("1.0") + P("1.0") - P("2.5")), mp.exp(1 + P("0.5") - P("1.0") + P("1.5") - P("2.5"))],
       [mp.exp(4 + P("2.0") - P("3.0") + P("2.0") - P("5.0")), mp.exp(4 + P("2.0") - P("3.0") + P("3.0") - P("5.0"))],
       [mp.exp(2 + P("5.0") - P("6.0") + P("5.0") - P("7.0")), mp.exp(2 + P("5.0") - P("6.0") + P("2.0") - P("7.0"))]]
for w in range(2):
    s = ti2[0][w] + ti2[1][w] + ti2[2][w]
    for k in range(3):
        ti2[k][w] /= s
out["immctm_update_theta"] = {"ref": "test/immctm.jl:181-222", "lambda": [[1, 2, 3, 4, 1], [2, 3, 1, 4, 2]], "gamma": gim,
                              "theta_d1_m1": [fl(r) for r in ti], "theta_d2_m2": [fl(r) for r in ti2]}

# ---- test/immctm.jl:251-261 update_gamma ; :263-270 update_Elnphi ----------------------------------------------
out["immctm_update_gamma"] = {"ref": "test/immctm.jl:251-261", "theta": {"d1m1": [[0.4, 0.1], [0.6, 0.9]], "d2m1": [[0.3, 0.5], [0.7, 0.5]]},
                              "gamma_m1_k1_i1": fl([a + 5 * m_("0.4") + 8 * m_("0.1"), a + 4 * m_("0.3") + 9 * m_("0.5")]),
                              "gamma_m1_k1_i2": fl([a + 5 * m_("0.4") + 4 * m_("0.3"), a + 8 * m_("0.1") + 9 * m_("0.5")])}
out["immctm_update_Elnphi"] = {"ref": "test/immctm.jl:263-270", "gamma_m1_k1_i1": [1, 2], "Elnphi_1111": f(psi(1) - psi(3))}

# ---- test/immctm.jl:350-386 loglikelihoods -------------------------------------------------------------------
thl = [[mp.exp(e) / sum(mp.exp(x) for x in ed) for e in ed] for ed in eta_]
gll = [[[m_("0.1"), m_("0.2")], [m_("0.1"), m_("1.0")]], [[m_("0.1"), m_("0.1")], [m_("1.0"), m_("1.0")]]]
pl = [[[x / sum(gi) for x in gi] for gi in gk] for gk in gll]   # [k][i][j]
sl = (5 * mp.log(thl[0][0] * pl[0][0][0] * pl[0][1][0] + thl[0][1] * pl[1][0][0] * pl[1][1][0]) +
      8 * mp.log(thl[0][0] * pl[0][0][0] * pl[0][1][1] + thl[0][1] * pl[1][0][0] * pl[1][1][1]) +
      4 * mp.log(thl[1][0] * pl[0][0][1] * pl[0][1][0] + thl[1][1] * pl[1][0][1] * pl[1][1][0]) +
      9 * mp.log(thl[1][0] * pl[0][0][1] * pl[0][1][1] + thl[1][1] * pl[1][0][1] * pl[1][1][1]))
out["loglik_immctm"] = {"ref": "test/immctm.jl:350-386", "eta": eta_, "gamma_m1": [[[0.1, 0.2], [0.1, 1.0]], [[0.1, 0.1], [1.0, 1.0]]],
                        "modality_ll_m1": f(sl / 26)}

# ---- digamma spot values used across the tests (SpecialFunctions.digamma must agree with any correct psi) ------
xs = ["1e-7", "0.05", "0.1", "0.2", "0.3", "0.5", "1", "1.1", "1.5", "2", "2.5", "3", "6.9999", "7", "7.0001", "11", "100.5", "1e4", "1e6", "93102.1"]
out["digamma"] = {"x": [float(mp.mpf(s)) for s in xs], "psi": [f(psi(mp.mpf(float(mp.mpf(s))))) for s in xs]}
xs2 = ["0.1", "0.4", "1", "9.6", "96.5", "23029.1"]
out["lgamma"] = {"x": [float(mp.mpf(s)) for s in xs2], "lgamma": [f(mp.loggamma(mp.mpf(float(mp.mpf(s))))) for s in xs2]}

# ---- α_objective (common.jl:38-46), test/mmctm.jl:268-279 and test/immctm.jl:273-284: the test evaluates
# L = K (lgamma(V α) - V lgamma(α)) + α ΣElnϕ and dL/dα = K V (ψ(V α) - ψ(α)) + ΣElnϕ for the toy model (K = 2, V = 4,
# α = 0.1; IMMCTM: K = 2, J = 2); ΣElnϕ depends on the random init there, so fixed values are used here
cases = []
for (al, s_, K_, V_) in [("0.1", "-25.5", 2, 4), ("0.1", "-9.25", 2, 2), ("0.37", "-40.125", 3, 4), ("2.5", "-310.0", 7, 96), ("1e-3", "-50.0", 10, 48)]:
    a_ = mp.mpf(float(mp.mpf(al))); ss = mp.mpf(float(mp.mpf(s_)))
    L = K_ * (mp.loggamma(V_ * a_) - V_ * mp.loggamma(a_)) + a_ * ss
    g = K_ * V_ * (psi(V_ * a_) - psi(a_)) + ss
    cases.append({"alpha": float(a_), "sum_Elnphi": float(ss), "K": K_, "V": V_, "L": f(L), "grad": f(g)})
out["alpha_objective"] = {"ref": "test/mmctm.jl:268-279; test/immctm.jl:273-284; common.jl:38-46", "cases": cases}

# ---- ILDA (test/ilda.jl): features 4 x 2, X = [[1 5; 2 8], [3 2; 4 5]], K = 2 ------------------------------------
ILDA_FEATURES = [[1, 1], [1, 2], [2, 1], [2, 2]]
X_ILDA = [[[1, 5], [2, 8]], [[3, 2], [4, 5]]]
Elnth = [[mp.mpf("0.5"), mp.mpf("-1.1")], [mp.mpf("2.3"), mp.mpf("-0.7")]]                # [k][d]
Elnb = [[[mp.mpf("-0.2"), mp.mpf("-0.9")], [mp.mpf("-1.1"), mp.mpf("0.3")]],              # [i][j][k]
        [[mp.mpf("0.5"), mp.mpf("0.1")], [mp.mpf("-0.1"), mp.mpf("-0.4")]]]
phis = []
for d in range(2):
    cols = []
    for w in range(2):
        v = X_ILDA[d][w][0] - 1
        col = [mp.e ** (Elnth[k][d] + sum(Elnb[i][ILDA_FEATURES[v][i] - 1][k] for i in range(2))) for k in range(2)]
        tot = sum(col)
        cols.append([c / tot for c in col])
    phis.append([[f(cols[w][k]) for w in range(2)] for k in range(2)])                    # K x W
out["ilda_update_phi"] = {"ref": "test/ilda.jl:53-93", "Elntheta": [[0.5, -1.1], [2.3, -0.7]],
                          "Elnbeta": [[[-0.2, -0.9], [-1.1, 0.3]], [[0.5, 0.1], [-0.1, -0.4]]], "phi": phis}
eta_t = [mp.mpf("0.1"), mp.mpf("0.2")]
phi_t = [[[mp.mpf("0.4"), mp.mpf("0.2")], [mp.mpf("0.6"), mp.mpf("0.8")]], [[mp.mpf("0.1"), mp.mpf("0.6")], [mp.mpf("0.9"), mp.mpf("0.4")]]]   # [d][k][w]
lam_i = []
Eln_i = []
for i in range(2):
    lam = [[eta_t[i] for _ in range(2)] for _ in range(2)]                                # [j][k]
    for d in range(2):
        for w in range(2):
            v = X_ILDA[d][w][0] - 1; n = X_ILDA[d][w][1]
            j = ILDA_FEATURES[v][i] - 1
            for k in range(2):
                lam[j][k] += phi_t[d][k][w] * n
    lam_i.append([[f(lam[j][k]) for k in range(2)] for j in range(2)])
    Eln_i.append([[f(psi(lam[j][k]) - psi(lam[0][k] + lam[1][k])) for k in range(2)] for j in range(2)])
out["ilda_update_lambda"] = {"ref": "test/ilda.jl:114-160", "eta": [0.1, 0.2], "phi": [[[0.4, 0.2], [0.6, 0.8]], [[0.1, 0.6], [0.9, 0.4]]],
                             "lambda": lam_i, "Elnbeta": Eln_i}
g_t = [mp.mpf("0.1") + mp.mpf("0.4") * 5 + mp.mpf("0.2") * 8, mp.mpf("0.1") + mp.mpf("0.6") * 5 + mp.mpf("0.8") * 8]
out["ilda_update_gamma"] = {"ref": "test/ilda.jl:95-112", "phi_doc1": [[0.4, 0.2], [0.6, 0.8]], "gamma_doc1": fl(g_t),
                            "Elntheta_doc1": fl([psi(g_t[0]) - psi(g_t[0] + g_t[1]), psi(g_t[1]) - psi(g_t[0] + g_t[1])])}
out["corpora"]["X_ilda"] = X_ILDA
out["corpora"]["features_ilda"] = ILDA_FEATURES

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "reference_kats.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print("wrote", os.path.join(here, "reference_kats.json"), len(out), "entries")
