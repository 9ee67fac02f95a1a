#!/usr/bin/env python3
"""Golden trajectories of the CPU oracle on the reference's shipped BRCA-EU tables (BASELINE configs 1 and 3) with fixed
initialisations: per-pass log-likelihoods, ELBO and a few parameter checksums.  The reference's own tests pin no full-fit
value, and the Julia reference cannot run here, so these are the build's OWN golden vectors: they pin the oracle
against accidental change (tests/test_oracle_kats.py) and give the GPU tests a committed target that does not need the oracle
build at all (tests/test_brca_gpu.py).  Regenerate with: python tests/golden/make_trajectories.py"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as orc


def tables():
    def read(path):
        with open(path) as fh:
            header = fh.readline().rstrip("\n").split("\t")
            rows = [[int(float(x)) for x in line.rstrip("\n").split("\t")[1:]] for line in fh if "\t" in line]
        return header[1:], np.asarray(rows, dtype=np.int64)
    samples, snv = read(os.path.join(HERE, "brca-eu_snv_counts.tsv"))
    _, sv = read(os.path.join(HERE, "brca-eu_sv_counts.tsv"))
    return samples, snv, sv


def docs(mat):
    out = []
    for d in range(mat.shape[1]):
        nz = np.nonzero(mat[:, d])[0]
        out.append(np.stack([nz + 1, mat[nz, d]], axis=1).astype(np.int64).reshape(-1, 2))
    return out


samples, snv, sv = tables()
out = {"note": "oracle-generated (oracle/mmm_oracle.c); seeds are numpy default_rng seeds of the initialisations"}
# config 1: LDA K = 7, alpha = eta = 0.1, SNV table, lambda0 = default_rng(1).integers(1, 101, (96, 7))
X1 = docs(snv)
lam0 = np.random.default_rng(1).integers(1, 101, size=(96, 7)).astype(np.float64)
o = orc.LdaOracle(7, 0.1, 0.1, X1, V=96, lambda0=lam0)
ll = o.fit(maxiter=60, tol=1e-4)
out["config1_lda_k7"] = {"lambda0_seed": 1, "maxiter": 60, "tol": 1e-4, "ll": [float(x) for x in ll], "converged": bool(o.converged),
                         "elbo": float(o.elbo_value), "beta_colsum_check": float(np.abs(o.beta.reshape(96, 7, order="F")).sum()),
                         "lambda_sum": float(o.lam.sum()), "theta_first_doc": [float(x) for x in o.theta[:7]]}
# config 3: MMCTM K = [7, 7], alpha = [0.1, 0.1], SNV + SV, gamma0 from default_rng(2)
X3 = [[a, b] for a, b in zip(docs(snv), docs(sv))]
rng = np.random.default_rng(2)
g0 = [rng.integers(1, 101, size=(7, 96)).astype(np.float64), rng.integers(1, 101, size=(7, 48)).astype(np.float64)]
c = orc.CtmOracle([7, 7], [0.1, 0.1], X3, V=[96, 48], gamma0=np.concatenate([x.ravel() for x in g0]))
llc = c.fit(maxiter=12, tol=0.0)
out["config3_mmctm_77"] = {"gamma0_seed": 2, "maxiter": 12, "ll": [[float(x) for x in row] for row in llc], "elbo": float(c.elbo_value),
                           "mu": [float(x) for x in c.mu], "gamma_sum": float(c.gamma.sum())}
# the same fit by the order-matched variant (oracle/mmm_twin.c) in the launch geometry the library picks for 560 documents with
# sum K = 14 (16 lanes per document, 8-wave theta blocks -> 18 blocks; 18 moment blocks): the device reproduces it bit for bit
GEO3 = {"L": 16, "grid_e": 18, "waves_e": 8, "grid_m": 18}
t = orc.CtmOracle([7, 7], [0.1, 0.1], X3, V=[96, 48], gamma0=np.concatenate([x.ravel() for x in g0]), geometry=GEO3)
llt = t.fit(maxiter=12, tol=0.0)
out["config3_mmctm_77_device_order"] = {
    "geometry": GEO3, "maxiter": 12, "ll": [[float(x) for x in row] for row in llt], "elbo": float(t.elbo_value),
    "mu": [float(x) for x in t.mu], "invSigma_diag": [float(x) for x in t.invSigma.reshape(14, 14).diagonal()],
    "gamma_first_topic": [float(x) for x in t.gamma[:96]], "gamma_sum": float(t.gamma.sum()),
    "lambda_doc0": [float(x) for x in t.lam[:14]], "nu_doc0": [float(x) for x in t.nu[:14]],
    "lambda_sum": float(t.lam.sum()), "nu_sum": float(t.nu.sum()),
    "nev_nu_last_pass": [int(x) for x in t.nev_nu[:560]], "nev_lambda_last_pass": [int(x) for x in t.nev_lambda[:560]],
    "fork_vs_index_order_ll_rel": [float(x) for x in (np.abs(llt - llc).max(axis=1) / np.abs(llc).max(axis=1))]}
with open(os.path.join(HERE, "oracle_trajectories.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote oracle_trajectories.json", len(ll), "LDA passes")
