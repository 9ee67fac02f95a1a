"""Result tables of scripts/run_mmctm.jl:184-245 (`topicdf`/`writesigs`, `propdf`/`writeprops`, `cov2cor`, `writedlm` of μ, Σ)
without DataFrames: plain TSV written from the model's fields.  The input side (TSV -> count matrices -> CSR) is in utils.py
(`read_counts_tsv`, `format_counts_*`, `pack_lda`, `pack_mm`)."""
import numpy as np


def cov2cor(C):
    """run_mmctm.jl:184-187"""
    C = np.asarray(C, dtype=np.float64)
    sigma = np.sqrt(np.diag(C))
    return C / np.outer(sigma, sigma)


def topic_table(model, terms, modalities):
    """`topicdf` (run_mmctm.jl:189-210): rows (modality, topic, value, term, probability), probability = γ[m][k] / sum γ[m][k];
    topic and value are 1-based as in the reference's output."""
    rows = []
    for m in range(model.M):
        for k in range(model.K[m]):
            g = np.asarray(model.γ[m][k], dtype=np.float64)
            probs = g / g.sum()
            for v in range(model.V[m]):
                rows.append((modalities[m], k + 1, v + 1, terms[m][v], float(probs[v])))
    return rows


def props_table(model, samples, modalities):
    """`propdf` (run_mmctm.jl:217-240): (labels "modality-k", [sum K, D] matrix of softmax(λ block) per document)."""
    lam = model.lam_matrix()                       # [D, MK]
    out = np.empty((lam.shape[1], lam.shape[0]))
    start = 0
    for m in range(model.M):
        stop = start + model.K[m]
        e = np.exp(lam[:, start:stop])
        out[start:stop, :] = (e / e.sum(axis=1, keepdims=True)).T
        start = stop
    labels = ["%s-%d" % (modalities[m], k + 1) for m in range(model.M) for k in range(model.K[m])]
    if len(samples) != lam.shape[0]:
        raise ValueError("%d sample names for %d documents" % (len(samples), lam.shape[0]))
    return labels, out


def _fmt(x):
    return repr(float(x))


def write_sigs(filename, model, terms, modalities):
    """`writesigs` (run_mmctm.jl:212-215)"""
    with open(filename, "w") as fh:
        fh.write("modality\ttopic\tvalue\tterm\tprobability\n")
        for r in topic_table(model, terms, modalities):
            fh.write("%s\t%d\t%d\t%s\t%s\n" % (r[0], r[1], r[2], r[3], _fmt(r[4])))


def write_props(filename, model, samples, modalities):
    """`writeprops` (run_mmctm.jl:242-245)"""
    labels, P = props_table(model, samples, modalities)
    with open(filename, "w") as fh:
        fh.write("topic\t" + "\t".join(str(s) for s in samples) + "\n")
        for i, lab in enumerate(labels):
            fh.write(lab + "\t" + "\t".join(_fmt(x) for x in P[i]) + "\n")


def write_matrix(filename, A):
    """`writedlm(filename, A)` (run_mmctm.jl:276-283): tab-delimited, one row per line; a vector is written as a column."""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        A = A[:, None]
    with open(filename, "w") as fh:
        for row in A:
            fh.write("\t".join(_fmt(x) for x in row) + "\n")
