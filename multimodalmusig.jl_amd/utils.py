"""Count formatting -- host-side mirror of src/utils.jl (make_count_matrix :1-7, format_counts_lda :9-18,
format_counts_ctm :20-22, format_counts_mmctm :24-36) plus the CSR packing the C ABI takes.

`countsdf` is anything column-indexable by name (pandas.DataFrame, dict of arrays); a column is one sample,
rows are vocabulary terms.  The result keeps the reference's shape: a (W x 2) int64 matrix per document with
1-based term ids in column 0 and counts in column 1, zero-count terms dropped.
"""
import numpy as np


def make_count_matrix(counts):
    counts = np.asarray(counts)
    idx = np.nonzero(counts > 0)[0]
    out = np.empty((idx.size, 2), dtype=np.int64)
    out[:, 0] = idx + 1
    out[:, 1] = counts[idx]
    return out


def format_counts_lda(countsdf, cols):
    return [make_count_matrix(np.asarray(countsdf[c])) for c in cols]


def format_counts_mmctm(countdfs, cols):
    return [[make_count_matrix(np.asarray(df[c])) for df in countdfs] for c in cols]


def format_counts_ctm(countsdf, cols):
    return format_counts_mmctm([countsdf], cols)


def read_counts_tsv(path):
    """Read a `term<TAB>sample1<TAB>...` count table (the layout of the reference's data/*.tsv) without
    pandas: returns (terms, sample_names, counts[V, D] int64)."""
    with open(path) as fh:
        header = fh.readline().rstrip("\n").split("\t")
        terms, rows = [], []
        for line in fh:
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 2:
                continue
            terms.append(parts[0]); rows.append([int(float(x)) for x in parts[1:]])
    return terms, header[1:], np.asarray(rows, dtype=np.int64)


# ---- CSR packing (what include/mmmusig.h takes) ---------------------------------------------------------------
def _as_doc(x):
    return np.asarray(x, dtype=np.int64).reshape(-1, 2)


def pack_lda(X):
    """X[d] (W_d x 2, 1-based) -> doc_ptr int64[D+1], term int32[nnz] (0-based), count int32[nnz]."""
    D = len(X)
    docs = [_as_doc(x) for x in X]
    doc_ptr = np.zeros(D + 1, dtype=np.int64)
    if D:
        doc_ptr[1:] = np.cumsum([x.shape[0] for x in docs])
    if doc_ptr[-1]:
        allx = np.concatenate(docs, axis=0)
        if allx[:, 0].min() < 1 or allx[:, 1].min() < 0 or allx.max() >= 2 ** 31:
            raise ValueError("term ids must be >= 1, counts >= 0 and both < 2^31")
        term = (allx[:, 0] - 1).astype(np.int32); count = allx[:, 1].astype(np.int32)
    else:
        term = np.zeros(0, dtype=np.int32); count = np.zeros(0, dtype=np.int32)
    return doc_ptr, np.ascontiguousarray(term), np.ascontiguousarray(count)


def pack_mm(X, M):
    """X[d][m] -> modality-major concatenation; doc_ptr is M*(D+1) absolute offsets."""
    D = len(X)
    doc_ptr = np.zeros(M * (D + 1), dtype=np.int64)
    terms, counts = [], []
    base = 0
    for m in range(M):
        dp, t, c = pack_lda([X[d][m] for d in range(D)])
        doc_ptr[m * (D + 1):(m + 1) * (D + 1)] = dp + base
        base += int(dp[-1])
        terms.append(t); counts.append(c)
    term = np.concatenate(terms) if terms else np.zeros(0, np.int32)
    count = np.concatenate(counts) if counts else np.zeros(0, np.int32)
    return doc_ptr, np.ascontiguousarray(term, dtype=np.int32), np.ascontiguousarray(count, dtype=np.int32)


def shard_documents(X, nranks, rank):
    """Contiguous block of documents for `rank`, balanced by number of nonzeros: returns
    (d0, d1).  Works for LDA (X[d] matrices) and MMCTM (X[d][m]) nestings."""
    D = len(X)

    def nz(x):
        if isinstance(x, (list, tuple)):
            return sum(_as_doc(xx).shape[0] for xx in x)
        return _as_doc(x).shape[0]

    w = np.array([nz(x) + 1 for x in X], dtype=np.float64)   # +1: empty documents still cost a wave
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [int(np.searchsorted(cum, cum[-1] * r / nranks, side="left")) for r in range(nranks + 1)]
    bounds[0], bounds[-1] = 0, D
    for r in range(1, nranks + 1):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return bounds[rank], bounds[rank + 1]
