"""Top-K ranking metrics computed from the fused top-K output of ``predict_topk`` (host side, numpy).

The reference computes its metrics from the full score matrix: ``MetricTopK.calculate`` drops the users without
true interactions (recpack/metrics/base.py:97-118), takes per-user top-K ranks (``get_top_K_ranks``,
recpack/util.py:50-77) and evaluates NDCGK (recpack/metrics/dcg.py:98-137), RecallK / CalibratedRecallK
(recpack/metrics/recall.py:34-44, 75-85) and CoverageK (recpack/metrics/coverage.py:30-36).  Here the kernel has
already produced the K best items per user, so the same numbers follow from I x K values instead of U x I.

Rank order inside a user's list: larger score first, then smaller item id (the library's deterministic order; the
reference's order among equal scores is whatever numpy's argpartition leaves).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def topk_ranks(topk: sp.csr_matrix) -> sp.csr_matrix:
    """CSR with the same pattern as ``topk`` (one row per user, <= K stored scores) holding ranks 1..n."""
    topk = sp.csr_matrix(topk)
    n = np.diff(topk.indptr)
    rows = np.repeat(np.arange(topk.shape[0]), n)
    order = np.lexsort((topk.indices, -topk.data, rows))  # by user, score desc, item asc
    start = np.repeat(topk.indptr[:-1], n)
    ranks = np.empty(topk.nnz, dtype=np.int64)
    ranks[order] = np.arange(topk.nnz) - start + 1
    return sp.csr_matrix((ranks, topk.indices.copy(), topk.indptr.copy()), shape=topk.shape)


def _prepare(y_true, topk):
    y_true = sp.csr_matrix(y_true)
    y_true = sp.csr_matrix((np.ones(y_true.nnz), y_true.indices, y_true.indptr), shape=y_true.shape)
    y_true.eliminate_zeros()
    topk = sp.csr_matrix(topk)
    if y_true.shape != topk.shape:
        raise AssertionError(f"shape mismatch between y_true {y_true.shape} and y_pred {topk.shape}")  # metrics/base.py:73-75
    users = np.diff(y_true.indptr) > 0  # _eliminate_empty_users, metrics/base.py:97-118
    return y_true[users], topk[users]


def ndcg_k(y_true, topk, K: int) -> float:
    """metrics/dcg.py:98-137: mean over users with true interactions of DCG / IDCG(min(K, |true|))."""
    y_true, topk = _prepare(y_true, topk)
    ranks = topk_ranks(topk)
    hits = ranks.multiply(y_true).tocsr()
    hits.data = 1.0 / np.log2(hits.data + 1.0)
    dcg = np.asarray(hits.sum(axis=1)).ravel()
    discount = 1.0 / np.log2(np.arange(2, K + 2))
    idcg = np.concatenate([[1.0], np.cumsum(discount)])
    hist = np.minimum(np.diff(y_true.indptr), K)
    return float(np.mean(dcg / idcg[hist])) if len(dcg) else 0.0


def recall_k(y_true, topk, K: int) -> float:
    """metrics/recall.py:34-44: hits / |true| per user, averaged."""
    y_true, topk = _prepare(y_true, topk)
    hits = np.asarray(topk.astype(bool).multiply(y_true).sum(axis=1)).ravel()
    return float(np.mean(hits / np.diff(y_true.indptr))) if len(hits) else 0.0


def calibrated_recall_k(y_true, topk, K: int) -> float:
    """metrics/recall.py:75-85: hits / min(K, |true|) per user, averaged."""
    y_true, topk = _prepare(y_true, topk)
    hits = np.asarray(topk.astype(bool).multiply(y_true).sum(axis=1)).ravel()
    return float(np.mean(hits / np.minimum(np.diff(y_true.indptr), K))) if len(hits) else 0.0


def coverage_k(y_true, topk, K: int) -> float:
    """metrics/coverage.py:30-36: share of the items that appear in some user's top-K."""
    y_true, topk = _prepare(y_true, topk)
    return len(np.unique(topk.indices[topk.data != 0])) / topk.shape[1]
