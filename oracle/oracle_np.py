"""CPU ORACLE (numpy/scipy) for the item-similarity hot path of LienM/recpack v0.3.6.

TEST INFRASTRUCTURE ONLY.  Nothing under ``recpack_amd/`` may import this module: it is the
checker for ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference from /root/reference in
the build container, runs ``ItemKNN``/``EASE`` ``fit``/``predict`` on the reference's own test
fixtures and on seeded synthetic matrices, and stores inputs + outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function here against those vectors.

Each function restates one step of the reference (file:line relative to /root/reference):

* ``binarize``            recpack/util.py:99-109 (``to_binary``) via recpack/matrix/util.py:27-48
* ``item_degrees``        diag(X^T X); the l2 norms sklearn ``normalize`` uses inside
                          ``cosine_similarity`` (recpack/algorithms/nearest_neighbour.py:80)
* ``gram_counts``         X^T X as integer co-occurrence counts (nearest_neighbour.py:80, ease.py:81)
* ``itemknn_fit``         compute_cosine_similarity + setdiag(0) + get_top_K_values
                          (nearest_neighbour.py:69-84, :204-224; recpack/util.py:50-96)
* ``predict_full``        ItemSimilarityMatrixAlgorithm._predict, ``X @ similarity_matrix_``
                          (recpack/algorithms/base.py:237-255)
* ``topk_rows``           per-row top-K as used on predictions by the metrics
                          (recpack/util.py:50-77, recpack/metrics/base.py:189)
* ``ease_fit``            EASE._fit (recpack/algorithms/ease.py:63-95)

Deterministic ordering (the reference's own tie order is an artefact of numpy's introselect and of
scipy's unsorted SpGEMM output): larger key first, then SMALLER column index.
For ItemKNN the selection key of entry (i, j) is the IEEE double ``fl(c_ij*c_ij) * fl(1/c_j)``
(monotone in cosine for a fixed row i, computed from integers only), the stored value is
``c_ij / sqrt(c_i*c_j)`` in float64.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


# --------------------------------------------------------------------------------------
# input handling
# --------------------------------------------------------------------------------------
def binarize(X: sp.csr_matrix) -> sp.csr_matrix:
    """recpack/util.py:99-109: non-zero stored values -> 1 (explicit zeros are dropped)."""
    X = sp.csr_matrix(X)
    X = X.copy()
    X.sum_duplicates()
    X.eliminate_zeros()
    X.sort_indices()
    return sp.csr_matrix((np.ones(X.nnz, dtype=np.int64), X.indices, X.indptr), shape=X.shape)


def _libm_pow(x: np.ndarray, y: float) -> np.ndarray:
    """x ** y through the C library's pow, element by element: numpy's vectorised power may differ from libm in the last
    bit, and the selection key c_ij * c_j^-alpha (exact ties are common) must be the same double everywhere."""
    import math

    return np.array([math.pow(float(v), y) for v in np.asarray(x).ravel()], dtype=np.float64).reshape(np.shape(x))


def item_degrees(Xb: sp.csr_matrix) -> np.ndarray:
    return np.bincount(Xb.indices, minlength=Xb.shape[1]).astype(np.int64)


def gram_counts(Xb: sp.csr_matrix, rows=None) -> sp.csr_matrix:
    """Integer co-occurrence counts c_ij = |users(i) & users(j)|; ``rows`` restricts the item rows."""
    Xt = Xb.T.tocsr()
    if rows is not None:
        Xt = Xt[rows]
    G = (Xt @ Xb).tocsr()
    G.sort_indices()
    return G


# --------------------------------------------------------------------------------------
# top-K with the build's deterministic order
# --------------------------------------------------------------------------------------
def _segment_topk(indptr, cols, keys, K):
    """Return (sel_positions sorted by row then rank, row_counts) for top-K per CSR row.

    Order inside a row: key descending, then column ascending.
    """
    n_rows = indptr.shape[0] - 1
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(indptr))
    order = np.lexsort((cols, -keys, rows))  # last key is primary
    # rank inside row
    start = indptr[:-1][rows[order]]
    rank = np.arange(order.shape[0], dtype=np.int64) - start
    keep = rank < K
    sel = order[keep]
    cnt = np.minimum(np.diff(indptr), K).astype(np.int32)
    return sel, cnt


def itemknn_fit_normalized(X: sp.csr_matrix, K: int, similarity: str = "cosine", pop_discount=None):
    """ItemKNN with ``normalize_X=True`` (nearest_neighbour.py:207-210): the binarised rows are l1-normalised first,
    X[u, i] = w_u = 1/d_u.  Cosine (:69-84) then is G_ij / (n_i n_j) with G_ij = sum_{u in i&j} w_u^2 and n_i = sqrt(G_ii);
    conditional probability (:22-66: ``to_binary(X).T @ X``, binary frequencies) is G'_ij / (c_i c_j^alpha) with
    G'_ij = sum_{u in i&j} w_u.  Same return layout and rank order as ``itemknn_fit``.  Sums are float64 in scipy's
    order; the kernels add the same float64 terms in another order (last-bit differences).  ``K=None`` returns every
    candidate of the similarity as a CSR matrix instead."""
    Xb = binarize(X)
    n_items = Xb.shape[1]
    c = item_degrees(Xb)
    d = np.diff(Xb.indptr).astype(np.float64)
    with np.errstate(divide="ignore"):
        w = np.where(d > 0, 1.0 / d, 0.0)
    q = w * w if similarity == "cosine" else w
    Xq = sp.csr_matrix((np.repeat(q, np.diff(Xb.indptr)), Xb.indices, Xb.indptr), shape=Xb.shape)
    G = (Xb.T.tocsr() @ Xq).tocsr()
    G.sort_indices()
    gi = np.repeat(np.arange(n_items, dtype=np.int64), np.diff(G.indptr))
    gj = G.indices.astype(np.int64)
    gv = G.data.astype(np.float64)
    if similarity == "cosine":
        n2 = np.asarray(Xq.sum(axis=0)).ravel()  # G_ii
        with np.errstate(divide="ignore"):
            colw = np.where(n2 > 0, 1.0 / np.sqrt(n2), 0.0)
        roww = colw
    elif similarity == "conditional_probability":
        alpha = float(pop_discount or 0.0)
        colw = np.zeros(n_items)
        nz = c > 0
        colw[nz] = 1.0 if alpha == 0.0 else _libm_pow(c[nz], -alpha)
        with np.errstate(divide="ignore"):
            roww = np.where(c > 0, 1.0 / c.astype(np.float64), 0.0)
    else:
        raise ValueError(similarity)
    off = (gi != gj) & (gv != 0)
    cnt = np.bincount(gi[off], minlength=n_items)
    indptr = np.zeros(n_items + 1, dtype=np.int64)
    np.cumsum(cnt, out=indptr[1:])
    gi, gj, gv = gi[off], gj[off], gv[off]
    key = gv * colw[gj]
    if K is None:  # every candidate of the similarity, as CSR
        return sp.csr_matrix((key * roww[gi], gj, indptr), shape=(n_items, n_items))
    sel, nsel = _segment_topk(indptr, gj, key, K)
    val = key[sel] * roww[gi[sel]]
    idx = np.full((n_items, K), -1, dtype=np.int32)
    out = np.zeros((n_items, K), dtype=np.float64)
    r = np.repeat(np.arange(n_items), nsel)
    pos = np.arange(sel.shape[0]) - np.repeat(np.concatenate([[0], np.cumsum(nsel)[:-1]]), nsel)
    idx[r, pos] = gj[sel]
    out[r, pos] = val
    return {"idx": idx, "val": out, "nnz": nsel.astype(np.int32), "deg": c, "rows": np.arange(n_items, dtype=np.int64)}


def itemknn_fit(X: sp.csr_matrix, K: int, rows=None, similarity: str = "cosine", pop_discount=None):
    """ItemKNN (cosine, or conditional probability nearest_neighbour.py:22-66 on the binarised matrix).
    Returns dict(idx int32[R,K] (-1 pad), val float64[R,K] (0 pad), nnz int32[R], deg int64[I]).

    Rows of ``idx``/``val`` are in rank order (best neighbour first).  Conditional probability:
    sim(i,j) = c_ij / (c_i * c_j^alpha), alpha = pop_discount or 0; key = fl(c_ij) * c_j^-alpha, value = key / c_i.
    """
    Xb = binarize(X)
    n_items = Xb.shape[1]
    c = item_degrees(Xb)
    row_ids = np.arange(n_items, dtype=np.int64) if rows is None else np.asarray(rows, dtype=np.int64)
    G = gram_counts(Xb, None if rows is None else row_ids)
    R = row_ids.shape[0]
    gi = np.repeat(row_ids, np.diff(G.indptr))
    gj = G.indices.astype(np.int64)
    gc = G.data.astype(np.int64)
    off = gi != gj  # setdiag(0): the diagonal never survives get_top_K_values
    # rebuild CSR without the diagonal
    cnt = np.bincount(np.repeat(np.arange(R), np.diff(G.indptr))[off], minlength=R)
    indptr = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(cnt, out=indptr[1:])
    gi, gj, gc = gi[off], gj[off], gc[off]
    if similarity == "cosine":
        with np.errstate(divide="ignore"):
            inv = 1.0 / c.astype(np.float64)
        key = (gc * gc).astype(np.float64) * inv[gj]
    elif similarity == "conditional_probability":
        alpha = float(pop_discount or 0.0)
        colw = np.zeros(n_items)
        nz = c > 0
        colw[nz] = 1.0 if alpha == 0.0 else _libm_pow(c[nz], -alpha)
        key = gc.astype(np.float64) * colw[gj]
    else:
        raise ValueError(similarity)
    sel, nsel = _segment_topk(indptr, gj, key, K)
    if similarity == "cosine":
        val = gc[sel].astype(np.float64) / np.sqrt(c[gi[sel]].astype(np.float64) * c[gj[sel]].astype(np.float64))
    else:
        val = key[sel] / c[gi[sel]].astype(np.float64)
    idx = np.full((R, K), -1, dtype=np.int32)
    out = np.zeros((R, K), dtype=np.float64)
    r = np.repeat(np.arange(R), nsel)
    pos = np.arange(sel.shape[0]) - np.repeat(np.concatenate([[0], np.cumsum(nsel)[:-1]]), nsel)
    idx[r, pos] = gj[sel]
    out[r, pos] = val
    return {"idx": idx, "val": out, "nnz": nsel.astype(np.int32), "deg": c, "rows": row_ids}


def topk_to_csr(idx: np.ndarray, val: np.ndarray, nnz: np.ndarray, n_cols: int, normalize_l1: bool = False):
    """(idx,val,nnz) -> canonical CSR float64; ``normalize_l1`` = Normalizer('l1') of nearest_neighbour.py:220-222."""
    R, K = idx.shape
    mask = np.arange(K)[None, :] < nnz[:, None]
    v = val.astype(np.float64)
    if normalize_l1:
        s = np.where(mask, np.abs(v), 0).sum(axis=1, keepdims=True)
        s[s == 0] = 1.0
        v = v / s
    indptr = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(nnz, out=indptr[1:])
    S = sp.csr_matrix((v[mask], idx[mask], indptr), shape=(R, n_cols))
    S.sort_indices()
    return S


def predict_full(X: sp.csr_matrix, S) -> sp.csr_matrix:
    """base.py:237-255: scores = binarised X @ similarity, returned as CSR float64."""
    Xb = binarize(X).astype(np.float64)
    scores = Xb @ S
    if not isinstance(scores, sp.csr_matrix):
        scores = sp.csr_matrix(scores)
    scores.sort_indices()
    return scores


def topk_rows(scores: sp.csr_matrix, K: int, exclude: sp.csr_matrix | None = None, drop_nonpositive: bool = False):
    """Per-row top-K of stored scores (optionally with the history entries removed first,
    recpack/pipelines/pipeline.py:174-175). Returns idx int32[U,K] (-1 pad), val float64[U,K], nnz int32[U]."""
    scores = sp.csr_matrix(scores).copy()
    scores.sort_indices()
    U = scores.shape[0]
    rows = np.repeat(np.arange(U, dtype=np.int64), np.diff(scores.indptr))
    cols = scores.indices.astype(np.int64)
    vals = scores.data.astype(np.float64)
    keep = np.ones(vals.shape[0], dtype=bool)
    if exclude is not None:
        ex = binarize(exclude)
        ekey = np.repeat(np.arange(U, dtype=np.int64), np.diff(ex.indptr)) * scores.shape[1] + ex.indices
        keep &= ~np.isin(rows * scores.shape[1] + cols, ekey)
    if drop_nonpositive:
        keep &= vals > 0
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    cnt = np.bincount(rows, minlength=U)
    indptr = np.zeros(U + 1, dtype=np.int64)
    np.cumsum(cnt, out=indptr[1:])
    sel, nsel = _segment_topk(indptr, cols, vals, K)
    idx = np.full((U, K), -1, dtype=np.int32)
    out = np.zeros((U, K), dtype=np.float64)
    r = np.repeat(np.arange(U), nsel)
    pos = np.arange(sel.shape[0]) - np.repeat(np.concatenate([[0], np.cumsum(nsel)[:-1]]), nsel)
    idx[r, pos] = cols[sel]
    out[r, pos] = vals[sel]
    return idx, out, nsel.astype(np.int32)


# --------------------------------------------------------------------------------------
# reference-SHAPED restatements: the same third-party calls in the same order as the reference makes them (the
# vectorised functions above give the same sets faster; these are what bench.py's cpu_baseline times as "reference
# shaped", and tests/test_oracle_golden.py checks them against the same goldens)
# --------------------------------------------------------------------------------------
def topk_per_row_loop(Xs: sp.csr_matrix, K: int) -> sp.csr_matrix:
    """recpack/util.py:50-96 (get_top_K_ranks + get_top_K_values): a Python loop over the rows, np.argpartition with
    the full list of the last K positions on every row's data, the K stored values kept.  Tie order is numpy's."""
    Xs = sp.csr_matrix(Xs)
    rows, cols, vals = [], [], []
    for r, (le, ri) in enumerate(zip(Xs.indptr[:-1], Xs.indptr[1:])):
        k = min(K, ri - le)
        if k:
            pick = le + np.argpartition(Xs.data[le:ri], list(range(-k, 0)))[-k:]
            rows.append(np.full(k, r, dtype=np.int64))
            cols.append(Xs.indices[pick])
            vals.append(Xs.data[pick])
    if not rows:
        return sp.csr_matrix(Xs.shape, dtype=np.float64)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=Xs.shape)


def itemknn_fit_reference_shaped(X: sp.csr_matrix, K: int, rows=None) -> sp.csr_matrix:
    """nearest_neighbour.py:69-84 + :204-224: sklearn's cosine_similarity(X.T, dense_output=False), the diagonal
    zeroed, then the per-row argpartition loop.  `rows` restricts the left operand to a sample of item rows (the
    bench's bounded sample); the result has one row per sampled item."""
    from sklearn.metrics.pairwise import cosine_similarity

    Xt = sp.csr_matrix(binarize(X).T, dtype=np.float64)
    left = Xt if rows is None else Xt[np.asarray(rows)]
    S = sp.csr_matrix(cosine_similarity(left, Xt, dense_output=False))
    row_ids = np.arange(Xt.shape[0]) if rows is None else np.asarray(rows)
    own = np.repeat(row_ids, np.diff(S.indptr))
    S.data[S.indices == own] = 0.0  # setdiag(0) (kept as an explicit zero, as scipy does)
    return topk_per_row_loop(S, K)


def predict_topk_reference_shaped(X: sp.csr_matrix, S: sp.csr_matrix, Kp: int) -> sp.csr_matrix:
    """base.py:248 (X @ similarity_matrix_) followed by the metrics' per-row top-K loop (metrics/base.py:189 ->
    util.py:50-77)."""
    return topk_per_row_loop(sp.csr_matrix(binarize(X).astype(np.float64) @ S), Kp)


# --------------------------------------------------------------------------------------
# EASE
# --------------------------------------------------------------------------------------
def ease_fit(X: sp.csr_matrix, l2: float, alpha: float = 0.0) -> np.ndarray:
    """ease.py:78-90 in float64: B_ij = -P_ij / P_jj (i != j), B_ii = 0, optional column weights c_j^-alpha.

    The two n^3 products with diagonal matrices of the reference (ease.py:85, :90) are column scalings.
    Items never interacted with (c_j = 0) give 0/inf = NaN columns in the reference when alpha != 0;
    here their (all-zero) columns stay zero - documented deviation.
    """
    Xb = binarize(X)
    G = (Xb.T @ Xb).toarray().astype(np.float64)
    n = G.shape[0]
    # ease.py:82 builds l2 * np.identity(n, dtype=float32): lambda is ROUNDED TO float32 before the
    # int32 + float32 -> float64 promotion (0.03 becomes 0.029999999329447746).
    lam = float(np.float32(l2))
    P = np.linalg.inv(G + lam * np.identity(n))
    d = np.diag(P).copy()
    B = -P / d[None, :]
    B[np.diag_indices(n)] = 0.0
    if alpha != 0:
        c = np.diag(G)
        w = np.zeros(n)
        nz = c > 0
        w[nz] = 1.0 / c[nz] ** alpha
        B = B * w[None, :]
    return B


def ease_xy_fit(X: sp.csr_matrix, Y: sp.csr_matrix, l2: float, alpha: float = 0.0) -> np.ndarray:
    """experimental/ease_xy.py:76-92 in float64: B = P (X^T y) - P diag(diag(P X^T y) / diag(P)), P = (X^T X + l2 I)^-1 (the
    identity is float64 there: l2 is not rounded), optional column weights c_j^-alpha (0 for items nobody interacted with,
    where the reference divides by zero)."""
    Xb, Yb = binarize(X), binarize(Y)
    n = Xb.shape[1]
    XTX = (Xb.T @ Xb).toarray().astype(np.float64)
    P = np.linalg.inv(XTX + float(l2) * np.identity(n))
    B_rr = P @ (Xb.T @ Yb).toarray().astype(np.float64)
    B = B_rr - P * (np.diag(B_rr) / np.diag(P))[None, :]
    if alpha != 0:
        c = np.diag(XTX)
        w = np.zeros(n)
        w[c > 0] = np.power(c[c > 0], -float(alpha))
        B = B * w[None, :]
    np.fill_diagonal(B, 0.0)  # (exactly zero up to rounding in the reference as well)
    return B


def ease_prune(B: np.ndarray, density: float) -> np.ndarray:
    """ease.py:97-109 with np.prod for the removed np.product: keep the ``int(density*n*n)`` largest |B_ij|
    among the stored (off-diagonal, non-zero) entries."""
    n = B.shape[0]
    S = sp.csr_matrix(B)
    k = min(int(density * n * n), S.nnz)
    out = B.copy()
    if k < S.nnz:
        a = np.abs(S.data)
        drop = np.argpartition(a, -k)[: S.nnz - k] if k > 0 else np.arange(S.nnz)
        r = np.repeat(np.arange(n), np.diff(S.indptr))
        out[r[drop], S.indices[drop]] = 0.0
    return out


def pipeline_scores(scores: sp.csr_matrix, history: sp.csr_matrix | None = None, item_keep: np.ndarray | None = None) -> sp.csr_matrix:
    """What Pipeline does to the predicted scores before the metrics (pipelines/pipeline.py:170-180): history removal
    ``X_pred - X_pred.multiply(data_in.binary_values)`` and the post-filters ``X_pred.multiply(mask)``
    (postprocessing/filters.py:58-101)."""
    out = sp.csr_matrix(scores, dtype=np.float64)
    if history is not None:
        out = (out - out.multiply(binarize(sp.csr_matrix(history)))).tocsr()
    if item_keep is not None:
        out = out.multiply(np.asarray(item_keep, dtype=bool)).tocsr()
    out.eliminate_zeros()
    return out


def metrics_topk(y_true: sp.csr_matrix, scores: sp.csr_matrix, K: int) -> dict:
    """NDCGK / RecallK / CalibratedRecallK / CoverageK of the reference (metrics/dcg.py:98-137, recall.py:34-44, 75-85,
    coverage.py:30-36, user elimination metrics/base.py:97-118) from a FULL score matrix, per-user loop, with the
    deterministic rank order (score desc, item id asc) inside the top-K."""
    y_true = binarize(sp.csr_matrix(y_true))
    scores = sp.csr_matrix(scores)
    discount = 1.0 / np.log2(np.arange(2, K + 2))
    idcg = np.concatenate([[1.0], np.cumsum(discount)])
    ndcg, rec, crec = [], [], []
    covered = set()
    for u in np.nonzero(np.diff(y_true.indptr) > 0)[0]:
        cols = scores.indices[scores.indptr[u]:scores.indptr[u + 1]]
        vals = scores.data[scores.indptr[u]:scores.indptr[u + 1]]
        nz = vals != 0
        cols, vals = cols[nz], vals[nz]
        top = cols[np.lexsort((cols, -vals))[:K]]
        covered.update(top.tolist())
        true = y_true.indices[y_true.indptr[u]:y_true.indptr[u + 1]]
        hit = np.isin(top, true).astype(np.float64)
        ndcg.append(float((hit * discount[: len(top)]).sum()) / idcg[min(K, len(true))])
        rec.append(hit.sum() / len(true))
        crec.append(hit.sum() / min(K, len(true)))
    return {"ndcg": float(np.mean(ndcg)), "recall": float(np.mean(rec)), "calibrated_recall": float(np.mean(crec)),
            "coverage": len(covered) / scores.shape[1]}


# --------------------------------------------------------------------------------------
# time-aware ItemKNN (recpack/algorithms/time_aware_item_knn/)
# --------------------------------------------------------------------------------------
def tars_decay(kind: str, decay: float, x: np.ndarray, max_distance: float) -> np.ndarray:
    """decay_functions.py: ExponentialDecay :24-60, ConvexDecay :63-99, ConcaveDecay :102-142, LogDecay :145-185,
    LinearDecay :188-228, InverseDecay :231-252; ``none`` is NoDecay (:255-259), ``liu`` is LiuDecay (liu_2012.py:16-48)."""
    x = np.asarray(x, dtype=np.float64)
    if kind == "none":
        return np.exp(-0 * x)
    if kind == "exponential":
        return np.exp(-decay * x)
    if kind == "convex":
        return np.power(decay, x)
    if kind == "concave":
        return 1 - np.power(decay, 1 - (x / max_distance))
    if kind == "log":
        return np.log(((decay - 1) * (1 - x / max_distance)) + 1) / np.log(decay)
    if kind == "linear":
        r = 1 - (x / max_distance) * decay
        r[r < 0] = 0
        return r
    if kind == "inverse":
        r = x.astype(float).copy()
        r[r > 0] = 1 / r[r > 0]
        r[r == 0] = 1
        return r
    if kind == "liu":
        return (np.log(((decay - 1) * x) + 1) / np.log(decay)) + 1
    raise ValueError(kind)


def tars_weights(T: sp.csr_matrix, decay: float, decay_function: str, decay_interval: int = 1) -> sp.csr_matrix:
    """TARSItemKNN._add_decay_to_interaction_matrix (time_aware_item_knn/base.py:181-195) on the last-timestamp matrix T
    (stored entries, timestamp 0 included): now = max + 1, age = (now - t) / interval, weight = decay(age)."""
    T = sp.csr_matrix(T)
    now = T.data.max() + 1
    ages = (now - T.data) / decay_interval
    kind = "none" if decay == 0 else decay_function
    return sp.csr_matrix((tars_decay(kind, decay, ages, ages.max()), T.indices.copy(), T.indptr.copy()), shape=T.shape)


def tars_similarity(W: sp.csr_matrix, similarity: str) -> sp.csr_matrix:
    """The full item-item similarity of a weighted matrix, diagonal removed, zero sums not stored (scipy's product drops
    them): compute_cosine_similarity (nearest_neighbour.py:69-84: unit-length item vectors, then the product),
    compute_conditional_probability (:22-66, pop_discount 0: to_binary(X).T @ X, rows over the item frequencies),
    compute_pearson_similarity (:87-112: positive entries centred by the item's mean of positive entries, then cosine)."""
    W = sp.csr_matrix(W, dtype=np.float64)
    n = W.shape[1]
    if similarity == "pearson":
        if (W == 1).sum() == W.nnz:
            raise ValueError("Pearson similarity can not be computed on a binary matrix.")
        cnt = np.asarray((W > 0).sum(axis=0)).ravel()
        avg = np.asarray(W.sum(axis=0)).ravel().astype(float)
        avg[cnt > 0] = avg[cnt > 0] / cnt[cnt > 0]
        W = sp.csr_matrix(W - (W > 0).multiply(avg[None, :]))
    if similarity in ("cosine", "pearson"):
        Wt = sp.csr_matrix(W.T)
        norms = np.sqrt(np.asarray(Wt.multiply(Wt).sum(axis=1)).ravel())
        norms[norms == 0] = 1.0
        Wn = Wt.copy()  # rows of X.T to unit length: a division per entry, as sklearn's normalize does
        Wn.data = Wt.data / np.repeat(norms, np.diff(Wt.indptr))
        S = sp.csr_matrix(Wn @ Wn.T)
    elif similarity == "conditional_probability":
        Wb = sp.csr_matrix(W.astype(bool).astype(np.float64))
        co = Wb.T @ W
        c = np.asarray(Wb.sum(axis=0)).ravel()
        inv = np.zeros(n)
        inv[c != 0] = 1.0 / c[c != 0]
        S = sp.csr_matrix(sp.diags(inv) @ co)
    else:
        raise ValueError(similarity)
    S.setdiag(0)
    S.eliminate_zeros()
    S.sort_indices()
    return S


def tars_fit(W: sp.csr_matrix, K: int, similarity: str):
    """TARSItemKNN._fit (base.py:166-179): similarity of the weighted matrix, then get_top_K_values (recpack/util.py:80-96)
    over the STORED entries - the non-zero similarities plus the explicit 0 that setdiag(0) leaves on the diagonal, which
    outranks negative entries (pearson) - and the result without its zeros.  Rank order: value desc, column asc."""
    S = tars_similarity(W, similarity)
    n = S.shape[0]
    gi = np.concatenate([np.repeat(np.arange(n, dtype=np.int64), np.diff(S.indptr)), np.arange(n, dtype=np.int64)])
    gj = np.concatenate([S.indices.astype(np.int64), np.arange(n, dtype=np.int64)])
    gv = np.concatenate([S.data.astype(np.float64), np.zeros(n)])
    order = np.lexsort((gj, gi))
    gi, gj, gv = gi[order], gj[order], gv[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(gi, minlength=n), out=indptr[1:])
    sel, _ = _segment_topk(indptr, gj, gv, K)
    sel = sel[gv[sel] != 0]
    nsel = np.bincount(gi[sel], minlength=n)
    idx = np.full((n, K), -1, dtype=np.int32)
    out = np.zeros((n, K), dtype=np.float64)
    r = gi[sel]
    pos = np.arange(sel.shape[0]) - np.repeat(np.concatenate([[0], np.cumsum(nsel)[:-1]]), nsel)
    idx[r, pos] = gj[sel]
    out[r, pos] = gv[sel]
    return {"idx": idx, "val": out, "nnz": nsel.astype(np.int32), "full": S}


def tars_predict(Wp: sp.csr_matrix, S: sp.csr_matrix) -> sp.csr_matrix:
    """TARSItemKNN._predict (base.py:139-150): the decayed history times the similarity matrix; zero sums are not stored."""
    P = sp.csr_matrix(sp.csr_matrix(Wp, dtype=np.float64) @ sp.csr_matrix(S, dtype=np.float64))
    P.eliminate_zeros()
    P.sort_indices()
    return P


def tars_cooc_similarity(T: sp.csr_matrix, similarity: str, decay_function: str, decay: float, decay_interval: int = 1) -> sp.csr_matrix:
    """TARSItemKNNCoocDistance._fit before the top-K (time_aware_item_knn/base.py:260-294), all rows at once: with
    t = last timestamp / interval, S_ij = sum over the users with t_ui > 0 and t_uj > 0, t_uj != t_ui, of
    decay(|t_uj - t_ui|) (equal timestamps: the difference is not stored, :279), divided by the number of users with
    t_ui > 0 for "conditional_probability" (:288-289).  The bounded decays are normalised by max - min over all stored
    timestamps (:268)."""
    T = sp.csr_matrix(T, dtype=np.float64) / decay_interval
    T = sp.csr_matrix(T)
    n = T.shape[1]
    max_dist = T.data.max() - T.data.min()
    keep = T.data > 0
    rows = np.repeat(np.arange(T.shape[0]), np.diff(T.indptr))[keep]
    cols, t = T.indices[keep].astype(np.int64), T.data[keep]
    d = np.bincount(rows, minlength=T.shape[0])
    start = np.concatenate([[0], np.cumsum(d)[:-1]])
    # all ordered pairs (a, b) of entries of the same user
    a = np.repeat(np.arange(rows.shape[0]), d[rows])
    b = np.arange(a.shape[0]) - np.repeat(np.cumsum(d[rows]) - d[rows], d[rows]) + start[rows[a]]
    dist = np.abs(t[b] - t[a])
    nz = dist != 0
    kind = "none" if decay == 0 else decay_function
    w = tars_decay(kind, decay, dist[nz], max_dist)
    S = sp.coo_matrix((w, (cols[a][nz], cols[b][nz])), shape=(n, n)).tocsr()
    if similarity == "conditional_probability":
        c = np.bincount(cols, minlength=n).astype(np.float64)
        inv = np.zeros(n)
        inv[c > 0] = 1 / c[c > 0]
        S = sp.csr_matrix(sp.diags(inv) @ S)
    elif similarity != "cooc":
        raise ValueError(similarity)
    S.eliminate_zeros()
    S.sort_indices()
    return S


def tars_cooc_fit(T: sp.csr_matrix, K: int, similarity: str, decay_function: str, decay: float, decay_interval: int = 1):
    """... then get_top_K_values per row (base.py:292): the K largest stored (positive) sums; value desc, column asc."""
    S = tars_cooc_similarity(T, similarity, decay_function, decay, decay_interval)
    idx, val, nnz = topk_rows(S, K)
    return {"idx": idx, "val": val, "nnz": nnz, "full": S}


def tars_hermann_similarity(T: sp.csr_matrix, decay_interval: int = 1) -> sp.csr_matrix:
    """TARSItemKNNHermann._fit before the top-K (time_aware_item_knn/hermann_2010.py:64-103), all rows at once: with
    t = last timestamp / interval (> 0 only) and now = max t + 1 / interval (:70),
    S_ij = (1 / n_ij) * sum over the users holding both, t_uj != t_ui, of 1 / (|t_uj - t_ui| + now - min(t_ui, t_uj)),
    n_ij = the number of users holding both (:97-98, ``invert`` then ``multiply``)."""
    T = sp.csr_matrix(sp.csr_matrix(T, dtype=np.float64) / decay_interval)
    n = T.shape[1]
    now = T.max() + 1 / decay_interval
    keep = T.data > 0
    rows = np.repeat(np.arange(T.shape[0]), np.diff(T.indptr))[keep]
    cols, t = T.indices[keep].astype(np.int64), T.data[keep]
    d = np.bincount(rows, minlength=T.shape[0])
    start = np.concatenate([[0], np.cumsum(d)[:-1]])
    a = np.repeat(np.arange(rows.shape[0]), d[rows])
    b = np.arange(a.shape[0]) - np.repeat(np.cumsum(d[rows]) - d[rows], d[rows]) + start[rows[a]]
    dist = np.abs(t[b] - t[a])
    N = sp.coo_matrix((np.ones(a.shape[0]), (cols[a], cols[b])), shape=(n, n)).tocsr()
    nz = dist != 0
    w = 1 / (dist[nz] + (now - np.minimum(t[a][nz], t[b][nz])))
    S = sp.coo_matrix((w, (cols[a][nz], cols[b][nz])), shape=(n, n)).tocsr()
    Ninv = N.copy()
    Ninv.data = 1 / N.data
    S = sp.csr_matrix(S.multiply(Ninv))
    S.setdiag(0)
    S.eliminate_zeros()
    S.sort_indices()
    return S


def tars_hermann_fit(T: sp.csr_matrix, K: int, decay_interval: int = 1):
    S = tars_hermann_similarity(T, decay_interval)
    idx, val, nnz = topk_rows(S, K)
    return {"idx": idx, "val": val, "nnz": nnz, "full": S}
