"""Time-aware ItemKNN family (recpack/algorithms/time_aware_item_knn/) on MI355X.

Host-side mirror of ``TARSItemKNN`` (time_aware_item_knn/base.py:34-201) and the variants that only change how an
interaction is weighted: ``TARSItemKNNDing`` (ding_2005.py), ``TARSItemKNNLiu`` (liu_2010.py), ``TARSItemKNNVaz``
(vaz_2013.py), ``TARSItemKNNLiu2012`` (liu_2012.py), ``TARSItemKNNLee`` (lee_2007.py).  The host computes one float64
weight per stored interaction (an O(nnz) numpy pass, as the reference does); the Gram of the weighted matrix, the
similarity, the per-row top-K and the weighted predict run on the GPU (``rphip_csr_set_values``,
``rphip_itemknn_fit_values``, the predict entry points on a valued matrix).  No CPU fallback.

Input: recpack's ``InteractionMatrix`` (anything with ``last_timestamps_matrix``, ``timestamps``, ``shape``) or the
stand-in :class:`TimedInteractions` built from ``(user, item, timestamp)`` arrays.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from recpack_amd import _lib
from recpack_amd.algorithms import TopKItemSimilarityMatrixAlgorithm, get_context, topk_arrays_to_csr


class TimedInteractions:
    """What the time-aware algorithms read from recpack's ``InteractionMatrix`` (matrix/interaction_matrix.py:202-275):
    ``shape``, ``has_timestamps``, ``timestamps`` (Series indexed by (uid, iid)), ``last_timestamps_matrix``,
    ``binary_values``."""

    USER_IX, ITEM_IX, TIMESTAMP_IX = "uid", "iid", "ts"

    def __init__(self, uid, iid, ts, shape=None):
        self._uid = np.asarray(uid, dtype=np.int64)
        self._iid = np.asarray(iid, dtype=np.int64)
        self._ts = np.asarray(ts)
        if not (self._uid.shape == self._iid.shape == self._ts.shape) or self._uid.ndim != 1:
            raise ValueError("uid, iid and ts must be 1-d arrays of the same length")
        if shape is None:
            shape = (int(self._uid.max()) + 1 if self._uid.size else 0, int(self._iid.max()) + 1 if self._iid.size else 0)
        self.shape = (int(shape[0]), int(shape[1]))
        self.has_timestamps = True

    @property
    def timestamps(self):
        import pandas as pd

        index = pd.MultiIndex.from_arrays([self._uid, self._iid], names=[self.USER_IX, self.ITEM_IX])
        return pd.Series(self._ts, index=index, name=self.TIMESTAMP_IX)

    @property
    def last_timestamps_matrix(self) -> sp.csr_matrix:
        # the last event of every (user, item) pair; a timestamp of 0 stays a stored entry
        key = self._uid * self.shape[1] + self._iid
        order = np.lexsort((self._ts, key))
        key_s = key[order]
        last = np.ones(key_s.shape[0], dtype=bool)
        last[:-1] = key_s[1:] != key_s[:-1]
        sel = order[last]
        return sp.csr_matrix((self._ts[sel], (self._uid[sel], self._iid[sel])), shape=self.shape)

    @property
    def binary_values(self) -> sp.csr_matrix:
        M = self.last_timestamps_matrix
        M.data = np.ones_like(M.data, dtype=np.int64)
        return M

    @property
    def values(self) -> sp.csr_matrix:
        return self.binary_values


# ---- decay functions (time_aware_item_knn/decay_functions.py) -----------------------------------------------------------
def _check_range(ok: bool, decay, text: str):
    if not ok:
        raise ValueError(f"Decay parameter = {decay} is not in the supported range: {text}.")


_DECAY_RANGES = {
    "exponential": (lambda d: 0 <= d <= 1, "[0, 1]"),
    "convex": (lambda d: 0 < d <= 1, "]0, 1]"),
    "concave": (lambda d: 0 < d <= 1, "]0, 1]"),
    "log": (lambda d: 1 < d, "]1, inf["),
    "linear": (lambda d: 0 <= d, "[0, +inf["),
}


def validate_decay(kind: str, decay: float):
    if kind in _DECAY_RANGES:
        ok, text = _DECAY_RANGES[kind]
        _check_range(ok(decay), decay, text)


def apply_decay(kind: str, decay: float, x: np.ndarray, max_distance: float) -> np.ndarray:
    """decay_functions.py:24-252: the weight of an event of age ``x`` (``max_distance`` normalises the bounded ones)."""
    x = np.asarray(x, dtype=np.float64)
    if kind in ("concave", "log", "linear") and (x > max_distance).any():
        raise ValueError("At least one of the distances is bigger than the specified max_distance.")
    if kind == "exponential":
        return np.exp(-decay * x)
    if kind == "convex":
        return np.power(decay, x)
    if kind == "concave":
        return 1 - np.power(decay, 1 - (x / max_distance))
    if kind == "log":
        return np.log(((decay - 1) * (1 - x / max_distance)) + 1) / np.log(decay)
    if kind == "linear":
        w = 1 - (x / max_distance) * decay
        w[w < 0] = 0
        return w
    if kind == "inverse":
        w = x.copy()
        pos = w > 0
        w[pos] = 1 / w[pos]
        w[x == 0] = 1
        return w
    raise ValueError(f"Decay function {kind} is not supported.")


def _canonical_valued(W: sp.csr_matrix) -> sp.csr_matrix:
    """Entries of weight 0 never contribute (scipy's products drop them): remove them; sorted, float64."""
    W = sp.csr_matrix(W, dtype=np.float64, copy=True)
    W.sum_duplicates()
    W.eliminate_zeros()
    W.sort_indices()
    W.data = W.data + 0.0  # -0.0 -> +0.0 cannot occur after eliminate_zeros; keeps a private array
    return W


class TARSItemKNN(TopKItemSimilarityMatrixAlgorithm):
    """time_aware_item_knn/base.py:34-201: ItemKNN on interactions weighted by a decay of their age, at fit and at predict.

    Same constructor arguments, validation and error texts as the reference.
    """

    SUPPORTED_SIMILARITIES = ["cosine", "conditional_probability", "pearson"]
    DECAY_FUNCTIONS = ["exponential", "log", "linear", "concave", "convex", "inverse"]

    def __init__(self, K: int = 200, fit_decay: float = 1 / (24 * 3600), predict_decay: float = 1 / (24 * 3600),
                 decay_interval: int = 1, similarity: str = "cosine", decay_function: str = "exponential"):
        super().__init__(K=K)
        if decay_interval <= 0 or type(decay_interval) == float:
            raise ValueError("Parameter decay_interval needs to be a positive integer.")
        self.decay_interval = decay_interval
        if similarity not in self.SUPPORTED_SIMILARITIES:
            raise ValueError(f"Similarity {similarity} is not supported.")
        self.similarity = similarity
        if decay_function not in self.DECAY_FUNCTIONS:
            raise ValueError(f"Decay function {decay_function} is not supported.")
        self.decay_function = decay_function
        if decay_function in ("exponential", "log", "linear", "concave", "convex"):
            if fit_decay != 0:
                validate_decay(decay_function, fit_decay)
            if predict_decay != 0:
                validate_decay(decay_function, predict_decay)
        self.fit_decay = fit_decay
        self.predict_decay = predict_decay

    # ---- input checks (base.py:152-164 via Algorithm._assert_*) ----
    def _check_timed(self, X):
        if isinstance(X, (sp.spmatrix, np.ndarray)) or not hasattr(X, "last_timestamps_matrix"):
            raise TypeError(f"{self.name} requires Interaction Matrix as input. Got {type(X)}.")
        if not getattr(X, "has_timestamps", True):
            raise ValueError(f"{self.name} requires timestamp information in the InteractionMatrix.")
        return X

    def _transform_fit_input(self, X):
        return self._check_timed(X)

    def _transform_predict_input(self, X):
        return self._check_timed(X)

    # ---- weights (base.py:181-201) ----
    def _decay(self, decay: float, ages: np.ndarray) -> np.ndarray:
        if decay == 0:
            return np.exp(-0.0 * np.asarray(ages, dtype=np.float64))  # NoDecay: a binary matrix
        return apply_decay(self.decay_function, decay, ages, float(np.max(ages)))

    def _add_decay_to_interaction_matrix(self, X, decay: float) -> sp.csr_matrix:
        T = sp.csr_matrix(X.last_timestamps_matrix)
        if T.nnz == 0:
            return sp.csr_matrix(T.shape, dtype=np.float64)
        now = T.data.max() + 1  # the newest event has age 1: no weight vanishes
        ages = (now - T.data) / self.decay_interval
        return sp.csr_matrix((self._decay(decay, ages), T.indices, T.indptr), shape=T.shape)

    def _add_decay_to_fit_matrix(self, X) -> sp.csr_matrix:
        return self._add_decay_to_interaction_matrix(X, self.fit_decay)

    def _add_decay_to_predict_matrix(self, X) -> sp.csr_matrix:
        return self._add_decay_to_interaction_matrix(X, self.predict_decay)

    # ---- fit (base.py:166-179) ----
    def _fit(self, X) -> None:
        self._fit_weighted(self._add_decay_to_fit_matrix(X))

    def _fit_weighted(self, W: sp.csr_matrix) -> None:
        n_items = W.shape[1]
        row_scale, left_binary = None, False
        if self.similarity == "pearson":  # nearest_neighbour.py:87-112
            if (W == 1).sum() == W.nnz:
                raise ValueError("Pearson similarity can not be computed on a binary matrix.")
            Wc = sp.csc_matrix(W, dtype=np.float64)
            cnt = np.diff(sp.csc_matrix(W > 0).indptr).astype(np.float64)
            tot = np.asarray(W.sum(axis=0)).ravel().astype(np.float64)
            avg = tot.copy()
            avg[cnt > 0] = tot[cnt > 0] / cnt[cnt > 0]
            cols = np.repeat(np.arange(n_items), np.diff(Wc.indptr))
            Wc.data = Wc.data - np.where(Wc.data > 0, avg[cols], 0.0)  # only the positive entries are centred
            W = sp.csr_matrix(Wc)
        W = _canonical_valued(W)
        if self.similarity in ("cosine", "pearson"):
            # sklearn's cosine_similarity: every item vector to unit length first, then the sparse product
            n2 = np.bincount(W.indices, weights=W.data * W.data, minlength=n_items)
            norm = np.sqrt(n2)
            norm[norm == 0] = 1.0
            W.data = W.data / norm[W.indices]
        else:  # conditional probability: to_binary(X).T @ X, rows divided by the item frequencies (:22-66, pop_discount 0)
            c = np.bincount(W.indices, minlength=n_items).astype(np.float64)
            row_scale = np.zeros(n_items)
            row_scale[c > 0] = 1.0 / c[c > 0]
            left_binary = True
        ctx = get_context()
        dW = _lib.DeviceCSR.from_host_valued(ctx, W)
        try:
            model = _lib.itemknn_fit_values(ctx, dW, int(self.K), left_binary=left_binary, row_scale=row_scale)
            idx, val, nnz = model.topk_download()
        finally:
            dW.close()
        self._model = model
        self._empty_rows_ = int(np.count_nonzero(nnz == 0))
        self.similarity_matrix_ = topk_arrays_to_csr(idx, val, nnz, n_items)

    # ---- predict (base.py:139-150) ----
    def _predict_weights(self, X) -> sp.csr_matrix:
        return self._add_decay_to_predict_matrix(X)

    def _valued_upload(self, X):
        W = _canonical_valued(self._predict_weights(X))
        ctx = get_context()
        return ctx, W, _lib.DeviceCSR.from_host_valued(ctx, W)

    def _predict(self, X) -> sp.csr_matrix:
        ctx, W, dW = self._valued_upload(X)
        try:
            indptr, indices, data = _lib.predict_full(ctx, self._model, dW)
        finally:
            dW.close()
        return sp.csr_matrix((data, indices, indptr), shape=W.shape)

    def predict(self, X) -> sp.csr_matrix:
        self._check_fit_complete()
        X = self._transform_predict_input(X)
        X_pred = self._predict(X)
        self._check_prediction(X_pred, sp.csr_matrix(X.last_timestamps_matrix))
        return X_pred

    def predict_topk(self, X, K: int, exclude_history: bool = False) -> sp.csr_matrix:
        """Top-K weighted scores per user, fused on the device (see ItemSimilarityMatrixAlgorithm.predict_topk)."""
        self._check_fit_complete()
        X = self._transform_predict_input(X)
        ctx, W, dW = self._valued_upload(X)
        try:
            idx, val, nnz = _lib.predict_topk(ctx, self._model, dW, min(int(K), max(1, W.shape[1])), exclude_history)
        finally:
            dW.close()
        return topk_arrays_to_csr(idx, val.astype(np.float64), nnz, W.shape[1])


class TARSItemKNNDing(TARSItemKNN):
    """ding_2005.py:12-47: plain ItemKNN similarities, exponentially decayed histories at predict."""

    SUPPORTED_SIMILARITIES = ["cosine", "conditional_probability"]

    def __init__(self, K: int = 200, predict_decay: float = 1 / (24 * 3600), similarity: str = "cosine"):
        super().__init__(K=K, fit_decay=0, predict_decay=predict_decay, similarity=similarity, decay_function="exponential")


class TARSItemKNNLiu(TARSItemKNN):
    """liu_2010.py:12-59: exponential decay at fit and predict, cosine."""

    def __init__(self, K: int = 200, fit_decay: float = 1 / (24 * 3600), predict_decay: float = 1 / (24 * 3600)):
        super().__init__(K=K, fit_decay=fit_decay, predict_decay=predict_decay, similarity="cosine", decay_function="exponential")


class TARSItemKNNVaz(TARSItemKNN):
    """vaz_2013.py:12-45: exponential decay at fit and predict, pearson."""

    def __init__(self, K: int = 200, fit_decay: float = 1 / (24 * 3600), predict_decay: float = 1 / (24 * 3600)):
        super().__init__(K, fit_decay, predict_decay, similarity="pearson", decay_function="exponential")


class TARSItemKNNLiu2012(TARSItemKNN):
    """liu_2012.py:16-120: weight log_a((a - 1) x + 1) + 1 of the event's position x in [0, 1] between the user's first
    event and last event, cosine, at fit and predict."""

    DECAY_FUNCTIONS = ["liu"]

    def __init__(self, K: int = 200, decay: float = 2.0):
        if not decay > 1:
            raise ValueError(f"Decay parameter = {decay} is not in the supported range: ]1, inf [")
        super().__init__(K=K, fit_decay=decay, predict_decay=decay, decay_function="liu", similarity="cosine")
        self.decay = decay

    def _add_decay_to_interaction_matrix(self, X, decay: float) -> sp.csr_matrix:
        T = sp.csr_matrix(X.last_timestamps_matrix)
        n_users = T.shape[0]
        ts = X.timestamps
        first = np.zeros(n_users)
        g = ts.groupby(level=0).min()
        first[np.asarray(g.index, dtype=np.int64)] = g.values
        last = np.asarray(T.max(axis=1).todense()).ravel().astype(np.float64)
        rows = np.repeat(np.arange(n_users), np.diff(T.indptr))
        # entries whose first- or last-event factor is 0 are dropped by the reference's sparse multiply (liu_2012.py:95-103):
        # the arrays it divides are then misaligned; such inputs (timestamps of 0) are refused here
        if np.any(first[rows] == 0) or np.any(last[rows] == 0):
            raise ValueError(f"{self.name}: timestamps of 0 are not supported (the reference's weights are undefined for them)")
        x = (T.data - first[rows]) / last[rows]
        w = (np.log(((decay - 1) * x) + 1) / np.log(decay)) + 1
        return sp.csr_matrix((w, T.indices, T.indptr), shape=T.shape)


class TARSItemKNNLee(TARSItemKNN):
    """lee_2007.py:18-126: a w x w table of weights indexed by the item's launch window and the event's time window;
    weights at fit only (cosine or pearson), binary histories at predict."""

    SUPPORTED_SIMILARITIES = ["cosine", "pearson"]
    W_MAP = {
        3: np.array([[0.7, 1.7, 2.7], [1.0, 2.0, 3.0], [1.3, 2.3, 3.3]]),
        5: np.array([[0.2, 1.2, 2.2, 3.2, 4.2], [0.4, 1.4, 2.4, 3.4, 4.4], [0.6, 1.6, 2.6, 3.6, 4.6], [0.8, 1.8, 2.8, 3.8, 4.8],
                     [1.0, 2.0, 3.0, 4.0, 5.0]]),
    }

    def __init__(self, K: int = 200, w: int = 5, similarity: str = "cosine"):
        super().__init__(K, similarity=similarity, fit_decay=0, predict_decay=0)
        if w not in self.W_MAP:
            raise ValueError(f"Weight matrix {w} is not supported. Only 3 and 5 are allowed.")
        self.w = w
        self.W = self.W_MAP[w]

    def _add_decay_to_fit_matrix(self, X) -> sp.csr_matrix:
        T = sp.csr_matrix(X.last_timestamps_matrix)
        ts = X.timestamps
        launch = np.zeros(T.shape[1])
        g = ts.groupby(level=1).min()
        launch[np.asarray(g.index, dtype=np.int64)] = g.values
        l_min = launch.min()
        l_splits = np.array([np.ceil(l_min + i * ((launch.max() - l_min) / self.w)) for i in range(1, self.w + 1)])
        t_min, t_max = ts.min(), ts.max()
        t_splits = np.array([np.ceil(t_min + i * ((t_max - t_min) / self.w)) for i in range(1, self.w + 1)])
        # the reference walks timestamps_mat.nonzero(): pairs whose last timestamp is 0 get no weight
        keep = T.data != 0
        rows = np.repeat(np.arange(T.shape[0]), np.diff(T.indptr))[keep]
        cols = T.indices[keep]
        l_ix = np.searchsorted(l_splits, launch[cols], side="left")  # first split >= value
        t_ix = np.searchsorted(t_splits, T.data[keep], side="left")
        return sp.csr_matrix((self.W[l_ix, t_ix], (rows, cols)), shape=T.shape)


class TARSItemKNNCoocDistance(TARSItemKNN):
    """time_aware_item_knn/base.py:204-294: the similarity of two items is the decayed time between a user's interactions
    with them, summed over the users that hold both (``"cooc"``) or averaged over the users of the row's item
    (``"conditional_probability"``).  Binary histories at predict."""

    SUPPORTED_SIMILARITIES = ["cooc", "conditional_probability"]

    def __init__(self, K: int = 200, fit_decay: float = 1 / (24 * 3600), decay_interval: int = 1, similarity: str = "cooc",
                 decay_function: str = "exponential"):
        super().__init__(K, fit_decay, 0, decay_interval, similarity, decay_function)

    def _fit(self, X) -> None:
        T = sp.csr_matrix(X.last_timestamps_matrix, dtype=np.float64) / self.decay_interval
        n_items = T.shape[1]
        if T.nnz == 0:
            max_dist = 0.0
        else:
            max_dist = float(T.data.max() - T.data.min())  # over every stored pair (base.py:268)
        T = sp.csr_matrix(T)
        T.data = np.where(T.data > 0, T.data, 0.0)  # only timestamps > 0 take part (the `> 0` masks of base.py:273-279)
        T = _canonical_valued(T)
        kind = "none" if self.fit_decay == 0 else self.decay_function
        row_scale = None
        if self.similarity == "conditional_probability":
            c = np.bincount(T.indices, minlength=n_items).astype(np.float64)
            row_scale = np.zeros(n_items)
            row_scale[c > 0] = 1 / c[c > 0]
        ctx = get_context()
        dT = _lib.DeviceCSR.from_host_valued(ctx, T)
        try:
            model = _lib.itemknn_fit_timedist(ctx, dT, int(self.K), kind, float(self.fit_decay), max_dist, row_scale=row_scale)
            idx, val, nnz = model.topk_download()
        finally:
            dT.close()
        self._model = model
        self._empty_rows_ = int(np.count_nonzero(nnz == 0))
        self.similarity_matrix_ = topk_arrays_to_csr(idx, val, nnz, n_items)


class TARSItemKNNXia(TARSItemKNNCoocDistance):
    """xia_2010.py:12-84: co-occurrence similarity with a concave, convex or linear decay of the time between the two
    interactions, time in days by default."""

    SUPPORTED_DECAY_FUNCTIONS = ["concave", "convex", "linear"]

    def __init__(self, K: int = 200, fit_decay: float = 0.5, decay_function: str = "convex", decay_interval: int = 24 * 3600):
        if decay_function not in self.SUPPORTED_DECAY_FUNCTIONS:
            raise ValueError(f"decay_function {decay_function} not supported")
        super().__init__(K=K, fit_decay=fit_decay, decay_interval=decay_interval, similarity="cooc", decay_function=decay_function)


class TARSItemKNNHermann(TopKItemSimilarityMatrixAlgorithm):
    """hermann_2010.py:20-103: the similarity of two items is the average, over the users that hold both, of
    1 / (time between the two interactions + age of the older one); binary histories at predict."""

    def __init__(self, K: int = 200, decay_interval: int = 1):
        super().__init__(K=K)
        self.decay_interval = decay_interval

    def _transform_fit_input(self, X):
        if isinstance(X, (sp.spmatrix, np.ndarray)) or not hasattr(X, "last_timestamps_matrix"):
            raise TypeError(f"{self.name} requires Interaction Matrix as input. Got {type(X)}.")
        if not getattr(X, "has_timestamps", True):
            raise ValueError(f"{self.name} requires timestamp information in the InteractionMatrix.")
        return X

    def _fit(self, X) -> None:
        T = sp.csr_matrix(sp.csr_matrix(X.last_timestamps_matrix, dtype=np.float64) / self.decay_interval)
        n_items = T.shape[1]
        now = (T.max() if T.nnz else 0.0) + 1 / self.decay_interval  # hermann_2010.py:70
        T.data = np.where(T.data > 0, T.data, 0.0)  # only timestamps > 0 take part (:76-79)
        T = _canonical_valued(T)
        ctx = get_context()
        dT = _lib.DeviceCSR.from_host_valued(ctx, T)
        try:
            model = _lib.itemknn_fit_hermann(ctx, dT, int(self.K), float(now))
            idx, val, nnz = model.topk_download()
        finally:
            dT.close()
        self._model = model
        self._empty_rows_ = int(np.count_nonzero(nnz == 0))
        self.similarity_matrix_ = topk_arrays_to_csr(idx, val, nnz, n_items)
