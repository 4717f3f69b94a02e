"""Drop-in ``ItemKNN`` and ``EASE`` with recpack's ``Algorithm.fit()/predict()`` surface, computed on MI355X.

Host-side mirror of the reference interface for the hot path (paths relative to /root/reference):

* ``Algorithm``            recpack/algorithms/base.py:33-217   (name, identifier, fit/predict wrappers, post-conditions)
* ``ItemSimilarityMatrixAlgorithm`` / ``TopKItemSimilarityMatrixAlgorithm``   base.py:220-304
* ``ItemKNN``              recpack/algorithms/nearest_neighbour.py:114-224
* ``EASE``                 recpack/algorithms/ease.py:19-109

Same constructor arguments (stored under the same attribute names so that sklearn's ``get_params`` and
recpack's ``identifier`` work), same exceptions and warning texts.  All arithmetic happens in
``librecpack_hip.so`` through :mod:`recpack_amd._lib`; there is no CPU fallback.

Extensions (not in the reference): ``predict_topk(X, K, exclude_history)`` returns only the K best
scores per user, fused on the device (what ``Pipeline`` + ``MetricTopK`` compute from the full matrix).

Verbatim interface blocks.  ``Algorithm.name`` / ``identifier`` / ``__str__`` / ``set_params`` / ``fit`` / ``predict`` and the
argument checks of ``ItemKNN.__init__`` (the pop_discount warning and the two ``ValueError`` texts) reproduce the reference's
class protocol character for character (base.py:42-72, 165-217; nearest_neighbour.py:168-202): attribute names, log and
warning texts and exception messages ARE the interface recpack's Pipeline, registry and tests rely on.  Everything else in
this module is original.  When recpack itself is importable, :class:`Algorithm` additionally derives from recpack's own
``Algorithm`` so that ``isinstance`` / ``issubclass`` checks in recpack (e.g. the algorithm registry) accept these classes.

K: every fit takes ANY ``K`` at any item count, as the reference does (round 6; beyond K x column tiles = 32 768 the rows' per-tile lists are
merged where they lie and the rows are fitted in chunks of bounded scratch), and so does the ``K`` of ``predict_topk`` - an extension, not part
of the reference: beyond 2048 it takes predict()'s full scores per chunk of users and cuts them on the device.
"""
from __future__ import annotations

import logging
import os
import time
import warnings
from typing import Optional

import numpy as np
import scipy.sparse as sp
from sklearn.base import BaseEstimator
from sklearn.utils.validation import check_is_fitted

from recpack_amd import _lib

logger = logging.getLogger("recpack")

RP_WTILE_MAX = 12288  # rp_itemknn_w.hip: items per column tile of the normalize_X fit (float64 sums)
RP_K_TILES_MAX = 32768  # rp_select.h (RPS_FLAG): from K x column tiles = this the fits take the large-K merge (rp_merge_big.h)
RP_K_MAX = 2048  # rp_common.h: largest Kp of the FUSED top-Kp kernels (beyond: full scores per user chunk + the large-K selection)
RP_TILE_MAX = 25600  # rp_common.h: items of one column tile - up to here ItemKNN takes any K (round 4)

try:  # recpack's own base class, when it is installed: registry / isinstance checks then accept the classes below
    from recpack.algorithms.base import Algorithm as _RecpackAlgorithm
except Exception:  # noqa: BLE001 - recpack is optional (it is absent on the GPU box)
    _RecpackAlgorithm = None
_AlgorithmBases = (_RecpackAlgorithm,) if _RecpackAlgorithm is not None else (BaseEstimator,)

_contexts: dict = {}


def get_context(device: Optional[int] = None) -> "_lib.Context":
    """Process-wide context per GPU (device defaults to LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = _lib.Context(device)
    return ctx


class UnsupportedTypeError(Exception):
    """recpack/matrix/util.py:64-80."""

    def __init__(self, X):
        super().__init__(
            "Recpack only supports matrix types InteractionMatrix, csr_matrix. Received {}.".format(type(X).__name__)
        )


def to_binary_csr(X) -> sp.csr_matrix:
    """``to_csr_matrix(X, binary=True)`` (recpack/matrix/util.py:27-48, recpack/util.py:99-109).

    Accepts a scipy ``csr_matrix`` or anything with recpack's ``InteractionMatrix.values`` property.
    Only the sparsity pattern matters downstream: explicit zeros are dropped, duplicates merged, indices sorted.
    """
    if isinstance(X, sp.csr_matrix):
        M = X
    else:
        M = getattr(X, "values", None)  # InteractionMatrix.values rebuilds the CSR on every access: read once
        if not isinstance(M, sp.csr_matrix):
            raise UnsupportedTypeError(X)
    if not M.has_canonical_format or (M.nnz and not np.all(M.data)):
        M = M.copy()
        M.sum_duplicates()
        M.eliminate_zeros()
        M.sort_indices()
    return M


class Algorithm(*_AlgorithmBases):
    """recpack/algorithms/base.py:33-217 (the public protocol below mirrors the reference verbatim - module docstring)."""

    def __init__(self):
        super().__init__()

    @property
    def name(self):
        return self.__class__.__name__

    @property
    def identifier(self):
        paramstring = ",".join((f"{k}={v}" for k, v in self.get_params().items()))
        return self.name + "(" + paramstring + ")"

    def __str__(self):
        return self.name

    def set_params(self, **params):
        super().set_params(**params)

    def _fit(self, X):
        raise NotImplementedError("Please implement _fit")

    def _predict(self, X):
        raise NotImplementedError("Please implement _predict")

    def _check_fit_complete(self):
        check_is_fitted(self)

    def _check_prediction(self, X_pred: sp.csr_matrix, X: sp.csr_matrix) -> None:
        # base.py:108-127, without materialising two Python sets of user ids
        has_hist = np.diff(X.indptr) > 0
        has_pred = np.diff(X_pred.indptr) > 0
        if X_pred.nnz and not np.all(X_pred.data):
            has_pred = np.asarray((X_pred != 0).sum(axis=1)).ravel() > 0
        missing = int(np.count_nonzero(has_hist & ~has_pred))
        if missing > 0:
            warnings.warn(f"{self.name} failed to recommend any items " f"for {missing} users")

    def _transform_fit_input(self, X):
        return to_binary_csr(X)

    def _transform_predict_input(self, X):
        return to_binary_csr(X)

    def fit(self, X) -> "Algorithm":
        start = time.time()
        X = self._transform_fit_input(X)
        self._fit(X)
        self._check_fit_complete()
        end = time.time()
        logger.info(f"Fitting {self.name} complete - Took {end - start :.3}s")
        return self

    def predict(self, X) -> sp.csr_matrix:
        self._check_fit_complete()
        X = self._transform_predict_input(X)
        X_pred = self._predict(X)
        self._check_prediction(X_pred, X)
        return X_pred


class ItemSimilarityMatrixAlgorithm(Algorithm):
    """base.py:220-279; the model lives on the GPU, ``similarity_matrix_`` is its host view.

    Pickling (``pickle`` / ``joblib.dump``, what a ``BaseEstimator`` user expects - recpack/algorithms/base.py:33): the device handle
    never travels.  ``__getstate__`` replaces it by host arrays (a neighbour-list model is rebuilt from ``similarity_matrix_``, which
    the estimator holds anyway; a dense model is downloaded once as float32), ``__setstate__`` keeps them, and the model is uploaded
    again on its first use - on whatever MI355X the unpickling process has; without one that first use raises, as every other call."""

    _model_dev = None      # _lib.Model on the device
    _model_host = None     # what an unpickled estimator re-uploads from on first use: ("topk", idx, val, nnz) or ("dense", B float32)
    _empty_rows_ = None    # rows of the similarity matrix without any stored value (set by _fit)

    @property
    def _model(self):
        if self._model_dev is None and self._model_host is not None:
            kind = self._model_host[0]
            ctx = get_context()
            if kind == "topk":
                self._model_dev = _lib.Model.from_topk_host(ctx, *self._model_host[1:])
            else:
                self._model_dev = _lib.Model.from_dense_host(ctx, self._model_host[1])
            self._model_host = None
        return self._model_dev

    @_model.setter
    def _model(self, m):
        self._model_dev = m
        self._model_host = None

    def _host_model(self):
        """The fitted model as host arrays (None when nothing is fitted)."""
        if self._model_host is not None:
            return self._model_host
        m = self._model_dev
        if m is None:
            return None
        if m.kind == _lib.RPHIP_MODEL_TOPK:
            return ("topk",) + tuple(m.topk_download())
        return ("dense", m.dense_download())

    def __getstate__(self):
        base = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__
        state = dict(base)
        state.pop("_model_dev", None)
        state.pop("_sim_cache", None)
        state["_model_host"] = self._host_model()
        return state

    def __setstate__(self, state):
        host = state.pop("_model_host", None)
        if hasattr(super(), "__setstate__"):
            super().__setstate__(state)
        else:
            self.__dict__.update(state)
        self._model_dev = None
        self._model_host = host

    def _upload(self, X: sp.csr_matrix):
        ctx = get_context()
        return ctx, _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)

    def _predict(self, X: sp.csr_matrix) -> sp.csr_matrix:
        ctx, dX = self._upload(X)
        try:
            indptr, indices, data = _lib.predict_full(ctx, self._model, dX)
        finally:
            dX.close()
        return sp.csr_matrix((data, indices, indptr), shape=X.shape)

    def predict_topk(self, X, K: int, exclude_history: bool = False, exclude_items=None, select_items=None) -> sp.csr_matrix:
        """Top-K scores per user (column-sorted CSR rows, float32 scores widened to float64).

        The steps recpack's Pipeline applies between predict and the metrics, fused into the kernel:
        ``exclude_history`` (pipelines/pipeline.py:174-175), ``exclude_items`` / ``select_items`` = the
        ExcludeItems / SelectItems post-filters (postprocessing/filters.py:58-101, same argument checks), and the
        per-user top-K of MetricTopK (metrics/base.py:189)."""
        self._check_fit_complete()
        X = self._transform_predict_input(X)
        n_items = X.shape[1]
        keep = None
        for items, select in ((exclude_items, False), (select_items, True)):
            if items is None:
                continue
            items = np.asarray(items)
            if len(items) == 0 or np.amax(items) > n_items:  # filters.py:69-70, 91-92
                raise ValueError(f"{X.shape} and {items.shape}")
            m = np.zeros(n_items, dtype=bool) if select else np.ones(n_items, dtype=bool)
            m[items] = select
            keep = m if keep is None else (keep & m)
        ctx, dX = self._upload(X)
        try:
            idx, val, nnz = _lib.predict_topk(ctx, self._model, dX, min(int(K), max(1, n_items)), exclude_history, item_keep=keep)
        finally:
            dX.close()
        mask = np.arange(idx.shape[1])[None, :] < nnz[:, None]
        indptr = np.zeros(X.shape[0] + 1, dtype=np.int64)
        np.cumsum(nnz, out=indptr[1:])
        return sp.csr_matrix((val[mask].astype(np.float64), idx[mask], indptr), shape=X.shape)

    def _check_fit_complete(self):
        super()._check_fit_complete()
        assert hasattr(self, "similarity_matrix_")
        if self._empty_rows_ is not None:
            missing = int(self._empty_rows_)
        else:  # a similarity matrix set from outside: the reference's own check (base.py:272-279)
            items_with_score = set(self.similarity_matrix_.nonzero()[0])
            missing = self.similarity_matrix_.shape[0] - len(items_with_score)
        if missing > 0:
            warnings.warn(f"{self.name} missing similar items for {missing} items.")


class TopKItemSimilarityMatrixAlgorithm(ItemSimilarityMatrixAlgorithm):
    """base.py:282-304."""

    def __init__(self, K):
        super().__init__()
        self.K = K


def topk_arrays_to_csr(idx: np.ndarray, val: np.ndarray, nnz: np.ndarray, n_cols: int) -> sp.csr_matrix:
    mask = np.arange(idx.shape[1])[None, :] < nnz[:, None]
    indptr = np.zeros(idx.shape[0] + 1, dtype=np.int64)
    np.cumsum(nnz, out=indptr[1:])
    return sp.csr_matrix((val[mask], idx[mask], indptr), shape=(idx.shape[0], n_cols))


class ItemKNN(TopKItemSimilarityMatrixAlgorithm):
    """Item K Nearest Neighbours (recpack/algorithms/nearest_neighbour.py:114-224) on MI355X.

    ``similarity="cosine"`` and ``"conditional_probability"`` (with ``pop_discount``) run on the GPU from the same
    integer Gram; ``normalize_X`` (rows l1-normalised first, nearest_neighbour.py:207-210) from a float64 weighted
    Gram.  ``normalize_sim`` is a host epilogue on the I x K result.
    """

    SUPPORTED_SIMILARITIES = ["cosine", "conditional_probability"]

    def __init__(self, K=200, similarity: str = "cosine", pop_discount: Optional[float] = None, normalize_X: bool = False,
                 normalize_sim: bool = False):
        super().__init__(K)
        if similarity not in self.SUPPORTED_SIMILARITIES:
            raise ValueError(f"similarity {similarity} not supported")
        self.similarity = similarity
        if self.similarity != "conditional_probability" and pop_discount:
            warnings.warn(
                "Argument pop_discount is incompatible with all similarity \
                functions except conditional probability. \
                This argument will be ignored, \
                popularity discounting won't be applied.",
                UserWarning,
            )
        if type(pop_discount) == float and (pop_discount < 0 or pop_discount > 1):
            raise ValueError("Invalid value for pop_discount. Value should be between 0 and 1.")
        self.pop_discount = pop_discount
        self.normalize_X = normalize_X
        self.normalize_sim = normalize_sim

    def _fit(self, X: sp.csr_matrix) -> None:
        # ANY K at any item count, as the reference (recpack/util.py:63-66, base.py:302-304) - binary fits, normalize_X and (time_aware.py) the
        # valued fits alike since round 6: beyond K x column tiles = 32 768 the per-tile lists are merged where they lie (rp_merge_big.h)
        # and the rows are fitted in chunks of bounded scratch
        ctx, dX = self._upload(X)
        try:
            if self.similarity == "cosine":
                model = _lib.itemknn_fit(ctx, dX, int(self.K), normalize_X=bool(self.normalize_X))
            else:  # nearest_neighbour.py:213-215: `if pop_discount:` - None and 0 mean no discount
                model = _lib.itemknn_fit(ctx, dX, int(self.K), similarity=_lib.SIM_CONDITIONAL_PROBABILITY,
                                         pop_discount=float(self.pop_discount or 0.0), normalize_X=bool(self.normalize_X))
            idx, val, nnz = model.topk_download()
        finally:
            dX.close()
        self._set_topk(ctx, model, idx, val, nnz, X.shape[1])

    def _set_topk(self, ctx, model, idx, val, nnz, n_items):
        if self.normalize_sim:
            # Normalizer(norm="l1") on the rows of the top-K matrix (nearest_neighbour.py:220-222)
            mask = np.arange(idx.shape[1])[None, :] < nnz[:, None]
            s = np.where(mask, np.abs(val), 0.0).sum(axis=1, keepdims=True)
            s[s == 0] = 1.0
            val = val / s
            model = _lib.Model.from_topk_host(ctx, idx, val, nnz)
        self._model = model
        self._empty_rows_ = int(np.count_nonzero(nnz == 0))
        self.similarity_matrix_ = topk_arrays_to_csr(idx, val, nnz, n_items)


class EASE(ItemSimilarityMatrixAlgorithm):
    """EASE (recpack/algorithms/ease.py:19-109) on MI355X.

    The dense item-item matrix B stays on the device in float32; ``similarity_matrix_`` is materialised on the
    host (float64 CSR, as the reference stores it, ease.py:92) only when it is read - for 50K items that is a
    20 GB download plus a CSR conversion, so predict() never touches it.
    ``density`` (ease.py:97-109) prunes on the device with the reference's rule.
    """

    _sim_cache = None  # host CSR view of B (never pickled: rebuilt from the model on demand)

    def __init__(self, l2=1e3, alpha=0, density=None):
        super().__init__()
        self.l2 = l2
        self.alpha = alpha
        self.density = density

    def _fit(self, X: sp.csr_matrix):
        ctx, dX = self._upload(X)
        try:
            self._model = _lib.ease_fit(ctx, dX, float(self.l2), float(self.alpha))
        finally:
            dX.close()
        self._sim_cache = None
        self.n_items_ = X.shape[1]
        if self.density:
            self._prune()
        self._empty_rows_ = self._model.empty_rows()

    def _prune(self):
        """ease.py:97-109: keep the ``int(density * n * n)`` largest |B_ij| (np.prod for the removed np.product), on the
        device: B never leaves HBM (an exact radix select over the magnitudes, then one zeroing pass)."""
        n = int(self.n_items_)
        self._model.dense_prune(int(self.density * n * n))
        self._sim_cache = None

    @property
    def similarity_matrix_(self) -> sp.csr_matrix:
        if getattr(self, "_model", None) is None:
            raise AttributeError("similarity_matrix_")
        if self._sim_cache is None:
            self._sim_cache = sp.csr_matrix(self._model.dense_download().astype(np.float64))
        return self._sim_cache

    @similarity_matrix_.setter
    def similarity_matrix_(self, value):  # an externally supplied matrix (reference tests do this before _prune)
        value = sp.csr_matrix(value)
        self._model = _lib.Model.from_dense_host(get_context(), value.toarray().astype(np.float32))
        self._sim_cache = value
        self.n_items_ = value.shape[0]
        self._empty_rows_ = self._model.empty_rows()

    def _check_fit_complete(self):
        check_is_fitted(self)
        if self._empty_rows_ is not None and self._empty_rows_ > 0:
            warnings.warn(f"{self.name} missing similar items for {self._empty_rows_} items.")


class EASEGram:
    """The part of an EASE fit that does not depend on ``l2`` or ``alpha``, computed once: the device copy of X and its Gram.

    recpack's pipelines grid-search ``l2`` (doc/source/guides.create_algorithm.rst:681, ``"l2": [10, 100, 1000]``) and the
    reference recomputes X^T X in every ``_fit`` (ease.py:81).  Usage::

        with EASEGram(X) as g:
            for l2 in (10, 100, 1000):
                algo = g.fit(EASE(l2=l2))      # == EASE(l2=l2).fit(X), bit for bit
    """

    def __init__(self, X):
        X = to_binary_csr(X)
        self.shape = X.shape
        self._ctx = get_context()
        self._dX = _lib.DeviceCSR.from_host(self._ctx, X.indptr, X.indices, X.shape)
        try:
            self._gram = _lib.ease_gram_create(self._ctx, self._dX)
        except Exception:
            self._dX.close()
            raise

    def fit(self, algo: "EASE") -> "EASE":
        """Fit ``algo`` (an EASE with its own l2 / alpha / density) from the cached Gram; returns ``algo``."""
        if self._gram is None:
            raise RuntimeError("EASEGram is closed")
        if type(algo) is not EASE:
            raise TypeError("EASEGram.fit takes an EASE instance")
        start = time.time()
        algo._model = _lib.ease_fit_gram(self._ctx, self._dX, self._gram, float(algo.l2), float(algo.alpha))
        algo._sim_cache = None
        algo.n_items_ = self.shape[1]
        if algo.density:
            algo._prune()
        algo._empty_rows_ = algo._model.empty_rows()
        algo._check_fit_complete()
        logger.info(f"fitting {algo.name} complete - Took {time.time() - start :.3}s")
        return algo

    def close(self):
        if self._gram is not None:
            self._gram.close()
            self._dX.close()
            self._gram = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class EASE_XY(EASE):
    """EASE_XY (recpack/algorithms/experimental/ease_xy.py:20-96): EASE trained to reconstruct a second matrix ``y`` from ``X``
    (``fit(X, y)``), B = P (X^T y) - P diag(diag(P X^T y) / diag(P)) with P = (X^T X + l2 I)^-1; predict as EASE."""

    def fit(self, X, y) -> "EASE_XY":
        start = time.time()
        X, y = to_binary_csr(X), to_binary_csr(y)
        if X.shape != y.shape:
            raise ValueError(f"X {X.shape} and y {y.shape} must have the same shape")
        ctx = get_context()
        dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
        dY = _lib.DeviceCSR.from_host(ctx, y.indptr, y.indices, y.shape)
        try:
            self._model = _lib.ease_xy_fit(ctx, dX, dY, float(self.l2), float(self.alpha))
        finally:
            dX.close()
            dY.close()
        self._sim_cache = None
        self.n_items_ = X.shape[1]
        if self.density:
            self._prune()
        self._empty_rows_ = self._model.empty_rows()
        self._check_fit_complete()
        logger.info(f"fitting {self.name} complete - Took {time.time() - start :.3}s")
        return self

