"""Oracle-backed stand-ins for the entry points of ``recpack_amd._lib`` that the drop-in classes call (TEST INFRASTRUCTURE: used
only by tests/pipeline_dropin_check.py, in the build container, where there is no GPU).  They let the reference's REAL
``Pipeline.run()`` drive ``recpack_amd.ItemKNN`` / ``EASE`` end to end - registry lookup, constructor, ``fit(InteractionMatrix)``,
``predict``, history removal, post-filters, metrics keyed by ``identifier`` - with the numbers coming from oracle/oracle_np.py instead
of the HIP library.  The product path has no such fallback: without the library every call raises."""
import numpy as np
import scipy.sparse as sp

from oracle import oracle_np as onp

RPHIP_MODEL_TOPK, RPHIP_MODEL_DENSE = 0, 1
SIM_COSINE, SIM_CONDITIONAL_PROBABILITY = 0, 1
CALLS = []  # (entry point, ...) in call order: the check asserts that the drop-in classes really went through here


class Context:
    def __init__(self, device=0, stream=None):
        self.device = device

    def sync(self):
        pass


class DeviceCSR:
    def __init__(self, X):
        self.X = X
        self.shape = X.shape
        self.nnz = X.nnz

    @classmethod
    def from_host(cls, ctx, indptr, indices, shape):
        CALLS.append(("csr_from_host", tuple(shape)))
        return cls(sp.csr_matrix((np.ones(len(indices), dtype=np.float64), np.asarray(indices), np.asarray(indptr)), shape=shape))

    def close(self):
        pass


class Model:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    @classmethod
    def from_topk_host(cls, ctx, idx, val, nnz):
        return cls(RPHIP_MODEL_TOPK, idx=np.asarray(idx), val=np.asarray(val), nnz=np.asarray(nnz))

    @classmethod
    def from_dense_host(cls, ctx, B):
        return cls(RPHIP_MODEL_DENSE, B=np.asarray(B, dtype=np.float32))

    def topk_download(self):
        return self.idx, self.val, self.nnz

    def dense_download(self):
        return self.B

    def empty_rows(self):
        return int(np.count_nonzero(~(self.B != 0).any(axis=1)))

    def dense_prune(self, keep):
        self.B = onp.ease_prune(self.B.astype(np.float64), keep / float(self.B.shape[0] ** 2)).astype(np.float32)

    def similarity(self, n_items):
        if self.kind == RPHIP_MODEL_TOPK:  # (predict uses the float32 similarities, as the device model does)
            return onp.topk_to_csr(self.idx, self.val.astype(np.float32).astype(np.float64), self.nnz, n_items)
        return sp.csr_matrix(self.B.astype(np.float64))

    def close(self):
        pass


def itemknn_fit(ctx, X, K, row_begin=0, row_end=None, similarity=SIM_COSINE, pop_discount=0.0, normalize_X=False):
    CALLS.append(("itemknn_fit", int(K), int(similarity), bool(normalize_X)))
    sim = "cosine" if similarity == SIM_COSINE else "conditional_probability"
    if normalize_X:
        ref = onp.itemknn_fit_normalized(X.X, K, similarity=sim, pop_discount=pop_discount)
    else:
        ref = onp.itemknn_fit(X.X, K, similarity=sim, pop_discount=pop_discount)
    key = np.where(ref["idx"] < 0, np.iinfo(np.int32).max, ref["idx"])  # rank order -> the device's column order
    perm = np.argsort(key, axis=1, kind="stable")
    return Model(RPHIP_MODEL_TOPK, idx=np.take_along_axis(ref["idx"], perm, axis=1), val=np.take_along_axis(ref["val"], perm, axis=1), nnz=ref["nnz"])


def ease_fit(ctx, X, l2, alpha):
    CALLS.append(("ease_fit", float(l2), float(alpha)))
    return Model(RPHIP_MODEL_DENSE, B=np.nan_to_num(onp.ease_fit(X.X, l2, alpha)).astype(np.float32))


def predict_full(ctx, model, X, user_begin=0, user_end=None):
    CALLS.append(("predict_full", X.shape))
    P = onp.predict_full(X.X, model.similarity(X.shape[1]))
    return P.indptr, P.indices, P.data


def predict_topk(ctx, model, X, Kp, exclude_history=False, item_keep=None, user_begin=0, user_end=None):
    CALLS.append(("predict_topk", int(Kp), bool(exclude_history)))
    S = onp.pipeline_scores(onp.predict_full(X.X, model.similarity(X.shape[1])), history=X.X if exclude_history else None, item_keep=item_keep)
    idx, val, nnz = onp.topk_rows(S, Kp)
    order = np.argsort(np.where(idx < 0, np.iinfo(np.int32).max, idx), axis=1, kind="stable")
    return np.take_along_axis(idx, order, axis=1), np.take_along_axis(val, order, axis=1).astype(np.float32), nnz


def install(lib_module):
    """Replace the entry points of recpack_amd._lib with the ones above."""
    for name in ("Context", "DeviceCSR", "Model", "itemknn_fit", "ease_fit", "predict_full", "predict_topk"):
        setattr(lib_module, name, globals()[name])
