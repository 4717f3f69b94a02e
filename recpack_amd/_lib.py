"""ctypes binding of librecpack_hip.so (C ABI: include/recpack_hip.h).

The product path has NO CPU fallback: if the shared library is missing or no MI355X is visible,
calls raise :class:`RecpackHipError`.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C recpack_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RECPACK_HIP_LIB: another build of the same library - the diagnostic build `make diag`, an A/B variant of a tool run)
LIB_PATH = os.environ.get("RECPACK_HIP_LIB") or os.path.join(_HERE, "csrc", "librecpack_hip.so")

RPHIP_T_NAMES = [
    "csr_transpose", "fit_plan", "gram_topk", "topk_merge", "predict", "gram_dense", "cholesky", "inverse",
    "ease_scale", "h2d", "d2h", "gram_finalize", "gram_topk_wave", "sweep_update", "sweep_chain", "mirror", "predict_dense",
]
RPHIP_T_COUNT = 24
RPHIP_EINVAL, RPHIP_EHIP, RPHIP_ENOMEM, RPHIP_ENODEV, RPHIP_EUNSORTED, RPHIP_ELIMIT, RPHIP_ESTATE, RPHIP_ENUMERIC = -1, -2, -3, -4, -5, -6, -7, -8
RPHIP_MODEL_TOPK = 1
RPHIP_MODEL_DENSE = 2


class RecpackHipError(RuntimeError):
    """Raised for every failure of the HIP library (message from rphip_last_error)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"librecpack_hip error {code}: {message}")
        self.code = code


class Timings(C.Structure):
    _fields_ = [
        ("ms", C.c_double * RPHIP_T_COUNT),
        ("launches", C.c_int64 * RPHIP_T_COUNT),
        ("work", C.c_double * RPHIP_T_COUNT),
    ]

    def as_dict(self):
        return {
            n: {"ms": self.ms[i], "launches": self.launches[i], "work": self.work[i]}
            for i, n in enumerate(RPHIP_T_NAMES)
        }


_p = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/recpack_hip.h one to one
PROTOTYPES = {
    "rphip_version": (C.c_int, []),
    "rphip_last_error": (C.c_char_p, []),
    "rphip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rphip_ctx_create": (C.c_int, [C.c_int, _p, _pp]),
    "rphip_ctx_destroy": (C.c_int, [_p]),
    "rphip_ctx_sync": (C.c_int, [_p]),
    "rphip_get_timings": (C.c_int, [_p, C.POINTER(Timings)]),
    "rphip_reset_timings": (C.c_int, [_p]),
    "rphip_get_counters": (C.c_int, [_p, _p]),
    "rphip_device_info": (C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(_i64), C.c_char_p, C.c_int]),
    "rphip_csr_from_host": (C.c_int, [_p, _i64, _i64, _i64, _p, C.c_int, _p, _pp]),
    "rphip_csr_from_device": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _pp]),
    "rphip_csr_from_device_rows": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _pp]),
    "rphip_csr_from_device_shard": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _pp]),
    "rphip_csr_degrees_device": (C.c_int, [_p, _pp]),
    "rphip_csr_finish_degrees": (C.c_int, [_p, _p]),
    "rphip_csr_from_coo_host": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _pp]),
    "rphip_csr_from_coo_device": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _pp]),
    "rphip_csr_download": (C.c_int, [_p, _p, _p, _p]),
    "rphip_csr_destroy": (C.c_int, [_p]),
    "rphip_csr_shape": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "rphip_csr_item_degrees": (C.c_int, [_p, _p, _p]),
    "rphip_csr_num_products": (C.c_int, [_p, _p, C.POINTER(_i64)]),
    "rphip_itemknn_fit": (C.c_int, [_p, _p, _i64, _i64, _i32, _pp]),
    "rphip_itemknn_fit_ex": (C.c_int, [_p, _p, _i64, _i64, _i32, _i32, _f64, _i32, _pp]),
    "rphip_model_topk_download": (C.c_int, [_p, _p, _p, _p, _p]),
    "rphip_model_topk_device": (C.c_int, [_p, _pp, _pp, _pp, C.POINTER(_i64), C.POINTER(_i32)]),
    "rphip_model_from_topk_host": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _pp]),
    "rphip_model_from_topk_device": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _pp]),
    "rphip_model_destroy": (C.c_int, [_p]),
    "rphip_model_info": (C.c_int, [_p, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "rphip_predict_topk": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p]),
    "rphip_predict_topk_dev": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p]),
    "rphip_predict_topk_filtered": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _p]),
    "rphip_predict_topk_filtered_dev": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _p]),
    "rphip_predict_full_count": (C.c_int, [_p, _p, _p, _i64, _i64, _p, C.POINTER(_i64)]),
    "rphip_predict_full_fill": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p, _p]),
    "rphip_csr_set_values": (C.c_int, [_p, _p, _p]),
    "rphip_itemknn_fit_values": (C.c_int, [_p, _p, _i64, _i64, _i32, _i32, _p, _pp]),
    "rphip_itemknn_fit_timedist": (C.c_int, [_p, _p, _i64, _i64, _i32, _i32, _f64, _f64, _p, _pp]),
    "rphip_itemknn_fit_hermann": (C.c_int, [_p, _p, _i64, _i64, _i32, _f64, _pp]),
    "rphip_ease_fit": (C.c_int, [_p, _p, _f64, _f64, _pp]),
    "rphip_model_dense_download": (C.c_int, [_p, _p, _p]),
    "rphip_model_dense_device": (C.c_int, [_p, _pp, C.POINTER(_i64), C.POINTER(_i64)]),
    "rphip_model_dense_prune": (C.c_int, [_p, _p, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "rphip_model_dense_changed": (C.c_int, [_p, _p]),
    "rphip_model_from_dense_host": (C.c_int, [_p, _i64, _p, _pp]),
    "rphip_model_empty_rows": (C.c_int, [_p, _p, C.POINTER(_i64)]),
    "rphip_ctx_trim": (C.c_int, [_p]),
    "rphip_ctx_set_option": (C.c_int, [_p, C.c_char_p, _i64]),
    "rphip_gemm_nt_f32": (C.c_int, [_p, _i64, _i64, _i64, C.c_float, _p, _i64, _p, _i64, C.c_float, _p, _i64]),
    "rphip_ease_xy_fit": (C.c_int, [_p, _p, _p, _f64, _f64, _pp]),
    "rphip_ease_gram_create": (C.c_int, [_p, _p, _pp]),
    "rphip_ease_fit_gram": (C.c_int, [_p, _p, _p, _f64, _f64, _pp]),
    "rphip_model_from_dense_cols": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _pp]),
    "rphip_predict_merge_topk": (C.c_int, [_p, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p]),
    "rphip_comm_rccl_unique_id": (C.c_int, [_p]),
    "rphip_comm_rccl_create": (C.c_int, [_p, _i32, _i32, _p, _pp]),
    "rphip_comm_from_ops": (C.c_int, [_p, _p, _pp]),
    "rphip_ease_model_allgather": (C.c_int, [_p, _p, _p, _i64, _i64, _pp]),
    "rphip_comm_destroy": (C.c_int, [_p]),
    "rphip_comm_info": (C.c_int, [_p, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "rphip_comm_all_reduce_sum_i32": (C.c_int, [_p, _p, _p, _i64]),
    "rphip_ease_fit_dist": (C.c_int, [_p, _p, _p, _f64, _f64, _i32, _pp]),
    "rphip_model_dense_layout": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)]),
    "rphip_itemknn_gather": (C.c_int, [_p, _p, _p, _p, _pp]),
    "rphip_ease_model_allgather_i8": (C.c_int, [_p, _p, _p, _i64, _i64, _pp]),
    "rphip_predict_topk_owner": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _p, _p, _p]),
    "rphip_predict_topk_dist": (C.c_int, [_p, _p, _p, _p, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64), _p, _p, _p]),
    "rphip_comm_selftest": (C.c_int, [_p, _p, _i32, _i64, C.POINTER(_i64), C.POINTER(_f64)]),
}
# include/recpack_hip_diag.h: measuring instruments and test hooks (tools/, tests/), not the drop-in boundary
DIAG_PROTOTYPES = {
    "rphip_model_adopt_gather": (C.c_int, [_p, _p]),
    "rphip_potrf_trtri_128": (C.c_int, [_p, _p, _i64, _p, _p]),
    "rphip_gemm_nt_f32_ex": (C.c_int, [_p, _i64, _i64, _i64, _i64, C.c_float, _p, _i64, _p, _i64, C.c_float, _p, _i64, _i32, _i32, _i64, _i64, _i32]),
    "rphip_comm_loopback_create": (C.c_int, [_p, _i32, _i32, C.c_double, C.c_double, _pp]),
    "rphip_ease_fit_dist_stats": (C.c_int, [_p, _p]),
    "rphip_predict_owner_stats": (C.c_int, [_p, _p]),
    "rphip_predict_stats": (C.c_int, [_p, _p]),
}

_lock = threading.Lock()
_lib = None


def _build_in_tree():
    """`make -C recpack_amd/csrc` when hipcc is there and the library has not been built yet.  Serialised across
    processes by an flock on csrc/.build.lock (a torchrun start on a fresh checkout: one rank builds, the others wait
    on the lock and find the finished library); make links to a temporary name and renames it, so a half-linked file
    is never dlopened.  make's output is shown when it fails."""
    from .build import build_hip

    try:
        build_hip(jobs=6, quiet=True, only_if_missing=True)
    except Exception as e:  # the loader's own message follows
        sys.stderr.write(f"recpack_amd: in-tree build failed: {e}\n")


def load_library(path: str | None = None):
    """dlopen the library and attach prototypes. Raises RecpackHipError if it is not built."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p) and path is None:
            _build_in_tree()  # a fresh checkout: the library is a build product (never a different code path)
        if not os.path.exists(p):
            raise RecpackHipError(-100, f"{p} not found - build it first (make -C recpack_amd/csrc); there is no CPU fallback")
        try:
            # torch bundles its own libamdhip64.so.7: import it first so that both share ONE HIP runtime
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for single-GPU host-array use
            pass
        lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
        for name, (res, args) in list(PROTOTYPES.items()) + list(DIAG_PROTOTYPES.items()):
            fn = getattr(lib, name)  # AttributeError => the .so is older than the header
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def _torch_inputs_ready(ctx=None):
    """The library runs on its own HIP stream: tensors a caller has just produced on torch's current stream must be
    complete before the library reads them.  Nothing to do when the context was created on that very stream."""
    import torch

    cur = torch.cuda.current_stream()
    if ctx is not None and ctx.stream is not None and int(cur.cuda_stream) == ctx.stream:
        return
    cur.synchronize()


def check(code: int):
    if code != 0:
        msg = load_library().rphip_last_error()
        raise RecpackHipError(code, msg.decode("utf-8", "replace") if msg else "")


class _Child:
    """A native object made on a context (matrix, model, communicator): it points into the context, so it is destroyed before it
    - `Context.close` sees to that - and NOT after it: when the garbage collector finalises the context of an unreachable cycle first
    (any order is allowed there, and the weak references are gone by then), the child finds the context closed and leaves its native
    object alone.  Nothing here may depend on module globals: at interpreter exit they are None before the last objects die."""

    _destroy = ""        # the library's destructor for this kind
    _close_last = False  # communicators go after the models that may still hold operations on them

    def close(self):
        ctx = getattr(self, "ctx", None)
        if getattr(self, "handle", None) and ctx is not None and getattr(ctx, "handle", None):
            getattr(ctx.lib, self._destroy)(self.handle)
        self.handle = None

    __del__ = close


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU + one stream (rphip_ctx)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = C.c_void_p()
        check(self.lib.rphip_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = int(device)
        self.stream = int(stream) if stream else None  # the caller's HIP stream, if the context runs on one
        self._children = weakref.WeakSet()  # matrices, models and communicators made on this context (they point into it)

    def close(self):
        """Destroy the context - after everything that was made on it: a matrix, model or communicator that is still alive is closed
        first (its native object holds a pointer to the context).  When the garbage collector finalises a context BEFORE its children
        (objects of one unreachable cycle are finalised in any order, and the weak references to them are already gone), the children
        find the context closed and leave their native objects alone: see `_Child`."""
        if getattr(self, "handle", None):
            for child in sorted(list(getattr(self, "_children", ())), key=lambda c: c._close_last):  # (communicators last)
                child.close()
            self.lib.rphip_ctx_destroy(self.handle)
            self.handle = None

    __del__ = close

    def trim(self):
        """Release the context's cached device blocks (rphip_ctx_trim)."""
        check(self.lib.rphip_ctx_trim(self.handle))

    def set_option(self, name: str, value: int):
        """rphip_ctx_set_option: "predict_exact64" (0/1), "predict_tile32" (2048/4096), "cache_limit_bytes"."""
        check(self.lib.rphip_ctx_set_option(self.handle, name.encode(), int(value)))

    def sync(self):
        check(self.lib.rphip_ctx_sync(self.handle))

    def timings(self) -> dict:
        t = Timings()
        check(self.lib.rphip_get_timings(self.handle, C.byref(t)))
        return t.as_dict()

    def counters(self) -> np.ndarray:
        out = np.zeros(8, dtype=np.int64)
        check(self.lib.rphip_get_counters(self.handle, _np_ptr(out)))
        return out

    def reset_timings(self):
        check(self.lib.rphip_reset_timings(self.handle))

    def device_info(self) -> dict:
        ncu, hbm = C.c_int(), C.c_int64()
        buf = C.create_string_buffer(64)
        check(self.lib.rphip_device_info(self.handle, C.byref(ncu), C.byref(hbm), buf, 64))
        return {"n_cu": ncu.value, "hbm_bytes": hbm.value, "arch": buf.value.decode()}


class DeviceCSR(_Child):
    """Device-resident binary interaction matrix (rphip_csr): X, X^T, item degrees."""

    _destroy = "rphip_csr_destroy"

    def __init__(self, ctx: Context, handle, keepalive=None):
        self.ctx, self.handle, self._keep = ctx, handle, keepalive
        ctx._children.add(self)
        nu, ni, nz = C.c_int64(), C.c_int64(), C.c_int64()
        check(ctx.lib.rphip_csr_shape(handle, C.byref(nu), C.byref(ni), C.byref(nz)))
        self.shape = (nu.value, ni.value)
        self.nnz = nz.value

    @classmethod
    def from_host(cls, ctx: Context, indptr: np.ndarray, indices: np.ndarray, shape):
        indptr = np.ascontiguousarray(indptr)
        if indptr.dtype not in (np.int32, np.int64):
            indptr = indptr.astype(np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        h = C.c_void_p()
        check(ctx.lib.rphip_csr_from_host(ctx.handle, int(shape[0]), int(shape[1]), int(indices.shape[0]), _np_ptr(indptr),
                                          1 if indptr.dtype == np.int64 else 0, _np_ptr(indices), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device(cls, ctx: Context, indptr_t, indices_t, shape, fit_rows=None):
        """indptr_t / indices_t: torch int32 CUDA tensors (kept alive by this object).  ``fit_rows=(begin, end)``: this
        rank only fits those item rows (multi-GPU shard): X^T is filled for them alone."""
        _torch_inputs_ready(ctx)
        h = C.c_void_p()
        b, e = (0, -1) if fit_rows is None else (int(fit_rows[0]), int(fit_rows[1]))
        check(ctx.lib.rphip_csr_from_device_rows(ctx.handle, int(shape[0]), int(shape[1]), int(indices_t.numel()),
                                                 C.c_void_p(indptr_t.data_ptr()), C.c_void_p(indices_t.data_ptr()), b, e, C.byref(h)))
        return cls(ctx, h, keepalive=(indptr_t, indices_t))

    @classmethod
    def from_device_shard(cls, ctx: Context, indptr_t, indices_t, shape, fit_rows):
        """A shard's matrix whose degree pass is sharded too (rphip_csr_from_device_shard): degrees hold the item rows
        ``fit_rows`` only until the caller has all-reduced ``degrees_device_ptr()`` and called ``finish_degrees()``
        (recpack_amd.dist.csr_for_shard does both)."""
        _torch_inputs_ready(ctx)
        h = C.c_void_p()
        check(ctx.lib.rphip_csr_from_device_shard(ctx.handle, int(shape[0]), int(shape[1]), int(indices_t.numel()),
                                                  C.c_void_p(indptr_t.data_ptr()), C.c_void_p(indices_t.data_ptr()), int(fit_rows[0]),
                                                  int(fit_rows[1]), C.byref(h)))
        return cls(ctx, h, keepalive=(indptr_t, indices_t))

    def degrees_device_ptr(self) -> int:
        p = C.c_void_p()
        check(self.ctx.lib.rphip_csr_degrees_device(self.handle, C.byref(p)))
        return int(p.value)

    def finish_degrees(self):
        _torch_inputs_ready(self.ctx)
        check(self.ctx.lib.rphip_csr_finish_degrees(self.ctx.handle, self.handle))

    @classmethod
    def from_coo(cls, ctx: Context, uid, iid, shape):
        """From raw (user id, item id) pairs - numpy int arrays, or torch int64 CUDA tensors - in any order, repeats
        allowed: the distinct pairs become the binary matrix (InteractionMatrix.values + to_binary, built on the GPU)."""
        h = C.c_void_p()
        if hasattr(uid, "data_ptr"):  # torch
            import torch

            if uid.dtype != torch.int64 or iid.dtype != torch.int64 or not uid.is_cuda or not iid.is_cuda:
                raise TypeError("from_coo: torch inputs must be int64 CUDA tensors")
            uid, iid = uid.contiguous(), iid.contiguous()
            _torch_inputs_ready(ctx)
            check(ctx.lib.rphip_csr_from_coo_device(ctx.handle, int(shape[0]), int(shape[1]), int(uid.numel()), C.c_void_p(uid.data_ptr()),
                                                    C.c_void_p(iid.data_ptr()), C.byref(h)))
            ctx.sync()  # the pairs may be freed by the caller right away
        else:
            uid = np.ascontiguousarray(uid, dtype=np.int64)
            iid = np.ascontiguousarray(iid, dtype=np.int64)
            if uid.shape != iid.shape or uid.ndim != 1:
                raise ValueError("from_coo: uid and iid must be 1-d arrays of the same length")
            check(ctx.lib.rphip_csr_from_coo_host(ctx.handle, int(shape[0]), int(shape[1]), int(uid.shape[0]), _np_ptr(uid), _np_ptr(iid),
                                                  C.byref(h)))
        return cls(ctx, h)

    def download(self):
        """(indptr int32[n_users+1], indices int32[nnz]) of X."""
        indptr = np.empty(self.shape[0] + 1, dtype=np.int32)
        indices = np.empty(max(self.nnz, 1), dtype=np.int32)
        check(self.ctx.lib.rphip_csr_download(self.ctx.handle, self.handle, _np_ptr(indptr), _np_ptr(indices)))
        return indptr, indices[: self.nnz]

    @classmethod
    def from_host_valued(cls, ctx: Context, X):
        """From a scipy CSR with float values (sorted indices, no duplicates): pattern + values (rphip_csr_set_values)."""
        m = cls.from_host(ctx, X.indptr, X.indices, X.shape)
        try:
            m.set_values(X.data)
        except Exception:
            m.close()
            raise
        return m

    def set_values(self, values):
        """Attach float64 values to the stored entries (order of the uploaded indices); None detaches them."""
        if values is None:
            check(self.ctx.lib.rphip_csr_set_values(self.ctx.handle, self.handle, None))
            return
        values = np.ascontiguousarray(values, dtype=np.float64)
        if values.shape != (self.nnz,):
            raise ValueError(f"set_values: {values.shape} values for {self.nnz} stored entries")
        if self.nnz == 0:
            values = np.zeros(1)
        check(self.ctx.lib.rphip_csr_set_values(self.ctx.handle, self.handle, _np_ptr(values)))

    def item_degrees(self) -> np.ndarray:
        out = np.empty(self.shape[1], dtype=np.int32)
        check(self.ctx.lib.rphip_csr_item_degrees(self.ctx.handle, self.handle, _np_ptr(out)))
        return out

    def num_products(self) -> int:
        p = C.c_int64()
        check(self.ctx.lib.rphip_csr_num_products(self.ctx.handle, self.handle, C.byref(p)))
        return p.value



class Model(_Child):
    """Fitted similarity on the device (rphip_model)."""

    _destroy = "rphip_model_destroy"

    def __init__(self, ctx: Context, handle):
        self.ctx, self.handle = ctx, handle
        ctx._children.add(self)
        kind, K = C.c_int32(), C.c_int32()
        ni, r0, r1 = C.c_int64(), C.c_int64(), C.c_int64()
        check(ctx.lib.rphip_model_info(handle, C.byref(kind), C.byref(ni), C.byref(r0), C.byref(r1), C.byref(K)))
        self.kind, self.n_items, self.row_begin, self.row_end, self.K = kind.value, ni.value, r0.value, r1.value, K.value

    @property
    def n_rows(self):
        return self.row_end - self.row_begin

    def topk_download(self):
        idx = np.empty((self.n_rows, self.K), dtype=np.int32)
        val = np.empty((self.n_rows, self.K), dtype=np.float64)
        nnz = np.empty(self.n_rows, dtype=np.int32)
        check(self.ctx.lib.rphip_model_topk_download(self.ctx.handle, self.handle, _np_ptr(idx), _np_ptr(val), _np_ptr(nnz)))
        return idx, val, nnz

    def topk_device_ptrs(self):
        pi, pv, pn = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rows, K = C.c_int64(), C.c_int32()
        check(self.ctx.lib.rphip_model_topk_device(self.handle, C.byref(pi), C.byref(pv), C.byref(pn), C.byref(rows), C.byref(K)))
        return pi.value, pv.value, pn.value, rows.value, K.value

    @classmethod
    def from_topk_host(cls, ctx: Context, idx: np.ndarray, val: np.ndarray, nnz: np.ndarray):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        nnz = np.ascontiguousarray(nnz, dtype=np.int32)
        h = C.c_void_p()
        check(ctx.lib.rphip_model_from_topk_host(ctx.handle, idx.shape[0], idx.shape[1], _np_ptr(idx), _np_ptr(val), _np_ptr(nnz), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_topk_device(cls, ctx: Context, idx_t, val_t, nnz_t):
        """torch CUDA tensors: idx int32[I,K], val float64[I,K], nnz int32[I] (copied)."""
        _torch_inputs_ready(ctx)
        h = C.c_void_p()
        check(ctx.lib.rphip_model_from_topk_device(ctx.handle, int(idx_t.shape[0]), int(idx_t.shape[1]), C.c_void_p(idx_t.data_ptr()),
                                                   C.c_void_p(val_t.data_ptr()), C.c_void_p(nnz_t.data_ptr()), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_dense_cols(cls, ctx: Context, n_items: int, col_begin: int, n_cols: int, B_t):
        """A dense model on a COLUMN SLAB: B_t[i, q] = B[i, col_begin + q] for all n_items rows (torch float32 CUDA matrix, not
        copied).  predict_topk* against it scores only those columns (rphip_model_from_dense_cols)."""
        _torch_inputs_ready(ctx)
        h = C.c_void_p()
        check(ctx.lib.rphip_model_from_dense_cols(ctx.handle, int(n_items), int(col_begin), int(n_cols), C.c_void_p(B_t.data_ptr()), B_t.stride(0),
                                                  C.byref(h)))
        m = cls(ctx, h)
        m._keep = B_t
        return m

    @classmethod
    def from_dense_host(cls, ctx: Context, B: np.ndarray):
        B = np.ascontiguousarray(B, dtype=np.float32)
        assert B.ndim == 2 and B.shape[0] == B.shape[1]
        h = C.c_void_p()
        check(ctx.lib.rphip_model_from_dense_host(ctx.handle, B.shape[0], _np_ptr(B), C.byref(h)))
        return cls(ctx, h)

    def empty_rows(self) -> int:
        v = C.c_int64()
        check(self.ctx.lib.rphip_model_empty_rows(self.ctx.handle, self.handle, C.byref(v)))
        return v.value

    def dense_device_ptr(self):
        """(device pointer of B, n_items, leading dimension in floats) of a dense model."""
        ptr, n, ld = C.c_void_p(), C.c_int64(), C.c_int64()
        check(self.ctx.lib.rphip_model_dense_device(self.handle, C.byref(ptr), C.byref(n), C.byref(ld)))
        return ptr.value, n.value, ld.value

    def dense_layout(self) -> dict:
        """Which items the columns of a dense model's B are (rphip_model_dense_layout): ``n_local`` columns; with ``panel`` > 0
        they are block-cyclic panels with reflected odd rounds (local panel q = l // panel is the matrix's panel q * world + rank for even q,
        q * world + world - 1 - rank for odd q), otherwise the
        range starting at ``col_begin``."""
        n, nb, cb = C.c_int64(), C.c_int64(), C.c_int64()
        w, r = C.c_int32(), C.c_int32()
        check(self.ctx.lib.rphip_model_dense_layout(self.handle, C.byref(n), C.byref(nb), C.byref(w), C.byref(r), C.byref(cb)))
        return {"n_local": n.value, "panel": nb.value, "world": w.value, "rank": r.value, "col_begin": cb.value}

    def local_columns(self) -> np.ndarray:
        """Item ids of the local columns of a dense model (ascending)."""
        lay = self.dense_layout()
        l = np.arange(lay["n_local"], dtype=np.int64)
        if lay["panel"] > 0:  # rp_common.h rp_cyc_panel: block-cyclic with reflected odd rounds
            q = l // lay["panel"]
            pos = np.where(q % 2 == 1, lay["world"] - 1 - lay["rank"], lay["rank"])
            return (q * lay["world"] + pos) * lay["panel"] + l % lay["panel"]
        return lay["col_begin"] + l

    def dense_prune(self, keep: int):
        """Keep the ``keep`` largest |B_ij| on the device (ease.py:97-109); returns (nnz before, nnz after)."""
        a, b = C.c_int64(), C.c_int64()
        check(self.ctx.lib.rphip_model_dense_prune(self.ctx.handle, self.handle, int(keep), C.byref(a), C.byref(b)))
        return a.value, b.value

    def dense_changed(self):
        """The caller rewrote the borrowed matrix of this dense model: the two-pass predict's quantised copy is rebuilt on next use."""
        check(self.ctx.lib.rphip_model_dense_changed(self.ctx.handle, self.handle))

    def dense_download(self) -> np.ndarray:
        out = np.empty((self.n_items, self.n_items), dtype=np.float32)
        check(self.ctx.lib.rphip_model_dense_download(self.ctx.handle, self.handle, _np_ptr(out)))
        return out

    def adopt_gather(self, other: "Model"):
        """(diagnostic: include/recpack_hip_diag.h) take over the pending panel gather of `other` - tools/dist_rank_emulate.py."""
        check(self.ctx.lib.rphip_model_adopt_gather(self.handle, other.handle))



# ---- communicator (include/recpack_hip.h: rphip_comm) --------------------------------------------------------------------
_BCAST_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
_AGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
_EXCH_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                      C.POINTER(C.c_int64), C.c_void_p)
_ARED_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class CommOps(C.Structure):
    _fields_ = [("self", C.c_void_p), ("world", C.c_int32), ("rank", C.c_int32), ("broadcast", _BCAST_T), ("all_gather", _AGATHER_T),
                ("exchange", _EXCH_T), ("all_reduce_sum_i32", _ARED_T)]


class _DevBytes:
    """Zero-copy uint8 view of device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class Comm(_Child):
    """A group of ranks, one GPU each, with stream-ordered collectives on device memory (rphip_comm).

    ``Comm.rccl`` is the library's own RCCL communicator (the product path on a multi-GPU node); ``Comm.torch`` hands the
    library a table of callbacks that run the same operations through a torch.distributed group - any backend, so the
    distributed solve can be driven over gloo with the ranks sharing a GPU (tests)."""

    _destroy = "rphip_comm_destroy"
    _close_last = True

    def __init__(self, ctx: Context, handle, keep=None):
        self.ctx, self.handle, self._keep = ctx, handle, keep
        ctx._children.add(self)
        w, r, k = C.c_int32(), C.c_int32(), C.c_int32()
        check(ctx.lib.rphip_comm_info(handle, C.byref(w), C.byref(r), C.byref(k)))
        self.world, self.rank, self.is_rccl = w.value, r.value, bool(k.value)
        self.error = None

    @classmethod
    def rccl(cls, ctx: Context, group=None):
        """The library's RCCL communicator over the ranks of a torch.distributed group (only the 128-byte id travels through
        torch.distributed; a process outside any group gets a one-rank communicator)."""
        import torch.distributed as dist

        ident = C.create_string_buffer(128)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if rank == 0:
            check(ctx.lib.rphip_comm_rccl_unique_id(ident))
        if world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        check(ctx.lib.rphip_comm_rccl_create(ctx.handle, world, rank, ident, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def torch(cls, ctx: Context, device, group=None):
        """Callbacks over a torch.distributed group.  Every operation first waits for the stream it was ordered on, then runs
        (through host memory when the backend is gloo) and is complete when the callback returns."""
        import torch
        import torch.distributed as dist

        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        via_host = dist.is_initialized() and dist.get_backend(group) != "nccl"
        box = {"error": None}

        def grank(r):
            return dist.get_global_rank(group, r) if group is not None else r

        def view(ptr, nbytes):
            return torch.as_tensor(_DevBytes(ptr, nbytes), device=device)

        def wait(stream):
            if stream:
                torch.cuda.ExternalStream(int(stream), device=device).synchronize()
            else:
                torch.cuda.synchronize(device)

        def guarded(fn):
            def run(*a):
                try:
                    fn(*a)
                    torch.cuda.synchronize(device)
                    return 0
                except Exception as e:  # noqa: BLE001 - reported to the library as a failed transport call
                    box["error"] = e
                    return 1
            return run

        def bcast(_self, buf, nbytes, root, stream):
            wait(stream)
            t = view(buf, nbytes)
            if via_host:
                h = t.cpu()
                dist.broadcast(h, src=grank(root), group=group)
                if rank != root:
                    t.copy_(h)
            else:
                dist.broadcast(t, src=grank(root), group=group)

        def agather(_self, send, recv, nbytes, stream):
            wait(stream)
            s, r = view(send, nbytes), view(recv, nbytes * world)
            if via_host:
                h = torch.empty(nbytes * world, dtype=torch.uint8)
                dist.all_gather_into_tensor(h, s.cpu(), group=group)
                r.copy_(h)
            else:
                dist.all_gather_into_tensor(r, s, group=group)

        def exch(_self, send, soff, sby, recv, roff, rby, stream):
            wait(stream)
            ops, landed = [], []
            for p in range(world):
                if p == rank:
                    if sby[p] > 0:
                        view(recv + roff[p], sby[p]).copy_(view(send + soff[p], sby[p]))
                    continue
                if sby[p] > 0:
                    t = view(send + soff[p], sby[p])
                    ops.append(dist.P2POp(dist.isend, t.cpu() if via_host else t, grank(p), group))
                if rby[p] > 0:
                    t = view(recv + roff[p], rby[p])
                    h = torch.empty(rby[p], dtype=torch.uint8) if via_host else t
                    landed.append((t, h))
                    ops.append(dist.P2POp(dist.irecv, h, grank(p), group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            if via_host:
                for t, h in landed:
                    t.copy_(h)

        def ared(_self, buf, count, stream):
            wait(stream)
            t = view(buf, count * 4).view(torch.int32)
            if via_host:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

        cbs = (_BCAST_T(guarded(bcast)), _AGATHER_T(guarded(agather)), _EXCH_T(guarded(exch)), _ARED_T(guarded(ared)))
        ops = CommOps(None, world, rank, *cbs)
        h = C.c_void_p()
        check(ctx.lib.rphip_comm_from_ops(ctx.handle, C.byref(ops), C.byref(h)))
        c = cls(ctx, h, keep=(cbs, ops, box))
        c.error = box
        return c

    @classmethod
    def loopback(cls, ctx: Context, world: int, rank: int, gbytes_per_s: float = 100.0, latency_us: float = 10.0):
        """ONE rank of a ``world``-rank run measured alone on one GPU (rphip_comm_loopback_create): the collectives move nothing
        between ranks and hold their stream for the time the modelled link would take.  Results are meaningless; the kernel sequence
        is rank ``rank``'s (tools/dist_rank_emulate.py)."""
        h = C.c_void_p()
        check(ctx.lib.rphip_comm_loopback_create(ctx.handle, int(world), int(rank), float(gbytes_per_s), float(latency_us), C.byref(h)))
        return cls(ctx, h)

    def all_reduce_sum_i32(self, ptr: int, count: int):
        check(self.ctx.lib.rphip_comm_all_reduce_sum_i32(self.ctx.handle, self.handle, C.c_void_p(ptr), int(count)))

    SELFTEST_OPS = ("broadcast", "all_gather", "exchange", "all_reduce_sum_i32", "broadcast_large")

    def selftest_op(self, op: int, words: int = 1 << 18):
        """One operation of the transport on a known pattern, every received word checked on the device (rphip_comm_selftest):
        (mismatching words, milliseconds).  Blocks until the operation is complete - put a watchdog around it."""
        bad, ms = C.c_int64(-1), C.c_double(0.0)
        check(self.ctx.lib.rphip_comm_selftest(self.ctx.handle, self.handle, int(op), int(words), C.byref(bad), C.byref(ms)))
        return bad.value, ms.value

    def selftest(self, words: int = 1 << 18, watchdog_s: float = 30.0, on_hang=None) -> dict:
        """All four operations once (every rank must call this).  A watchdog thread fires `on_hang(name)` - default: print which
        operation hangs and leave the process with exit code 6 - when one of them does not complete within `watchdog_s` seconds:
        a hung collective cannot be cancelled from inside the process.  Returns {"ok": bool, "ops": {name: {"mismatches", "ms"}}}."""
        import threading

        def default_hang(name):
            sys.stderr.write(f"recpack_amd: transport self-test: {name} did not complete within {watchdog_s:.0f} s (rank {self.rank} of {self.world})\n")
            sys.stderr.flush()
            os._exit(6)

        out = {"ok": True, "world": self.world, "transport": "rccl" if self.is_rccl else "callbacks", "words_per_rank": int(words), "ops": {}}
        for op, name in enumerate(self.SELFTEST_OPS):
            dog = threading.Timer(watchdog_s, (on_hang or default_hang), args=(name,))
            dog.daemon = True
            dog.start()
            try:
                bad, ms = self.selftest_op(op, words)
            finally:
                dog.cancel()
            out["ops"][name] = {"mismatches": int(bad), "ms": round(ms, 3)}
            out["ok"] = out["ok"] and bad == 0
        return out



def itemknn_gather(ctx: Context, comm: Comm, local: Model, row_bounds) -> Model:
    """The ranks' row shards of a multi-GPU ItemKNN fit into one model with all rows (rphip_itemknn_gather: one all-gather through
    the communicator, landing in the model's own arrays).  ``row_bounds``: world + 1 row indices, 0 .. n_items."""
    b = np.ascontiguousarray(row_bounds, dtype=np.int64)
    assert b.shape == (comm.world + 1,)
    h = C.c_void_p()
    try:
        check(ctx.lib.rphip_itemknn_gather(ctx.handle, comm.handle, local.handle, _np_ptr(b), C.byref(h)))
    except RecpackHipError:
        err = comm.error["error"] if isinstance(comm.error, dict) else None
        if err is not None:
            comm.error["error"] = None
            raise err
        raise
    return Model(ctx, h)


def ease_fit_dist(ctx: Context, comm: Comm, X: DeviceCSR, l2: float, alpha: float = 0.0, panel_tiles: int = 0):
    """EASE fit over the ranks of ``comm`` with the matrix sharded in memory (rphip_ease_fit_dist: block-cyclic panels,
    symmetric block sweeps, n^3 / world flop per rank).  Returns this rank's model (its columns of B: ``Model.local_columns``)
    or None on a rank that owns only padding."""
    h = C.c_void_p()
    try:
        check(ctx.lib.rphip_ease_fit_dist(ctx.handle, comm.handle, X.handle, float(l2), float(alpha), int(panel_tiles), C.byref(h)))
    except RecpackHipError:
        err = comm.error["error"] if isinstance(comm.error, dict) else None
        if err is not None:
            comm.error["error"] = None
            raise err
        raise
    return Model(ctx, h) if h.value else None


def ease_model_allgather(ctx: Context, comm: Comm, model, n_items: int, panel_cols: int) -> Model:
    """Every rank's column panels of a distributed EASE fit into one full model on this rank (rphip_ease_model_allgather): predict
    then shards by users, as for ItemKNN.  ``model`` may be None on a rank that owns only padding."""
    h = C.c_void_p()
    try:
        check(ctx.lib.rphip_ease_model_allgather(ctx.handle, comm.handle, model.handle if model is not None else None, int(n_items), int(panel_cols),
                                                 C.byref(h)))
    except RecpackHipError:
        err = comm.error["error"] if isinstance(comm.error, dict) else None
        if err is not None:
            comm.error["error"] = None
            raise err
        raise
    return Model(ctx, h)


def _comm_call(comm: Comm, fn):
    """Run a library call that may invoke the communicator's callbacks: an exception raised inside one of them is re-raised here."""
    try:
        return fn()
    except RecpackHipError:
        err = comm.error["error"] if isinstance(comm.error, dict) else None
        if err is not None:
            comm.error["error"] = None
            raise err
        raise


def ease_model_allgather_i8(ctx: Context, comm: Comm, model, n_items: int, panel_cols: int) -> Model:
    """The gathered model's int8 rows alone (rphip_ease_model_allgather_i8): a fifth of the full gathered model's memory, what
    :func:`predict_topk_owner` needs on every rank.  ``model`` may be None on a rank that owns only padding."""
    h = C.c_void_p()
    _comm_call(comm, lambda: check(ctx.lib.rphip_ease_model_allgather_i8(ctx.handle, comm.handle, model.handle if model is not None else None,
                                                                        int(n_items), int(panel_cols), C.byref(h))))
    return Model(ctx, h)


def predict_topk_owner(ctx: Context, comm: Comm, local, rows_i8: Model, X: DeviceCSR, user_bounds, Kp: int, idx_t, val_t, nnz_t,
                       exclude_history: bool = False):
    """Top-Kp of this rank's users ``[user_bounds[rank], user_bounds[rank + 1])`` with the exact scores formed by the owners of the
    columns (rphip_predict_topk_owner): first pass on the gathered int8 rows, candidates all-gathered, sums from every rank's own
    panels, one exchange back, ranking and proof here; the single-GPU result bit for bit.  ``user_bounds``: world + 1 ints, the same
    on every rank.  The outputs (torch CUDA tensors) hold this rank's users."""
    _torch_inputs_ready(ctx)
    ub = np.ascontiguousarray(np.asarray(user_bounds, dtype=np.int64))
    if ub.shape != (comm.world + 1,):
        raise ValueError(f"user_bounds must hold world + 1 = {comm.world + 1} entries")
    _comm_call(comm, lambda: check(ctx.lib.rphip_predict_topk_owner(ctx.handle, comm.handle, local.handle if local is not None else None, rows_i8.handle,
                                                                   X.handle, _np_ptr(ub), int(Kp), 1 if exclude_history else 0,
                                                                   C.c_void_p(idx_t.data_ptr()), C.c_void_p(val_t.data_ptr()),
                                                                   C.c_void_p(nnz_t.data_ptr()))))


def predict_topk_dist(ctx: Context, comm: Comm, model, X: DeviceCSR, Kp: int, idx_t, val_t, nnz_t, exclude_history: bool = False):
    """Top-Kp of every user with the model's columns spread over the ranks of ``comm`` (rphip_predict_topk_dist): local lists,
    one exchange per chunk inside the library, merge on the rank that owns the user.  The outputs (torch CUDA tensors of
    ceil(n_users / world) rows) receive this rank's users; returns (first user, number of users)."""
    _torch_inputs_ready(ctx)
    ub, nu = C.c_int64(), C.c_int64()
    try:
        check(ctx.lib.rphip_predict_topk_dist(ctx.handle, comm.handle, model.handle if model is not None else None, X.handle, int(Kp),
                                              1 if exclude_history else 0, C.byref(ub), C.byref(nu), C.c_void_p(idx_t.data_ptr()),
                                              C.c_void_p(val_t.data_ptr()), C.c_void_p(nnz_t.data_ptr())))
    except RecpackHipError:
        err = comm.error["error"] if isinstance(comm.error, dict) else None
        if err is not None:
            comm.error["error"] = None
            raise err
        raise
    return ub.value, nu.value


def predict_owner_stats(ctx: Context) -> dict:
    """The last :func:`predict_topk_owner` of this context (diagnostic): users without a proof over all ranks / of this rank,
    candidates per user, and on a loop-back communicator the microseconds of the stand-in for the other ranks' parts."""
    out = np.zeros(4, dtype=np.int64)
    check(ctx.lib.rphip_predict_owner_stats(ctx.handle, _np_ptr(out)))
    return {"unproven_all_ranks": int(out[0]), "unproven_here": int(out[1]), "candidates_per_user": int(out[2]), "loopback_standin_us": int(out[3])}


def predict_stats(ctx: Context) -> dict:
    """The last single-GPU ``predict_topk*`` of this context (diagnostic): users the two-pass predict sent to the exact pass, users of the
    call (both -1 when the call took another route), candidates per user, whether the first pass was the per-user int8 kernel, how many
    users were given a second, longer list and how many of them it proved."""
    out = np.zeros(6, dtype=np.int64)
    check(ctx.lib.rphip_predict_stats(ctx.handle, _np_ptr(out)))
    return {"exact_pass_users": int(out[0]), "users": int(out[1]), "candidates_per_user": int(out[2]), "per_user_first_pass": bool(out[3]),
            "second_list_users": int(out[4]), "second_list_proven": int(out[5])}


def ease_fit_dist_stats(ctx: Context) -> dict:
    out = np.zeros(8, dtype=np.int64)
    check(ctx.lib.rphip_ease_fit_dist_stats(ctx.handle, _np_ptr(out)))
    return dict(zip(("slab_bytes", "other_peak_bytes", "bytes_received", "bytes_sent", "pivots", "panel", "n_pad", "flop"), (int(v) for v in out)))


SIM_COSINE, SIM_CONDITIONAL_PROBABILITY = 0, 1


def itemknn_fit(ctx: Context, X: DeviceCSR, K: int, row_begin: int = 0, row_end: int | None = None, similarity: int = SIM_COSINE,
                pop_discount: float = 0.0, normalize_X: bool = False) -> Model:
    h = C.c_void_p()
    row_end = X.shape[1] if row_end is None else row_end
    if similarity == SIM_COSINE and not normalize_X:
        check(ctx.lib.rphip_itemknn_fit(ctx.handle, X.handle, int(row_begin), int(row_end), int(K), C.byref(h)))
    else:
        check(ctx.lib.rphip_itemknn_fit_ex(ctx.handle, X.handle, int(row_begin), int(row_end), int(K), int(similarity),
                                           float(pop_discount), 1 if normalize_X else 0, C.byref(h)))
    return Model(ctx, h)


def itemknn_fit_values(ctx: Context, X: DeviceCSR, K: int, left_binary: bool = False, row_scale=None, row_begin: int = 0,
                       row_end=None) -> Model:
    """Top-K of G = L^T X on a valued matrix (L = X, or its pattern with left_binary), rows scaled by row_scale."""
    h = C.c_void_p()
    row_end = X.shape[1] if row_end is None else row_end
    rs = None
    if row_scale is not None:
        rs = np.ascontiguousarray(row_scale, dtype=np.float64)
        if rs.shape != (X.shape[1],):
            raise ValueError("row_scale must have one entry per item")
    check(ctx.lib.rphip_itemknn_fit_values(ctx.handle, X.handle, int(row_begin), int(row_end), int(K), 1 if left_binary else 0,
                                           _np_ptr(rs) if rs is not None else None, C.byref(h)))
    return Model(ctx, h)


DECAY_KINDS = {"none": 1, "exponential": 2, "convex": 3, "concave": 4, "log": 5, "linear": 6, "inverse": 7}


def itemknn_fit_timedist(ctx: Context, X: DeviceCSR, K: int, decay_function: str, decay: float, max_distance: float, row_scale=None,
                         row_begin: int = 0, row_end=None) -> Model:
    """Top-K of S_ij = row_scale_i * sum_u decay(|t_uj - t_ui|) on a matrix whose values are timestamps > 0."""
    h = C.c_void_p()
    row_end = X.shape[1] if row_end is None else row_end
    rs = None
    if row_scale is not None:
        rs = np.ascontiguousarray(row_scale, dtype=np.float64)
        if rs.shape != (X.shape[1],):
            raise ValueError("row_scale must have one entry per item")
    check(ctx.lib.rphip_itemknn_fit_timedist(ctx.handle, X.handle, int(row_begin), int(row_end), int(K), int(DECAY_KINDS[decay_function]),
                                             float(decay), float(max_distance), _np_ptr(rs) if rs is not None else None, C.byref(h)))
    return Model(ctx, h)


def itemknn_fit_hermann(ctx: Context, X: DeviceCSR, K: int, now: float, row_begin: int = 0, row_end=None) -> Model:
    """Top-K of the average over the common users of 1 / (|t_uj - t_ui| + now - min(t_ui, t_uj)) (values = timestamps > 0)."""
    h = C.c_void_p()
    row_end = X.shape[1] if row_end is None else row_end
    check(ctx.lib.rphip_itemknn_fit_hermann(ctx.handle, X.handle, int(row_begin), int(row_end), int(K), float(now), C.byref(h)))
    return Model(ctx, h)


def ease_fit(ctx: Context, X: DeviceCSR, l2: float, alpha: float = 0.0) -> Model:
    h = C.c_void_p()
    check(ctx.lib.rphip_ease_fit(ctx.handle, X.handle, float(l2), float(alpha), C.byref(h)))
    return Model(ctx, h)


def ease_gram_create(ctx: Context, X: DeviceCSR) -> Model:
    """G = X^T X kept on the device for a sweep over l2 (a handle for ease_fit_gram, not a model to predict with)."""
    h = C.c_void_p()
    check(ctx.lib.rphip_ease_gram_create(ctx.handle, X.handle, C.byref(h)))
    return Model(ctx, h)


def ease_fit_gram(ctx: Context, X: DeviceCSR, gram: Model, l2: float, alpha: float = 0.0) -> Model:
    h = C.c_void_p()
    check(ctx.lib.rphip_ease_fit_gram(ctx.handle, X.handle, gram.handle, float(l2), float(alpha), C.byref(h)))
    return Model(ctx, h)


def ease_xy_fit(ctx: Context, X: DeviceCSR, Y: DeviceCSR, l2: float, alpha: float = 0.0) -> Model:
    h = C.c_void_p()
    check(ctx.lib.rphip_ease_xy_fit(ctx.handle, X.handle, Y.handle, float(l2), float(alpha), C.byref(h)))
    return Model(ctx, h)


def gemm_nt(ctx: Context, A_t, B_t, C_t, alpha: float = 1.0, beta: float = 0.0):
    """C = alpha * A @ B.T + beta * C on torch float32 CUDA tensors (row-major, sizes multiples of 128 / 16)."""
    _torch_inputs_ready(ctx)
    M, K = A_t.shape
    N = B_t.shape[0]
    check(ctx.lib.rphip_gemm_nt_f32(ctx.handle, M, N, K, float(alpha), C.c_void_p(A_t.data_ptr()), A_t.stride(0),
                                    C.c_void_p(B_t.data_ptr()), B_t.stride(0), float(beta), C.c_void_p(C_t.data_ptr()), C_t.stride(0)))


def gemm_nt_ex(ctx: Context, A_t, B_t, C_t, k_begin: int, k_end: int, alpha: float = 1.0, beta: float = 0.0, k_from_row: bool = False,
               k_from_col: bool = False, m_off: int = 0, n_off: int = 0, lower_only: bool = False):
    """C (M x N) = alpha * A[:, k] @ B[:, k].T + beta * C over k in [k_begin, k_end), with per-tile K clipping for
    triangular operands (rphip_gemm_nt_f32_ex); torch float32 CUDA tensors / views with unit inner stride."""
    _torch_inputs_ready(ctx)
    M, N = C_t.shape
    check(ctx.lib.rphip_gemm_nt_f32_ex(ctx.handle, int(M), int(N), int(k_begin), int(k_end), float(alpha), C.c_void_p(A_t.data_ptr()),
                                       A_t.stride(0), C.c_void_p(B_t.data_ptr()), B_t.stride(0), float(beta), C.c_void_p(C_t.data_ptr()),
                                       C_t.stride(0), 1 if k_from_row else 0, 1 if k_from_col else 0, int(m_off), int(n_off),
                                       1 if lower_only else 0))


def potrf_trtri_128(ctx: Context, A_block_t, W_t, info_t):
    """In-place Cholesky of a 128 x 128 block (a view with row stride lda) and its inverse into W_t (128 x 128 contiguous)."""
    _torch_inputs_ready(ctx)
    check(ctx.lib.rphip_potrf_trtri_128(ctx.handle, C.c_void_p(A_block_t.data_ptr()), A_block_t.stride(0), C.c_void_p(W_t.data_ptr()),
                                        C.c_void_p(info_t.data_ptr())))


def predict_merge_topk(ctx: Context, cand_idx_t, cand_val_t, cand_nnz_t, idx_t, val_t, nnz_t):
    """Per-user merge of candidate lists: cand_idx_t / cand_val_t [U, L, Kp], cand_nnz_t [U, L] -> idx_t / val_t [U, Kp], nnz_t [U]."""
    _torch_inputs_ready(ctx)
    U, L, Kp = cand_idx_t.shape
    for t in (cand_idx_t, cand_val_t, cand_nnz_t, idx_t, val_t, nnz_t):
        assert t.is_contiguous()
    check(ctx.lib.rphip_predict_merge_topk(ctx.handle, L, Kp, U, C.c_void_p(cand_idx_t.data_ptr()), C.c_void_p(cand_val_t.data_ptr()),
                                           C.c_void_p(cand_nnz_t.data_ptr()), C.c_void_p(idx_t.data_ptr()), C.c_void_p(val_t.data_ptr()),
                                           C.c_void_p(nnz_t.data_ptr())))


def predict_topk(ctx: Context, model: Model, X: DeviceCSR, Kp: int, exclude_history: bool = False, user_begin: int = 0,
                 user_end: int | None = None, item_keep: np.ndarray | None = None):
    """item_keep: optional bool/uint8[n_items]; only items with a non-zero entry can be recommended
    (ExcludeItems / SelectItems of recpack/postprocessing/filters.py:58-101, fused)."""
    user_end = X.shape[0] if user_end is None else user_end
    n = user_end - user_begin
    idx = np.empty((n, Kp), dtype=np.int32)
    val = np.empty((n, Kp), dtype=np.float32)
    nnz = np.empty(n, dtype=np.int32)
    keep_p = None
    if item_keep is not None:
        item_keep = np.ascontiguousarray(np.asarray(item_keep) != 0, dtype=np.uint8)
        if item_keep.shape != (X.shape[1],):
            raise ValueError(f"item_keep has shape {item_keep.shape}, expected ({X.shape[1]},)")
        keep_p = _np_ptr(item_keep)
    check(ctx.lib.rphip_predict_topk_filtered(ctx.handle, model.handle, X.handle, int(user_begin), int(user_end), int(Kp),
                                              1 if exclude_history else 0, keep_p, _np_ptr(idx), _np_ptr(val), _np_ptr(nnz)))
    return idx, val, nnz


def predict_topk_dev(ctx: Context, model: Model, X: DeviceCSR, Kp: int, idx_t, val_t, nnz_t, exclude_history: bool = False,
                     user_begin: int = 0, user_end: int | None = None, item_keep_t=None):
    """Outputs into torch CUDA tensors idx int32[n,Kp], val float32[n,Kp], nnz int32[n]; asynchronous.
    item_keep_t: optional torch uint8 CUDA tensor [n_items]."""
    user_end = X.shape[0] if user_end is None else user_end
    _torch_inputs_ready(ctx)  # (also orders any pending torch fill of the OUTPUT tensors before the library's own writes)
    check(ctx.lib.rphip_predict_topk_filtered_dev(ctx.handle, model.handle, X.handle, int(user_begin), int(user_end), int(Kp),
                                                  1 if exclude_history else 0,
                                                  C.c_void_p(item_keep_t.data_ptr()) if item_keep_t is not None else None,
                                                  C.c_void_p(idx_t.data_ptr()), C.c_void_p(val_t.data_ptr()), C.c_void_p(nnz_t.data_ptr())))


def predict_full(ctx: Context, model: Model, X: DeviceCSR, user_begin: int = 0, user_end: int | None = None):
    """Every non-zero score as CSR arrays (indptr int64, indices int32, data float64)."""
    user_end = X.shape[0] if user_end is None else user_end
    n = user_end - user_begin
    row_nnz = np.zeros(n, dtype=np.int64)
    total = C.c_int64()
    check(ctx.lib.rphip_predict_full_count(ctx.handle, model.handle, X.handle, int(user_begin), int(user_end), _np_ptr(row_nnz), C.byref(total)))
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(row_nnz, out=indptr[1:])
    indices = np.empty(total.value, dtype=np.int32)
    data = np.empty(total.value, dtype=np.float64)
    check(ctx.lib.rphip_predict_full_fill(ctx.handle, model.handle, X.handle, int(user_begin), int(user_end), _np_ptr(indptr),
                                          _np_ptr(indices), _np_ptr(data)))
    return indptr, indices, data
