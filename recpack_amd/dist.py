"""Sharding of the item-similarity path over the GPUs of one node (one process per GPU, torch.distributed).

SURVEY.md §8(e): the rows of the similarity matrix are independent, so ItemKNN fit shards by ITEM ROWS
(X and X^T replicated, contiguous row ranges balanced by cost) and needs one exchange: an all-gather of the
per-shard neighbour lists (I*K*12 bytes in total) so that every GPU holds S for predict.  Predict shards by
USERS with no collective.  The reference has no distributed code at all (recpack/algorithms/base.py:454-455
leaves multi-GPU as a TODO); backend "nccl" is RCCL over xGMI on MI355X, "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def partition_by_cost(cost: np.ndarray, n_parts: int) -> List[Tuple[int, int]]:
    """Contiguous ranges [b, e) of near-equal total cost; every part is a valid (possibly empty) range."""
    n = int(cost.shape[0])
    if n_parts <= 1:
        return [(0, n)]
    c = np.cumsum(cost.astype(np.float64))
    total = float(c[-1]) if n else 0.0
    bounds = [0]
    for p in range(1, n_parts):
        b = int(np.searchsorted(c, total * p / n_parts, side="left")) + 1 if total > 0 else (n * p) // n_parts
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[i], bounds[i + 1]) for i in range(n_parts)]


def item_row_cost(item_deg: np.ndarray, nnz: int, n_products: int) -> np.ndarray:
    """Cost model of one Gram row: c_i * (P / nnz) scalar products plus a fixed per-row share for the top-K."""
    n_items = max(1, item_deg.shape[0])
    per_user = (n_products / nnz) if nnz else 1.0
    return item_deg.astype(np.float64) * per_user + 0.5 * (n_products / n_items if n_products else 1.0)


def partition_users(indptr_or_deg: np.ndarray, n_parts: int, is_indptr: bool = True) -> List[Tuple[int, int]]:
    """User ranges balanced by history length (the predict cost is K * d_u)."""
    d = np.diff(indptr_or_deg) if is_indptr else indptr_or_deg
    return partition_by_cost(d.astype(np.float64) + 1.0, n_parts)


def allgather_topk(idx_local, val_local, nnz_local, parts: Sequence[Tuple[int, int]], group=None):
    """All-gather the neighbour lists of every rank's row range into full [I, K] tensors (any torch device).

    ``idx_local`` int32[rows, K], ``val_local`` float64[rows, K], ``nnz_local`` int32[rows] hold this rank's
    rows ``parts[rank]``.  Shards are padded to the longest range so that ONE all_gather per array suffices
    (direct all-gather: every GPU talks to all peers at once over xGMI; no ring of broadcasts).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(parts) == world
    K = int(idx_local.shape[1])
    n_items = parts[-1][1]
    max_rows = max(e - b for b, e in parts)
    b, e = parts[rank]
    assert idx_local.shape[0] == e - b

    def padded(t, fill):
        out = torch.full((max_rows,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
        out[: t.shape[0]] = t
        return out

    outs = []
    for t, fill in ((idx_local, -1), (val_local, 0), (nnz_local, 0)):
        p = padded(t, fill)
        g = torch.empty((world * max_rows,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
        dist.all_gather_into_tensor(g, p, group=group)  # concatenation along dim 0 (works for nccl and gloo)
        g = g.view((world, max_rows) + tuple(p.shape[1:]))
        full = torch.empty((n_items,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
        for r, (rb, re) in enumerate(parts):
            full[rb:re] = g[r, : re - rb]
        outs.append(full)
    return tuple(outs)


class _DevPtr:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, shape, typestr: str, owner=None):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        self._owner = owner


def model_topk_tensors(model, device):
    """torch views (no copy) of a neighbour-list model's device arrays: idx, val (float64), nnz."""
    import torch

    pi, pv, pn, rows, K = model.topk_device_ptrs()
    if rows == 0:
        return (torch.empty((0, K), dtype=torch.int32, device=device), torch.empty((0, K), dtype=torch.float64, device=device),
                torch.empty((0,), dtype=torch.int32, device=device))
    idx = torch.as_tensor(_DevPtr(pi, (rows, K), "<i4", model), device=device)
    val = torch.as_tensor(_DevPtr(pv, (rows, K), "<f8", model), device=device)
    nnz = torch.as_tensor(_DevPtr(pn, (rows,), "<i4", model), device=device)
    return idx, val, nnz


def model_dense_tensor(model, device):
    """torch view (no copy) of a dense model's B: float32 [n_items, n_items] with the library's row stride."""
    import torch

    ptr, n, ld = model.dense_device_ptr()
    full = torch.as_tensor(_DevPtr(ptr, (n, ld), "<f4", model), device=device)
    return full[:, :n]


def csr_for_shard(ctx, indptr_t, indices_t, shape, rows: Tuple[int, int], device, group=None, comm=None):
    """K0 for one shard of a multi-GPU fit with the degree pass sharded: this rank histograms (and ranks) only its own item rows
    ``rows``, the int32 degree vector (4 * n_items bytes) is summed over the ranks with ONE all-reduce (RCCL over xGMI when the
    backend is nccl - BASELINE.json's north_star), then the reciprocal degrees are formed.  X^T is filled for the shard's rows."""
    import torch
    import torch.distributed as dist

    from recpack_amd import _lib

    if comm is not None and comm.world == 1 or comm is None and (not dist.is_initialized() or dist.get_world_size(group) == 1):
        return _lib.DeviceCSR.from_device(ctx, indptr_t, indices_t, shape, fit_rows=rows)
    dX = _lib.DeviceCSR.from_device_shard(ctx, indptr_t, indices_t, shape, rows)
    if comm is not None:  # the library's communicator, on the library's stream: no host synchronisation
        comm.all_reduce_sum_i32(dX.degrees_device_ptr(), int(shape[1]))
    else:
        deg = torch.as_tensor(_DevPtr(dX.degrees_device_ptr(), (int(shape[1]),), "<i4", dX), device=device)
        ctx.sync()  # the library's stream has written the partial degrees
        dist.all_reduce(deg, op=dist.ReduceOp.SUM, group=group)
    dX.finish_degrees()
    return dX


def fit_itemknn_sharded(ctx, dX, K: int, parts: Sequence[Tuple[int, int]], device, group=None, fit_fn=None, comm=None, **fit_kwargs):
    """Fit this rank's item rows on its GPU, all-gather, return a predict-ready full model (replicated).
    ``fit_kwargs`` (similarity, pop_discount, normalize_X) go to ``_lib.itemknn_fit``; ``fit_fn(row_begin, row_end)``
    replaces it for the other neighbour-list fits (``itemknn_fit_values`` / ``_timedist`` / ``_hermann`` of the
    time-aware family: item rows are independent there as well)."""
    import torch
    import torch.distributed as dist

    from recpack_amd import _lib

    rank = comm.rank if comm is not None else dist.get_rank(group)
    b, e = parts[rank]
    local = fit_fn(b, e) if fit_fn is not None else _lib.itemknn_fit(ctx, dX, K, b, e, **fit_kwargs)
    if comm is not None:  # one all-gather inside the library, straight into the full model's arrays (rphip_itemknn_gather)
        return _lib.itemknn_gather(ctx, comm, local, [p[0] for p in parts] + [parts[-1][1]])
    idx, val, nnz = model_topk_tensors(local, device)
    ctx.sync()  # the collective runs on torch's stream: order it after the library's stream
    idx_f, val_f, nnz_f = allgather_topk(idx, val, nnz, parts, group)
    torch.cuda.current_stream(device).synchronize()
    return _lib.Model.from_topk_device(ctx, idx_f, val_f, nnz_f)


def make_comm(ctx, device, group=None, transport: str = "auto"):
    """The communicator the library's distributed entry points run on (``_lib.Comm``).  ``transport``: "rccl" = the library's
    own RCCL communicator (one rank per GPU, RCCL over xGMI; the default whenever the group's backend is nccl), "torch" =
    callbacks through the torch.distributed group (any backend - what the gloo tests use), "auto" picks by the backend."""
    import torch.distributed as dist

    from recpack_amd import _lib

    if transport == "auto":  # (outside any process group: a one-rank communicator whose operations are never called)
        transport = "rccl" if (dist.is_initialized() and dist.get_backend(group) == "nccl") else "torch"
    if transport == "rccl":
        return _lib.Comm.rccl(ctx, group)
    if transport == "torch":
        return _lib.Comm.torch(ctx, device, group)
    raise ValueError(f"unknown transport {transport!r}")


def ease_fit_sharded(ctx, dX, l2: float, alpha: float, device, group=None, panel_tiles: int = 0, comm=None):
    """EASE fit (recpack/algorithms/ease.py:63-95) with the n x n matrix SHARDED IN MEMORY over the ranks of ``group``, one GPU
    each - the whole solve inside the library (rphip_ease_fit_dist): 1-D block-cyclic column panels,
    symmetric block Gauss-Jordan sweeps on the lower triangle (n^3 / world flop per rank - the count of potrf + trtri + lauum -
    every pivot block inverted by the blocked MFMA Cholesky), the pivot column gathered / broadcast on a communication stream
    while the previous update still runs (look-ahead of one panel), no host read-back before the end, the upper triangle filled
    by one personalised exchange per ``world`` panels, K7 on the panels.  Collectives: the library's own RCCL communicator
    (``make_comm``; callbacks through torch.distributed when the backend is not nccl).  No torch tensor is allocated.
    Returns (model with this rank's columns of B - ``model.local_columns()`` - or None when the rank owns only padding,
    layout dict of ``Model.dense_layout`` or None)."""
    from recpack_amd import _lib

    own = comm is None
    if own:
        comm = make_comm(ctx, device, group)
    try:
        m = _lib.ease_fit_dist(ctx, comm, dX, l2, alpha, panel_tiles)
    finally:
        if own:
            comm.close()
    return m, (m.dense_layout() if m is not None else None)


def ease_gather_model(ctx, model, n_items: int, comm):
    """Every rank's column panels of ``ease_fit_sharded`` into ONE full model on this rank (rphip_ease_model_allgather: strips of
    rows, one all-gather each).  While 3 n^2 floats fit a GPU this is the faster multi-GPU predict: every rank then runs the
    single-GPU fused predict on ITS USERS (``_lib.predict_topk_dev(..., user_begin, user_end)``) - full-width rows of B, no per-rank
    candidate lists rescored world times over, no exchange of lists; ``predict_topk_sharded`` stays for matrices that do not fit.
    ``model`` is None on a rank that owns only padding (it takes part all the same)."""
    from recpack_amd import _lib

    panel = _lib.ease_fit_dist_stats(ctx)["panel"]  # (of the fit that just ran on this context: known on every rank)
    return _lib.ease_model_allgather(ctx, comm, model, int(n_items), int(panel))


def ease_gather_rows_i8(ctx, model, n_items: int, comm):
    """Only the int8 rows of the gathered model on every rank (rphip_ease_model_allgather_i8: n^2 bytes instead of 5 n^2) - the
    model :func:`predict_topk_owner` runs its first pass on.  ``model`` is None on a rank that owns only padding."""
    from recpack_amd import _lib

    panel = _lib.ease_fit_dist_stats(ctx)["panel"]
    return _lib.ease_model_allgather_i8(ctx, comm, model, int(n_items), int(panel))


def predict_topk_owner(ctx, local, rows_i8, dX, Kp: int, user_parts: Sequence[Tuple[int, int]], device, comm, exclude_history: bool = False):
    """Top-Kp of this rank's users with the exact scores formed by the owners of the columns (rphip_predict_topk_owner; round 5):
    no rank holds the f32 matrix - the first pass reads the gathered int8 rows, the candidates are all-gathered, every rank sums the
    entries of the candidates in ITS columns for all users, one exchange returns the sums.  ``user_parts``: the ranks' contiguous user
    ranges (``partition_users``).  Returns (idx int32 [m, Kp], val float32 [m, Kp], nnz int32 [m]) for this rank's users - the
    single-GPU result bit for bit.  Raises ``_lib.RecpackHipError`` (code RPHIP_ELIMIT / RPHIP_ENOMEM, on every rank alike) when the
    route does not apply: take :func:`predict_topk_sharded` then."""
    import torch

    from recpack_amd import _lib

    bounds = [int(p[0]) for p in user_parts] + [int(user_parts[-1][1])]
    ub, ue = user_parts[comm.rank]
    oi = torch.empty((max(ue - ub, 1), Kp), dtype=torch.int32, device=device)
    ov = torch.empty((max(ue - ub, 1), Kp), dtype=torch.float32, device=device)
    on = torch.empty((max(ue - ub, 1),), dtype=torch.int32, device=device)
    _lib.predict_topk_owner(ctx, comm, local, rows_i8, dX, bounds, Kp, oi, ov, on, exclude_history=exclude_history)
    return oi[: ue - ub], ov[: ue - ub], on[: ue - ub]


def ease_gather_fits(n_items: int, panel: int, hbm_bytes: int, world: int = 1, free_bytes: int | None = None) -> bool:
    """Does the gathered model fit beside a rank's own panels?  Needed: B (4 n^2 bytes), its quantised copy for the first pass of
    K9 (n^2 bytes as int8, 2 n^2 as fp16: the larger is assumed), 2 GiB of staging strips - while the rank's own panels
    (4 n^2 / world) are still alive.  `free_bytes` = what the device can still hand out (hipMemGetInfo's free + the context's cached
    blocks); without it 80 % of the device less the rank's panels is assumed.  A wrong "yes" is not fatal: the ranks of
    rphip_ease_model_allgather agree on their allocations and fail together with RPHIP_ENOMEM."""
    n_pad = -(-int(n_items) // max(int(panel), 128)) * max(int(panel), 128)
    need = 6 * n_pad * n_pad + (2 << 30)
    if free_bytes is None:
        free_bytes = int(0.8 * hbm_bytes) - 4 * n_pad * n_pad // max(int(world), 1)
    return need < free_bytes


def predict_topk_sharded(ctx, model, dX, Kp: int, device, exclude_history: bool = False, group=None, user_chunk: int = 1 << 18, comm=None):
    """Top-Kp for every user with B sharded by item columns, inside the library
    (rphip_predict_topk_dist): every rank scores ALL users against its own columns (`model` = ease_fit_sharded's model, None
    on a rank that owns only padding) and keeps a local top-Kp; the lists travel to the rank that owns the user - one personalised
    exchange per chunk through the communicator, U * Kp * 8 bytes per rank in all - and are merged there by (score desc, column
    asc), the single-GPU order.  Returns this rank's part: (user_ids int64 [m], idx int32 [m, Kp], val float32 [m, Kp],
    nnz int32 [m]) as torch tensors; rank r owns the users [r * per, (r + 1) * per), per = ceil(U / world).
    (``user_chunk`` is kept for callers of the earlier torch-driven version; the library picks its own chunking.)"""
    import torch

    from recpack_amd import _lib

    own = comm is None
    if own:
        comm = make_comm(ctx, device, group)
    try:
        U = dX.shape[0]
        per = -(-max(U, 1) // comm.world)
        oi = torch.empty((per, Kp), dtype=torch.int32, device=device)
        ov = torch.empty((per, Kp), dtype=torch.float32, device=device)
        on = torch.empty((per,), dtype=torch.int32, device=device)
        ub, nu = _lib.predict_topk_dist(ctx, comm, model, dX, Kp, oi, ov, on, exclude_history=exclude_history)
        ctx.sync()
    finally:
        if own:
            comm.close()
    return torch.arange(ub, ub + nu, device=device), oi[:nu], ov[:nu], on[:nu]


