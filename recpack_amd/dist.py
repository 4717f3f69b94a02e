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


def ease_fit_sweeps(ctx, dX, l2: float, alpha: float, device, group=None, panel_tiles=None):
    """Round 2's column-sharded route, kept as the fallback: contiguous column slabs in torch tensors, FULL block sweeps driven
    from Python between torch.distributed collectives (2 n^3 / world flop, one blocking broadcast and one host read-back per
    pivot).  ``ease_fit_sharded`` is the route to use.

    EASE fit (recpack/algorithms/ease.py:63-95) with the n x n matrix SHARDED BY ITEM COLUMNS over the ranks of ``group``,
    one GPU each.  Rank r owns the columns [r w, (r + 1) w) of everything, w = n_pad / world: a slab
    of n_pad x w floats - G + l2 I first, -(G + l2 I)^-1 after the solve, its columns of B at the end.  Peak memory per rank:
    the slab (n^2 * 4 / world bytes) + two panels of n_pad x (128 panel_tiles) + a Gram staging buffer of the same order.

    The inverse is formed IN PLACE by block sweeps (Goodnight 1979; the symmetric Gauss-Jordan form of np.linalg.inv, ease.py:82).
    For the pivot block K (128 panel_tiles columns, owned by one rank) with P = A_KK^-1 and T = A[:, K] P:
        A[:, J] -= T A[K, J]  for every other column block J,   A[K, J] = T[J, :]^T,   A[:, K] = T,   A[K, K] = -P,
    and after the last block the matrix is -(G + l2 I)^-1.  Per block: the owner inverts the pivot block (blocked Cholesky on
    MFMA tiles, rphip_spd_inverse_f32) and broadcasts its panel column with P appended (n_pad * nb * 4 bytes - the only
    communication, n^2 * 4 bytes over the whole solve); every rank forms T (a skinny product, repeated on every rank: 2 n nb^2
    flop) and updates its slab with ONE plain product A[:, J] -= T A[J, K]^T (A is symmetric, so the rows J of the broadcast
    panel are A[K, J]^T) on the same MFMA kernel as the single-GPU solve.  The update touches both triangles (2 n^3 / world flop
    per rank against n^3 for the single-GPU Cholesky route): what that buys is no distributed triangular phases, no mirror
    exchange, a layout that never changes - and the columns a rank ends up with are exactly what the column-sharded predict
    (predict_topk_sharded) needs.  Accuracy: the same as the Cholesky route (tests: 1e-4 bar against float64).
    ``panel_tiles`` (128-column tiles per pivot block): by default 8 .. 32, as many as keep the two panel buffers within ~8 % of
    n_pad^2 floats - wide blocks amortise the read-modify-write of the slab (measured on one MI355X, 50 000 items: 8 / 16 / 32
    tiles = 1.33 / 1.18 / 1.11 s; 100 000 items: 8.8 s with 8, 7.4 s with 16).
    Returns (model on this rank's slab or None when the rank owns only padding, (col_begin, n_cols)).
    """
    import torch
    import torch.distributed as dist

    from recpack_amd import _lib

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = 128
    n = dX.shape[1]
    n_pad = -(-n // (T * world)) * (T * world)
    w = n_pad // world
    c0 = rank * w
    lib_stream_is_torchs = ctx.stream is not None and int(torch.cuda.current_stream().cuda_stream) == ctx.stream

    def to_torch():  # library work -> visible to torch's stream
        if not lib_stream_is_torchs:
            ctx.sync()

    def src_rank(r):
        return dist.get_global_rank(group, r) if group is not None else r

    S = torch.empty((n_pad, w), dtype=torch.float32, device=device)
    # ---- G + l2 I, this rank's columns: Gram ROWS [c0, c0 + w) in chunks (K4), transposed into the slab (G is symmetric)
    ch_rows = min(w, max(T, min(8 * T, int(0.08 * n_pad) // T * T)))  # the staging buffer stays within ~8 % of n_pad^2 floats
    stage = torch.empty((ch_rows, n_pad), dtype=torch.float32, device=device)
    for q in range(0, w, ch_rows):
        ch = min(ch_rows, w - q)
        _lib.ease_gram_rows_into(ctx, dX, c0 + q, c0 + q + ch, l2, stage, n_pad)
        _lib.transpose(ctx, stage[:ch], S[:, q:q + ch])
    to_torch()
    del stage
    # ---- the sweeps
    if panel_tiles is None:
        panel_tiles = max(8, min(32, int(0.04 * n_pad) // T))
    pb_cols = min(w, max(1, panel_tiles) * T)
    pbuf = torch.empty(((n_pad + pb_cols) * pb_cols,), dtype=torch.float32, device=device)
    tbuf = torch.empty((n_pad * pb_cols,), dtype=torch.float32, device=device)
    bad = None
    for owner in range(world):
        for q in range(0, w, pb_cols):
            nb = min(pb_cols, w - q)
            k0 = owner * w + q
            k1 = k0 + nb
            Pv = pbuf[: (n_pad + nb) * nb].view(n_pad + nb, nb)  # rows [0, n_pad): the panel column; then P
            if rank == owner:
                Pv[:n_pad].copy_(S[:, q:q + nb])
                Pv[n_pad:].copy_(Pv[k0:k1])
                try:
                    _lib.spd_inverse(ctx, Pv[n_pad:])
                except _lib.RecpackHipError as e:  # not positive definite: tell everyone through the panel
                    bad = e
                    Pv[n_pad:].fill_(float("nan"))
                to_torch()
            if dist.is_initialized():
                dist.broadcast(Pv, src=src_rank(owner), group=group)
            if bool(torch.isnan(Pv[n_pad, 0])):
                raise bad if bad is not None else _lib.RecpackHipError(-8, f"pivot block at column {k0} is not positive definite (reported by rank {owner})")
            Tv = tbuf[: n_pad * nb].view(n_pad, nb)
            _lib.gemm_nt(ctx, Pv[:n_pad], Pv[n_pad:], Tv)                       # T = A[:, K] P   (P symmetric)
            _lib.gemm_nt(ctx, Tv, Pv[c0:c0 + w], S, alpha=-1.0, beta=1.0)       # A[:, J] -= T A[J, K]^T
            _lib.transpose(ctx, Tv[c0:c0 + w], S[k0:k1])                        # A[K, J] = T[J, :]^T
            if rank == owner:
                to_torch()
                S[:, q:q + nb].copy_(Tv)                                        # A[:, K] = T
                S[k0:k1, q:q + nb].copy_(Pv[n_pad:]).neg_()                     # A[K, K] = -P
            to_torch()
    del pbuf, tbuf
    # ---- K7 on the slab: B_ij = -P_ij / P_jj * w_j, B_jj = 0 (ease.py:85-90); the slab holds -P, the sign cancels
    _lib.ease_scale_cols(ctx, dX, S, c0, alpha)
    to_torch()
    n_cols = max(0, min(w, n - c0))
    if n_cols == 0:
        return None, (c0, 0)
    return _lib.Model.from_dense_cols(ctx, n, c0, n_cols, S), (c0, n_cols)


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


def ease_fit_replicated(ctx, dX, l2: float, alpha: float, device, group=None, panel_tiles: int = 16):
    """EASE fit on all ranks with the matrix REPLICATED (the faster route while n^2 * 4 bytes fit every GPU: n^3 / world flop per
    rank through the Cholesky phases, against 2 n^3 / world for the column-sharded sweeps of ease_fit_sharded).

    Every rank keeps ONE n x n float32 matrix: first G + l2*I, then the Cholesky factor L below the diagonal and
    N = L^-T on and above it (a diagonal block is only needed until its panel is done, then it holds N's block), finally B.
    The O(n^3) work is done in wide panels (``panel_tiles`` x 128 columns): a panel takes the update of everything to its
    left in one product whose rows are split over the ranks, and the finished panel is all-gathered (n^2/2 * 4 bytes in
    total per phase, ~n / 2048 large collectives plus one small one per 128 columns inside a panel);
    P = N N^T (lower triangle, row slabs of equal area) is exchanged slab by slab (columns up to the diagonal only: n^2/2 * 4 bytes),
    mirrored and scaled; B is
    all-gathered into the big matrix, so predict can shard by users as for ItemKNN.  Peak memory n^2 * 4 * (1 + 1/world)
    bytes per GPU.  Returns a dense model on this rank's copy of B.
    """
    import torch
    import torch.distributed as dist

    from recpack_amd import _lib

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = 128
    n = dX.shape[1]
    n_pad = -(-n // (T * world)) * (T * world)  # rows split evenly in whole tiles
    rows = n_pad // world
    nb = n_pad // T
    lib_stream_is_torchs = ctx.stream is not None and int(torch.cuda.current_stream().cuda_stream) == ctx.stream

    def to_torch():  # library work -> visible to torch's stream
        if not lib_stream_is_torchs:
            ctx.sync()

    def gather(out, inp):
        to_torch()
        if dist.is_initialized():
            dist.all_gather_into_tensor(out, inp, group=group)
        else:
            out.view(-1)[: inp.numel()].copy_(inp.reshape(-1))

    A = torch.empty((n_pad, n_pad), dtype=torch.float32, device=device)
    W = torch.empty((nb, T, T), dtype=torch.float32, device=device)
    info = torch.zeros((1,), dtype=torch.int32, device=device)
    # ---- Gram rows of this rank, all-gathered
    mine = torch.empty((rows, n_pad), dtype=torch.float32, device=device)
    _lib.ease_gram_rows(ctx, dX, rank * rows, (rank + 1) * rows, l2, A)
    to_torch()
    mine.copy_(A[rank * rows:(rank + 1) * rows])
    gather(A, mine)
    PW = min(nb, panel_tiles)  # tiles per wide panel
    max_per = -(-nb // world)
    send = torch.empty((max_per * T, PW * T), dtype=torch.float32, device=device)
    recv = torch.empty((world * max_per * T, PW * T), dtype=torch.float32, device=device)
    tmp = torch.empty((max_per * T, PW * T), dtype=torch.float32, device=device)

    def split(first_tile, n_tiles):
        """This rank's share [lo, hi) of the tiles first_tile .. first_tile + n_tiles, and the per-rank chunk size."""
        per = -(-n_tiles // world)
        lo = min(n_tiles, rank * per)
        hi = min(n_tiles, (rank + 1) * per)
        return first_tile + lo, first_tile + hi, per

    def gather_rows(dst, n_rows, cols, per, mine_rows):
        """All-gather the ranks' row shares (mine_rows x cols, in send) and store the first n_rows rows into dst."""
        sv = send.view(-1)[: per * T * cols].view(per * T, cols)
        rv = recv.view(-1)[: world * per * T * cols].view(world * per * T, cols)
        del mine_rows
        gather(rv, sv)
        dst.copy_(rv[:n_rows])

    def share(cols, n_rows_mine):
        return send.view(-1)[: max(1, n_rows_mine) * cols].view(max(1, n_rows_mine), cols)[:n_rows_mine]

    # ---- K5: Cholesky in two levels, left-looking across wide panels.  Per panel: its columns take the update of
    # everything to their left in ONE wide product (rows split over the ranks, one all-gather of the panel), then the panel is
    # factored 128 columns at a time with K restricted to the panel (short products, rows split, small all-gathers).
    for pb in range(0, nb, PW):
        pe = min(nb, pb + PW)
        j, w = pb * T, (pe - pb) * T
        if j > 0:
            lo, hi, per = split(pb, nb - pb)
            mine_rows = (hi - lo) * T
            if hi > lo:
                R = slice(lo * T, hi * T)
                out = share(w, mine_rows)
                out.copy_(A[R, j:j + w])
                _lib.gemm_nt_ex(ctx, A[R], A[j:j + w], out, 0, j, alpha=-1.0, beta=1.0)  # A[R, panel] -= L[R, :j] L[panel, :j]^T
            gather_rows(A[j:, j:j + w], (nb - pb) * T, w, per, mine_rows)
            if not lib_stream_is_torchs:
                torch.cuda.current_stream().synchronize()
        for k in range(pb, pe):
            r0 = k * T
            diag = A[r0:r0 + T, r0:r0 + T]
            if r0 > j:  # every rank updates the diagonal tile itself (128 rows: negligible, and no broadcast)
                _lib.gemm_nt_ex(ctx, A[r0:r0 + T], A[r0:r0 + T], diag, j, r0, alpha=-1.0, beta=1.0)
            _lib.potrf_trtri_128(ctx, diag, W[k], info)
            lo, hi, per = split(k + 1, nb - k - 1)
            mine_rows = (hi - lo) * T
            if hi > lo:
                R = slice(lo * T, hi * T)
                if r0 > j:
                    _lib.gemm_nt_ex(ctx, A[R], A[r0:r0 + T], A[R, r0:r0 + T], j, r0, alpha=-1.0, beta=1.0)
                _lib.gemm_nt_ex(ctx, A[R, r0:r0 + T], W[k], share(T, mine_rows), 0, T)  # L[R, k] = A[R, k] W_k^T
            if nb - k - 1 > 0:
                gather_rows(A[r0 + T:, r0:r0 + T], (nb - k - 1) * T, T, per, mine_rows)
            else:
                to_torch()
            diag.copy_(W[k].t())  # from here on the block is N's diagonal block
            if not lib_stream_is_torchs:
                torch.cuda.current_stream().synchronize()
    # ---- K6a: N = L^-T above the diagonal, panel by panel: the panel's own triangle on every rank (tiny products, no
    # exchange), then T = N11 L21^T and N12 = -T N22 as two wide products with the rows split, one all-gather per panel
    for pb in range(0, nb, PW):
        pe = min(nb, pb + PW)
        j, w = pb * T, (pe - pb) * T
        for i in range(pb + 1, pe):
            c0 = i * T
            t1 = tmp.view(-1)[: (c0 - j) * T].view(c0 - j, T)
            _lib.gemm_nt_ex(ctx, A[j:c0], A[c0:c0 + T], t1, j, c0, k_from_row=True, m_off=j)
            _lib.gemm_nt_ex(ctx, t1, W[i], A[j:c0, c0:c0 + T], 0, T, alpha=-1.0)
        if j > 0:
            to_torch()
            NT = torch.triu(A[j:j + w, j:j + w]).t().contiguous()  # N22^T; the tiles below the diagonal hold L: zero
            lo, hi, per = split(0, pb)
            mine_rows = (hi - lo) * T
            if hi > lo:
                R = slice(lo * T, hi * T)
                t2 = tmp.view(-1)[: mine_rows * w].view(mine_rows, w)
                _lib.gemm_nt_ex(ctx, A[R], A[j:j + w], t2, 0, j, k_from_row=True, m_off=lo * T)
                _lib.gemm_nt_ex(ctx, t2, NT, share(w, mine_rows), 0, w, alpha=-1.0)
            gather_rows(A[:j, j:j + w], pb * T, w, per, mine_rows)
            if not lib_stream_is_torchs:
                torch.cuda.current_stream().synchronize()
            del NT
    # ---- K6b: P = N N^T, lower triangle only.  The row tiles are cut into `world` slabs of equal AREA under the diagonal
    # (slab r ends at n * sqrt((r + 1) / world)); a rank computes the lower tiles of its slab in chunks (a chunk goes to a
    # staging buffer and then over the L part of its rows - N on and above the diagonal is still read by the chunks after
    # it), the slabs are exchanged with one broadcast each, the lower half is mirrored and K7 runs on the whole matrix.
    import math

    cut = [min(nb, int(round(nb * math.sqrt(r / world)))) for r in range(world)] + [nb]
    for r in range(1, world + 1):
        cut[r] = max(cut[r], cut[r - 1])
    t_lo, t_hi = cut[rank], cut[rank + 1]
    chunk_tiles = max(1, min(max_per, 64))
    stage = torch.empty((chunk_tiles * T, n_pad), dtype=torch.float32, device=device)
    done = []
    for c_lo in range(t_lo, t_hi, chunk_tiles):
        c_hi = min(t_hi, c_lo + chunk_tiles)
        r0, r1 = c_lo * T, c_hi * T
        Pc = stage[: r1 - r0, :r1]
        _lib.gemm_nt_ex(ctx, A[r0:r1], A[:r1], Pc, 0, n_pad, k_from_row=True, k_from_col=True, m_off=r0, lower_only=True)
        done.append((r0, r1))
        to_torch()
        A[r0:r1, :r1].copy_(Pc)  # (the tiles above the diagonal inside the chunk's square are overwritten by the mirror)
    if not lib_stream_is_torchs:
        torch.cuda.current_stream().synchronize()
    if dist.is_initialized():
        dist.barrier(group=group)  # every rank has finished reading N
        # slab r: only its columns up to the diagonal (what was computed; the rest of those rows still holds N and is overwritten
        # by the mirror below, which must therefore run before anything reads the upper tiles) - packed chunk by chunk through
        # the staging buffer, since a collective wants contiguous memory: n^2 / 2 * 4 bytes in total
        flat = stage.view(-1)
        for r in range(world):
            src = dist.get_global_rank(group, r) if group is not None else r
            cols = cut[r + 1] * T
            for c_lo in range(cut[r], cut[r + 1], chunk_tiles):
                c_hi = min(cut[r + 1], c_lo + chunk_tiles)
                r0, r1 = c_lo * T, c_hi * T
                buf = flat[: (r1 - r0) * cols].view(r1 - r0, cols)
                if rank == r:
                    buf.copy_(A[r0:r1, :cols])
                dist.broadcast(buf, src=src, group=group)
                if rank != r:
                    A[r0:r1, :cols].copy_(buf)
    MB = 32 * T  # mirror the lower half over the upper one, a block row at a time (no second matrix)
    for r0 in range(0, n_pad, MB):
        r1 = min(n_pad, r0 + MB)
        blk = A[r0:r1, r0:r1]
        blk.copy_(torch.tril(blk) + torch.tril(blk, -1).t())
        if r1 < n_pad:
            A[r0:r1, r1:].copy_(A[r1:, r0:r1].t())
    # ---- K7: B_ij = -P_ij / P_jj * w_j, B_ii = 0 (ease.py:85-90), padding zeroed; every rank on its own copy
    d_all = torch.diagonal(A).clone()
    w = torch.ones((n_pad,), dtype=torch.float32, device=device)
    if alpha != 0.0:
        deg = torch.as_tensor(dX.item_degrees(), device=device).to(torch.float32)
        w[:n] = torch.where(deg > 0, deg.clamp(min=1).pow(-float(alpha)), torch.zeros_like(deg))
    scale = -w / d_all
    for r0 in range(0, n_pad, MB):  # (in block rows: no n x n temporary)
        A[r0:r0 + MB].mul_(scale[None, :])
    torch.diagonal(A).zero_()
    A[:, n:] = 0
    A[n:] = 0
    del stage
    if not lib_stream_is_torchs:
        torch.cuda.current_stream().synchronize()
    bad = int(info.item())
    if bad:
        raise _lib.RecpackHipError(-7, f"Cholesky broke down (non-positive pivot {bad} in a diagonal block): G + l2*I is not positive definite")
    return _lib.Model.from_dense_device(ctx, n, A)
