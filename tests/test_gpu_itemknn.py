"""GPU parity: ItemKNN cosine fit + predict through the C ABI against the golden vectors and the oracle.

Bars: neighbour index sets bit-exact against the oracle (integer work, same documented tie order) and
tie-aware identical against the reference's goldens (eps = 1e-9); similarity values
rtol 1e-12 (float64, closed form c_ij/sqrt(c_i c_j)); predicted scores rtol 2e-7 (f32 similarities, exact fixed-point sums).
"""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from, load_golden
from oracle import oracle_np as onp
from parity import tie_aware_topk, topk_vals_close

pytestmark = pytest.mark.gpu

VAL_RTOL = 1e-12
PRED_RTOL = 2e-7  # similarities are stored in f32 (6e-8 each), the sums are exact, top-Kp values round to f32 once more


@pytest.fixture(scope="module")
def hip():
    from recpack_amd import _lib, get_context

    ctx = get_context(0)
    return _lib, ctx


def fit_arrays(hip, X, K, rows=None):
    _lib, ctx = hip
    X = sp.csr_matrix(X)
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    r0, r1 = (0, X.shape[1]) if rows is None else rows
    m = _lib.itemknn_fit(ctx, dX, K, r0, r1)
    idx, val, nnz = m.topk_download()
    return dX, m, idx, val, nnz


def assert_same_as_oracle(idx, val, nnz, ref):
    k = min(idx.shape[1], ref["idx"].shape[1])
    np.testing.assert_array_equal(nnz, ref["nnz"])
    # the device emits a row in column order, the oracle in rank order
    for r in range(idx.shape[0]):
        n = nnz[r]
        o = np.argsort(ref["idx"][r, :n])
        np.testing.assert_array_equal(idx[r, :n], ref["idx"][r, :n][o], err_msg=f"row {r}")
        np.testing.assert_allclose(val[r, :n], ref["val"][r, :n][o], rtol=VAL_RTOL)
        assert (idx[r, n:] == -1).all() and (val[r, n:] == 0).all()


@pytest.mark.parametrize("name", ["ref_itemknn_data", "ref_itemknn_empty_col", "ref_itemknn_x_in", "ref_itemknn_larger_matrix",
                                  "syn_itemknn_2000x500"])
def test_fit_golden(hip, name):
    from recpack_amd.algorithms import to_binary_csr

    g = load_golden(name)
    X = to_binary_csr(csr_from(g, "X"))
    cand = csr_from(g, "cos")
    for K in g["Ks"]:
        K = int(K)
        _, m, idx, val, nnz = fit_arrays(hip, X, K)
        ref = csr_from(g, f"S{K}")
        bad, exact = tie_aware_topk(cand, ref, idx, nnz, eps=1e-9)
        assert not bad, (name, K, bad[:5])
        topk_vals_close(ref, idx, val, nnz, rtol=1e-11)
        assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, m.K))


def test_fit_golden_thresholds_10000x2000(hip):
    g = load_golden("syn_itemknn_10000x2000")
    X = csr_from(g, "X")
    _, m, idx, val, nnz = fit_arrays(hip, X, 100)
    ref = csr_from(g, "S100")
    assert (np.diff(ref.indptr) == nnz).all()
    topk_vals_close(ref, idx, val, nnz, rtol=1e-11)
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, 100))


def test_dropin_class_matches_reference_values(hip):
    # recpack/tests/test_algorithms/test_nearest_neighbour.py:44-70
    from recpack_amd import ItemKNN

    g = load_golden("ref_itemknn_data")
    algo = ItemKNN(K=2).fit(csr_from(g, "X"))
    e = 2 / np.sqrt(6)
    expected = np.array([[0, 0.5, e], [0.5, 0, e], [e, e, 0]])
    np.testing.assert_almost_equal(algo.similarity_matrix_.toarray(), expected)
    _in = sp.csr_matrix(([1, 1, 1], ([0, 1, 2], [0, 1, 2])), shape=(3, 3))
    np.testing.assert_almost_equal(algo.predict(_in).toarray(), expected, decimal=6)
    _in = sp.csr_matrix(([1, 1], ([0, 0], [0, 1])), shape=(1, 3))
    np.testing.assert_almost_equal(algo.predict(_in).toarray(), [[0.5, 0.5, 2 * e]], decimal=6)


def test_dropin_normalize_sim_and_warnings(hip):
    from recpack_amd import ItemKNN

    g = load_golden("ref_itemknn_data_normsim")
    algo = ItemKNN(K=2, normalize_sim=True).fit(csr_from(g, "X"))
    np.testing.assert_allclose(algo.similarity_matrix_.toarray(), csr_from(g, "S2").toarray(), rtol=1e-12)
    np.testing.assert_array_almost_equal(algo.similarity_matrix_.sum(axis=1), 1)
    # recpack/tests/test_algorithms/test_algorithms_base.py:49-67: an item without neighbours warns
    x_in = sp.csr_matrix(([1, 2, 1, 1, 1, 1], ([0, 0, 0, 2, 2, 2], [0, 2, 3, 1, 3, 1])), shape=(10, 5))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ItemKNN(2).fit(x_in)
        assert len(w) >= 1 and "1 items" in str(w[-1].message)


@pytest.mark.parametrize("shape,density,K,seed", [
    ((20000, 2000, 0.004), None, 200, 11),     # one LDS tile
    ((3000, 40000, 0.0008), None, 50, 12),     # two column tiles + merge
    ((150000, 300, 0.02), None, 20, 13),       # long user lists: rows cut into slices + partial sums
    ((500, 64, 0.5), None, 1000, 14),          # K >= n_items, dense rows
])
def test_fit_synthetic_vs_oracle(hip, shape, density, K, seed):
    from recpack_amd.synth import powerlaw_csr

    X = powerlaw_csr(shape[0], shape[1], shape[2], seed=seed).to_scipy()
    _, m, idx, val, nnz = fit_arrays(hip, X, K)
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, m.K))


def test_fit_tie_heavy(hip):
    """Many exact ties at the K boundary (all users have the same two-item baskets): smaller column wins."""
    rng = np.random.default_rng(5)
    n_items = 3000
    pairs = rng.integers(0, n_items, size=(20000, 2))
    rows = np.repeat(np.arange(20000), 2)
    X = sp.csr_matrix((np.ones(40000), (rows, pairs.ravel())), shape=(20000, n_items))
    X.sum_duplicates()
    X.data[:] = 1
    _, m, idx, val, nnz = fit_arrays(hip, X, 5)
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, 5))
    # one hub item co-occurring exactly once with thousands of degree-1 items: > RP_SORT_CAP exact ties
    n = 6000
    rows = np.repeat(np.arange(n - 1), 2)
    cols = np.stack([np.zeros(n - 1, dtype=np.int64), np.arange(1, n)], axis=1).ravel()
    H = sp.csr_matrix((np.ones(2 * (n - 1)), (rows, cols)), shape=(n - 1, n))
    _, m, idx, val, nnz = fit_arrays(hip, H, 7)
    assert nnz[0] == 7 and (idx[0, :7] == np.arange(1, 8)).all()
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(H, 7))


def test_fit_long_runs_of_empty_users(hip):
    """K0's partition pass caches the row pointers of the users an 8192-entry window meets: a run of more than 8192 history-less
    users inside a window must not overrun that cache - X^T, the degrees and both fits against the oracle."""
    _lib, ctx = hip
    rng = np.random.default_rng(77)
    # (a) 50 000 users, two of them with a history, far apart
    rows = np.r_[np.full(6, 3), np.full(5, 49_000)]
    cols = np.r_[[0, 2, 3, 5, 7, 8], [0, 3, 4, 8, 9]]
    A = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(50_000, 10))
    # (b) 95 % empty rows, runs of up to 40 000 empty users between active ones, enough entries for several partition blocks
    U, I = 400_000, 700
    active = np.sort(rng.choice(U, U // 20, replace=False))
    active = active[(active < 100_000) | (active > 140_000)]  # one run of 40 000 empty users in the middle
    deg = rng.integers(1, 6, active.size)
    r = np.repeat(active, deg)
    c = rng.integers(0, I, r.size)
    Bm = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(U, I))
    Bm.sum_duplicates()
    Bm.data[:] = 1
    # (c) the same with several column tiles (tile pointers of empty rows)
    c2 = rng.integers(0, 30_000, r.size)
    C = sp.csr_matrix((np.ones(r.size), (r, c2)), shape=(U, 30_000))
    C.sum_duplicates()
    C.data[:] = 1
    for X, K in ((A, 4), (Bm, 20), (C, 10)):
        dX, m, idx, val, nnz = fit_arrays(hip, X, K)
        np.testing.assert_array_equal(dX.item_degrees(), np.asarray(X.sum(axis=0)).ravel().astype(np.int32))
        assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, K))
    B_gpu = _lib.ease_fit(ctx, _lib.DeviceCSR.from_host(ctx, Bm.indptr, Bm.indices, Bm.shape), 5.0).dense_download()
    B_ref = onp.ease_fit(Bm, 5.0)
    assert np.abs(B_gpu - B_ref).max() <= 1e-4 * np.abs(B_ref).max()


def test_fit_edge_cases(hip):
    _lib, ctx = hip
    # empty matrix, empty rows / columns
    E = sp.csr_matrix((5, 7))
    _, m, idx, val, nnz = fit_arrays(hip, E, 3)
    assert (nnz == 0).all() and (idx == -1).all()
    X = sp.csr_matrix(([1, 1, 1], ([0, 0, 3], [1, 5, 5])), shape=(6, 8))
    _, m, idx, val, nnz = fit_arrays(hip, X, 3)
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, 3))
    # unsorted column indices are fine within one tile; int64 indptr accepted
    Xu = sp.csr_matrix((np.ones(4), np.array([5, 1, 5, 2], dtype=np.int32), np.array([0, 2, 2, 4], dtype=np.int64)), shape=(3, 8))
    dX = _lib.DeviceCSR.from_host(ctx, Xu.indptr.astype(np.int64), Xu.indices, Xu.shape)
    m = _lib.itemknn_fit(ctx, dX, 2)
    idx, val, nnz = m.topk_download()
    Xs = Xu.copy()
    Xs.sort_indices()
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(Xs, 2))
    # row shards give the same rows as the full fit
    from recpack_amd.synth import powerlaw_csr

    Y = powerlaw_csr(4000, 900, 0.01, seed=3).to_scipy()
    _, _, idx_f, val_f, nnz_f = fit_arrays(hip, Y, 30)
    _, _, idx_s, val_s, nnz_s = fit_arrays(hip, Y, 30, rows=(300, 711))
    np.testing.assert_array_equal(idx_s, idx_f[300:711])
    np.testing.assert_array_equal(val_s, val_f[300:711])
    # bad input -> error code, not a crash
    with pytest.raises(_lib.RecpackHipError):
        _lib.DeviceCSR.from_host(ctx, np.array([0, 2, 1], dtype=np.int32), np.array([0, 1], dtype=np.int32), (2, 3))
    with pytest.raises(_lib.RecpackHipError):
        _lib.DeviceCSR.from_host(ctx, np.array([0, 1], dtype=np.int32), np.array([9], dtype=np.int32), (1, 3))
    with pytest.raises(_lib.RecpackHipError):
        _lib.itemknn_fit(ctx, dX, 0)


# ---------------------------------------------------------------------------------------------------
def check_topk_scores(scores_ref: sp.csr_matrix, idx, val, nnz, Kp, exclude=None, eps=1e-6, hist=None, S=None):
    """Tie-aware check of a fused top-Kp against float64 oracle scores.  With exact 64-bit sums (hist is None) the only
    rounding is the f32 storage of the similarities: eps 1e-6, values rtol 2e-7.  With the default 32-bit per-user-scaled
    sums pass the histories `hist` and the model `S`: every score then carries the stated absolute bound
    (parity.fx32_bound), which widens the tie band by twice that and the value tolerance by once."""
    from parity import fx32_bound

    ridx, rval, rnnz = onp.topk_rows(scores_ref, Kp, exclude=exclude)
    cand = scores_ref
    if exclude is not None:
        cand = scores_ref - scores_ref.multiply(onp.binarize(exclude))
        cand.eliminate_zeros()
    ref_sel = onp.topk_to_csr(ridx, rval, rnnz, scores_ref.shape[1])
    bound = fx32_bound(hist, S) if hist is not None else None
    bad, exact = tie_aware_topk(sp.csr_matrix(cand), ref_sel, idx, nnz, eps=eps, abs_tol=None if bound is None else 2.0 * bound)
    assert not bad, bad[:5]
    lut = sp.csr_matrix(cand)
    for r in range(0, idx.shape[0], max(1, idx.shape[0] // 200)):
        n = nnz[r]
        assert (np.diff(idx[r, :n]) > 0).all()
        got = val[r, :n]
        want = np.asarray(lut[r, idx[r, :n]].todense()).ravel()
        np.testing.assert_allclose(got, want, rtol=PRED_RTOL, atol=0.0 if bound is None else float(bound[r]))
    return exact


@pytest.mark.parametrize("name", ["ref_itemknn_data", "ref_itemknn_x_in", "ref_itemknn_larger_matrix", "syn_itemknn_2000x500"])
def test_predict_full_golden(hip, name):
    """predict() from the reference's own similarity matrix reproduces the reference's scores."""
    _lib, ctx = hip
    g = load_golden(name)
    n_items = int(g["X_shape"][1])
    for K in g["Ks"]:
        K = int(K)
        S = csr_from(g, f"S{K}")
        Kst = max(1, int(np.diff(S.indptr).max()))
        idx = np.full((n_items, Kst), -1, dtype=np.int32)
        val = np.zeros((n_items, Kst))
        nnz = np.diff(S.indptr).astype(np.int32)
        for r in range(n_items):
            idx[r, : nnz[r]] = S.indices[S.indptr[r]: S.indptr[r + 1]]
            val[r, : nnz[r]] = S.data[S.indptr[r]: S.indptr[r + 1]]
        model = _lib.Model.from_topk_host(ctx, idx, val, nnz)
        n = 0
        while f"Pin{n}_indptr" in g:
            from recpack_amd.algorithms import to_binary_csr

            Pin = to_binary_csr(csr_from(g, f"Pin{n}"))
            dP = _lib.DeviceCSR.from_host(ctx, Pin.indptr, Pin.indices, Pin.shape)
            indptr, indices, data = _lib.predict_full(ctx, model, dP)
            out = sp.csr_matrix((data, indices, indptr), shape=Pin.shape)
            ref = csr_from(g, f"Pout{K}_{n}")
            ref.eliminate_zeros()
            assert (out.indptr == ref.indptr).all() and (out.indices == ref.indices).all()
            np.testing.assert_allclose(out.data, ref.data, rtol=PRED_RTOL)
            n += 1


@pytest.mark.parametrize("shape,K,Kp,seed", [((6000, 1500, 0.006), 100, 50, 21), ((2500, 33000, 0.001), 40, 20, 22)])
def test_predict_topk_vs_oracle(hip, shape, K, Kp, seed):
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(shape[0], shape[1], shape[2], seed=seed).to_scipy()
    dX, m, idx, val, nnz = fit_arrays(hip, X, K)
    S = onp.topk_to_csr(idx, val, nnz, X.shape[1])
    users = slice(0, 1200)
    scores = onp.predict_full(X[users], S)
    # full scores
    ip, ind, dat = _lib.predict_full(ctx, m, dX, 0, 1200)
    out = sp.csr_matrix((dat, ind, ip), shape=scores.shape)
    assert (out.indptr == scores.indptr).all() and (out.indices == scores.indices).all()
    np.testing.assert_allclose(out.data, scores.data, rtol=PRED_RTOL)
    # fused top-Kp, with and without history removal
    for excl in (False, True):
        pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl, user_begin=0, user_end=1200)
        check_topk_scores(scores, pi, pv, pn, Kp, exclude=X[users] if excl else None, hist=X[users], S=S)
    # item filter (ExcludeItems / SelectItems fused into the kernel), with and without history removal
    keep = np.random.default_rng(seed).random(X.shape[1]) < 0.7
    for excl in (False, True):
        pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl, user_begin=0, user_end=1200, item_keep=keep)
        assert keep[pi[pi >= 0]].all()
        check_topk_scores(onp.pipeline_scores(scores, item_keep=keep), pi, pv, pn, Kp, exclude=X[users] if excl else None, hist=X[users], S=S)
    # a user range in the middle equals the same rows of the full call
    pi2, pv2, pn2 = _lib.predict_topk(ctx, m, dX, Kp, user_begin=100, user_end=300)
    pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, user_begin=0, user_end=1200)
    np.testing.assert_array_equal(pi2, pi[100:300])
    np.testing.assert_array_equal(pn2, pn[100:300])


@pytest.mark.parametrize("exact64", [True, False])
@pytest.mark.parametrize("scale,signed", [(1.0, False), (3000.0, True), (1e-6, True)])
def test_predict_signed_models_and_reproducibility(hip, scale, signed, exact64):
    """Neighbour-list models with any sign and magnitude (e.g. a pruned EASE matrix loaded with from_topk_host):
    the fixed-point scale follows the model's largest |value| and the longest history (64-bit sums: exact) or the user's
    own score bound (32-bit sums, the default: within parity.fx32_bound); either way results are bit-identical from run
    to run (integer sums do not depend on the order of the adds)."""
    from parity import fx32_bound

    _lib, ctx = hip
    ctx.set_option("predict_exact64", 1 if exact64 else 0)
    rng = np.random.default_rng(5)
    I, K, U, Kp = 5000, 30, 700, 25
    idx = np.full((I, K), -1, np.int32)
    val = np.zeros((I, K))
    nnz = rng.integers(0, K + 1, I).astype(np.int32)
    for i in range(I):
        c = np.sort(rng.choice(I, nnz[i], replace=False))
        idx[i, : nnz[i]] = c
        v = rng.random(nnz[i]) + 0.01
        if signed:
            v *= rng.choice([-1.0, 1.0], nnz[i])
        val[i, : nnz[i]] = v * scale
    val = val.astype(np.float32).astype(np.float64)  # the model stores f32 similarities
    S = onp.topk_to_csr(idx, val, nnz, I)
    deg = np.minimum(rng.zipf(1.6, U) + 1, 400)
    rows = np.repeat(np.arange(U), deg)
    X = sp.csr_matrix((np.ones(len(rows)), (rows, rng.integers(0, I, len(rows)))), shape=(U, I))
    X.data[:] = 1
    X.sum_duplicates()
    X.data[:] = 1
    model = _lib.Model.from_topk_host(ctx, idx, val, nnz)
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    scores = sp.csr_matrix(X @ S)
    ip, ind, dat = _lib.predict_full(ctx, model, dX)
    out = sp.csr_matrix((dat, ind, ip), shape=scores.shape)
    dense_ref, dense_out = scores.toarray(), out.toarray()
    # sums of signed terms can cancel: the tolerance is relative to the sum of magnitudes
    mag = (X @ abs(S)).toarray()
    assert np.abs(dense_out - dense_ref).max() <= 1e-12 * max(1.0, mag.max())  # the full result is exact to f64 rounding
    assert (np.abs(dense_out - dense_ref) <= 1e-13 * mag + 1e-300).all()
    runs = []
    tol = 1e-12 * mag.max(axis=1) if exact64 else 2.0 * fx32_bound(X, S) + 1e-12 * mag.max(axis=1)
    for excl in (False, True, False, True):
        pi, pv, pn = _lib.predict_topk(ctx, model, dX, Kp, exclude_history=excl)
        runs.append((pi, pv, pn))
        cand = dense_ref.copy()
        if excl:
            cand[X.toarray() > 0] = 0
        for r in range(0, U, 7):
            nz = np.nonzero(cand[r])[0]
            n = min(Kp, len(nz))
            assert pn[r] == n
            if n == 0:
                continue
            order = nz[np.lexsort((nz, -cand[r, nz]))][:n]  # score desc, column asc
            t = cand[r, order[-1]]
            got = pi[r, :n]
            assert (np.diff(got) > 0).all()
            must = nz[cand[r, nz] > t + tol[r]]
            assert np.isin(must, got).all()
            assert (cand[r, got] >= t - tol[r]).all()
            np.testing.assert_allclose(pv[r, :n], cand[r, got], rtol=2e-7, atol=0.0 if exact64 else 0.5 * tol[r])
    ctx.set_option("predict_exact64", 0)
    for a, b in ((0, 2), (1, 3)):
        for x, y in zip(runs[a], runs[b]):
            np.testing.assert_array_equal(x, y)


def test_predict_edge_cases(hip):
    from recpack_amd import ItemKNN

    g = load_golden("ref_itemknn_data")
    algo = ItemKNN(K=2).fit(csr_from(g, "X"))
    # users without history get empty rows; a user whose only item has no neighbours warns
    P = sp.csr_matrix((4, 3))
    out = algo.predict(P)
    assert out.shape == (4, 3) and out.nnz == 0
    ec = load_golden("ref_itemknn_empty_col")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a2 = ItemKNN(K=2).fit(csr_from(ec, "X"))
    with pytest.warns(UserWarning, match="failed to recommend any items for 1 users"):
        a2.predict(sp.csr_matrix(([1], ([0], [0])), shape=(2, 3)))
    top = algo.predict_topk(sp.csr_matrix(([1, 1], ([0, 0], [0, 1])), shape=(1, 3)), 1)
    assert top.nnz == 1 and top.indices[0] == 2
    top = algo.predict_topk(sp.csr_matrix(([1, 1], ([0, 0], [0, 2])), shape=(1, 3)), 3, exclude_history=True)
    assert top.indices.tolist() == [1]


def test_sharded_fit_nccl_world1(hip):
    """The N > 1 code path (torch views of the library's arrays, RCCL all-gather, model from device arrays) on one GPU."""
    import torch
    import torch.distributed as dist

    from recpack_amd import dist as rdist
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        X = powerlaw_csr(5000, 1200, 0.008, seed=77).to_scipy()
        dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
        parts = rdist.partition_by_cost(rdist.item_row_cost(dX.item_degrees(), X.nnz, dX.num_products()), 1)
        full = rdist.fit_itemknn_sharded(ctx, dX, 40, parts, torch.device("cuda", 0))
        idx, val, nnz = full.topk_download()
        assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, 40))
        pi, pv, pn = _lib.predict_topk(ctx, full, dX, 10, user_begin=0, user_end=200)
        m2 = _lib.itemknn_fit(ctx, dX, 40)
        qi, qv, qn = _lib.predict_topk(ctx, m2, dX, 10, user_begin=0, user_end=200)
        np.testing.assert_array_equal(pi, qi)
        # the library's own RCCL communicator at one rank: every entry point of the transport self-test (bench.py --selftest)
        comm = rdist.make_comm(ctx, torch.device("cuda", 0), transport="rccl")
        st = comm.selftest(words=10_000)
        assert comm.is_rccl and st["ok"] and set(st["ops"]) == set(_lib.Comm.SELFTEST_OPS), st
        comm.close()
    finally:
        dist.destroy_process_group()


def test_gpu_generator_matches_numpy():
    """bench.py builds its matrices with the torch backend on the GPU: same bits as the numpy generator."""
    import torch

    from recpack_amd.synth import powerlaw_csr

    ref = powerlaw_csr(20000, 3000, 0.003, seed=5)
    ip, ix = powerlaw_csr(20000, 3000, 0.003, seed=5, device=torch.device("cuda", 0))
    np.testing.assert_array_equal(ip.cpu().numpy(), ref.indptr)
    np.testing.assert_array_equal(ix.cpu().numpy(), ref.indices)


@pytest.mark.parametrize("name", ["ref_itemknn_data_condprob", "ref_itemknn_data_condprob_pd1.0", "ref_itemknn_data_condprob_pd0.2",
                                  "ref_itemknn_data_condprob_pd0.5", "syn_itemknn_condprob_2000x500", "syn_itemknn_condprob0_2000x500"])
def test_fit_conditional_probability_golden(hip, name):
    from recpack_amd.algorithms import to_binary_csr

    _lib, ctx = hip
    g = load_golden(name)
    X = to_binary_csr(csr_from(g, "X"))
    cand = csr_from(g, "cos")
    pd = float(g["pop_discount"])
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    for K in g["Ks"]:
        K = int(K)
        m = _lib.itemknn_fit(ctx, dX, K, similarity=_lib.SIM_CONDITIONAL_PROBABILITY, pop_discount=pd)
        idx, val, nnz = m.topk_download()
        ref = csr_from(g, f"S{K}")
        bad, exact = tie_aware_topk(cand, ref, idx, nnz, eps=1e-9)
        assert not bad, (name, K, bad[:5])
        topk_vals_close(ref, idx, val, nnz, rtol=1e-11)
        assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, m.K, similarity="conditional_probability", pop_discount=pd))


def test_conditional_probability_dropin_and_big(hip):
    # recpack/tests/test_algorithms/test_nearest_neighbour.py:127-181 through the drop-in class
    from recpack_amd import ItemKNN
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    g = load_golden("ref_itemknn_data")
    data = csr_from(g, "X")
    algo = ItemKNN(K=2, similarity="conditional_probability").fit(data)
    np.testing.assert_almost_equal(algo.similarity_matrix_.toarray(), [[0, 1 / 2, 2 / 2], [1 / 2, 0, 2 / 2], [2 / 3, 2 / 3, 0]])
    for pd in (1, 0.2, 0.5):
        algo = ItemKNN(K=2, similarity="conditional_probability", pop_discount=pd).fit(data)
        expected = np.array([[0, 1 / (2 * 2**pd), 2 / (2 * 3**pd)], [1 / (2 * 2**pd), 0, 2 / (2 * 3**pd)],
                             [2 / (3 * 2**pd), 2 / (3 * 2**pd), 0]])
        np.testing.assert_almost_equal(algo.similarity_matrix_.toarray(), expected)
    # several column tiles, sliced rows, light and heavy rows
    X = powerlaw_csr(30000, 30000, 0.001, seed=41).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    m = _lib.itemknn_fit(ctx, dX, 30, similarity=_lib.SIM_CONDITIONAL_PROBABILITY, pop_discount=0.3)
    idx, val, nnz = m.topk_download()
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, 30, similarity="conditional_probability", pop_discount=0.3))


@pytest.mark.parametrize("name", ["ref_itemknn_data_normX", "syn_itemknn_normX_2000x500", "syn_itemknn_normX_condprob_2000x500",
                                  "syn_itemknn_normX_6000x14000"])
def test_fit_normalize_X_golden(hip, name):
    """normalize_X=True (nearest_neighbour.py:207-210): float64 weighted Gram; tie-aware sets (eps 1e-9) and values
    (1e-11) against the reference's goldens, and against the oracle's closed form."""
    from recpack_amd.algorithms import to_binary_csr

    _lib, ctx = hip
    g = load_golden(name)
    X = to_binary_csr(csr_from(g, "X"))
    sim_name, pd = str(g["similarity"]), float(g["pop_discount"])
    sim = _lib.SIM_COSINE if sim_name == "cosine" else _lib.SIM_CONDITIONAL_PROBABILITY
    cand = csr_from(g, "cos") if "cos_indptr" in g else onp.itemknn_fit_normalized(X, None, similarity=sim_name, pop_discount=pd)
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    for K in g["Ks"]:
        K = int(K)
        m = _lib.itemknn_fit(ctx, dX, K, similarity=sim, pop_discount=pd, normalize_X=True)
        idx, val, nnz = m.topk_download()
        ref = csr_from(g, f"S{K}")
        assert (np.diff(ref.indptr) == nnz).all()
        live = np.arange(idx.shape[1])[None, :] < nnz[:, None]
        assert ((idx[:, 1:] > idx[:, :-1]) | ~live[:, 1:]).all()  # canonical rows
        bad, exact = tie_aware_topk(cand, ref, idx, nnz, eps=1e-9)
        assert not bad, (name, K, bad[:5])
        topk_vals_close(ref, idx, val, nnz, rtol=1e-11)
        o = onp.itemknn_fit_normalized(X, m.K, similarity=sim_name, pop_discount=pd)
        osel = onp.topk_to_csr(o["idx"], o["val"], o["nnz"], X.shape[1])
        bad, exact = tie_aware_topk(cand, osel, idx, nnz, eps=1e-9)
        assert not bad, ("oracle", name, K, bad[:5])
        topk_vals_close(osel, idx, val, nnz, rtol=1e-11)
        # the model predicts like any other neighbour-list model
        ip, ind, dat = _lib.predict_full(ctx, m, dX, 0, min(200, X.shape[0]))
        S_gpu = onp.topk_to_csr(idx, val, nnz, X.shape[1])
        want = onp.predict_full(X[: min(200, X.shape[0])], S_gpu)
        out = sp.csr_matrix((dat, ind, ip), shape=want.shape)
        assert (out.indptr == want.indptr).all() and (out.indices == want.indices).all()
        np.testing.assert_allclose(out.data, want.data, rtol=PRED_RTOL)


def test_normalize_X_dropin(hip):
    # recpack/tests/test_algorithms/test_nearest_neighbour.py:73-107 through the drop-in class
    import math

    from recpack_amd import ItemKNN

    g = load_golden("ref_itemknn_data")
    algo = ItemKNN(K=2, similarity="cosine", normalize_X=True).fit(csr_from(g, "X"))
    a = 1 / (3 * 3)
    b = 1 / (3 * 3) + 1 / (2 * 2)
    d = math.sqrt((1 / 2) ** 2 + (1 / 3) ** 2)
    f = math.sqrt(2 * (1 / 2) ** 2 + (1 / 3) ** 2)
    expected = np.array([[0, a / (d * d), b / (d * f)], [a / (d * d), 0, b / (d * f)], [b / (d * f), b / (d * f), 0]])
    np.testing.assert_almost_equal(algo.similarity_matrix_.toarray(), expected)
    # with normalize_sim on top (nearest_neighbour.py:220-222) every row sums to one
    algo = ItemKNN(K=2, normalize_X=True, normalize_sim=True).fit(csr_from(g, "X"))
    np.testing.assert_array_almost_equal(np.asarray(algo.similarity_matrix_.sum(axis=1)).ravel(), 1)
    # a shard of rows equals the same rows of the full fit
    _lib, ctx = hip
    from recpack_amd.synth import powerlaw_csr

    X = powerlaw_csr(3000, 900, 0.01, seed=9).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    full = _lib.itemknn_fit(ctx, dX, 40, normalize_X=True).topk_download()
    part = _lib.itemknn_fit(ctx, dX, 40, row_begin=300, row_end=700, normalize_X=True).topk_download()
    np.testing.assert_array_equal(part[2], full[2][300:700])
    np.testing.assert_allclose(part[1], full[1][300:700], rtol=1e-12)


def test_csr_from_coo(hip):
    """COO ingestion on the device equals scipy's csr_matrix((ones, (uid, iid))) binarised
    (recpack/matrix/interaction_matrix.py:202-217 + recpack/util.py:99-109): any order, repeats, empty users."""
    import torch

    _lib, ctx = hip
    rng = np.random.default_rng(4)
    for (U, I, n) in [(50, 20, 300), (3000, 70000, 200000), (7, 5, 0), (1000000, 100000, 3000000), (5, 3, 4097), (1200, 900, 50000)]:
        uid = rng.integers(0, max(1, U - 3), n)  # the last users stay empty
        iid = (rng.zipf(1.3, n) - 1) % I if n else np.zeros(0, dtype=np.int64)
        if n:
            uid[: n // 10], iid[: n // 10] = uid[n // 2: n // 2 + n // 10], iid[n // 2: n // 2 + n // 10]  # repeats
        ref = sp.csr_matrix((np.ones(n), (uid, iid)), shape=(U, I))
        ref.data[:] = 1
        ref.sort_indices()
        for dev in (False, True):
            if dev:
                d = _lib.DeviceCSR.from_coo(ctx, torch.as_tensor(uid, dtype=torch.int64).cuda(), torch.as_tensor(iid, dtype=torch.int64).cuda(), (U, I))
            else:
                d = _lib.DeviceCSR.from_coo(ctx, uid, iid, (U, I))
            ip, ix = d.download()
            assert d.nnz == ref.nnz
            np.testing.assert_array_equal(ip, ref.indptr)
            np.testing.assert_array_equal(ix, ref.indices)
            np.testing.assert_array_equal(d.item_degrees(), np.asarray(ref.sum(axis=0)).ravel().astype(np.int32))
    # the fit on a COO-built matrix equals the fit on the host CSR
    d2 = _lib.DeviceCSR.from_host(ctx, ref.indptr, ref.indices, ref.shape)
    a, b = _lib.itemknn_fit(ctx, d, 30).topk_download(), _lib.itemknn_fit(ctx, d2, 30).topk_download()
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    with pytest.raises(_lib.RecpackHipError):
        _lib.DeviceCSR.from_coo(ctx, np.array([0, 5]), np.array([1, 1]), (5, 3))
    with pytest.raises(_lib.RecpackHipError):
        _lib.DeviceCSR.from_coo(ctx, np.array([0, 1]), np.array([1, -1]), (5, 3))


def test_csr_for_a_row_shard(hip):
    """rphip_csr_from_device_rows: X^T filled for a rank's item rows only gives the same neighbour lists for those rows;
    other rows, EASE and normalize_X are refused."""
    import torch

    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    dev = torch.device("cuda", 0)
    ip, ix = powerlaw_csr(9000, 30000, 0.002, seed=91, device=dev)  # two column tiles
    full = _lib.DeviceCSR.from_device(ctx, ip, ix, (9000, 30000))
    ref = _lib.itemknn_fit(ctx, full, 60).topk_download()
    for b, e in ((0, 4000), (4000, 30000), (12345, 12346)):
        part = _lib.DeviceCSR.from_device(ctx, ip, ix, (9000, 30000), fit_rows=(b, e))
        np.testing.assert_array_equal(part.item_degrees(), full.item_degrees())
        got = _lib.itemknn_fit(ctx, part, 60, row_begin=b, row_end=e).topk_download()
        for x, y in zip(got, ref):
            np.testing.assert_array_equal(x, y[b:e])
        with pytest.raises(_lib.RecpackHipError):
            _lib.itemknn_fit(ctx, part, 60)
        with pytest.raises(_lib.RecpackHipError):
            _lib.ease_fit(ctx, part, 10.0)
        with pytest.raises(_lib.RecpackHipError):
            _lib.itemknn_fit(ctx, part, 60, row_begin=b, row_end=e, normalize_X=True)
    # predict is unaffected by the shard
    m = _lib.itemknn_fit(ctx, full, 60)
    a = _lib.predict_topk(ctx, m, part, 20, user_begin=0, user_end=500)
    c = _lib.predict_topk(ctx, m, full, 20, user_begin=0, user_end=500)
    for x, y in zip(a, c):
        np.testing.assert_array_equal(x, y)


def test_csr_from_a_misaligned_index_array(hip):
    """K0 reads four column indices per 16-byte load when the caller's index array allows it: a view that starts 4 bytes into an
    allocation takes the scalar loads, the transposed matrix, degrees and neighbour lists are the same; several column tiles and a
    single tile; a matrix whose entries end inside a chunk of 4 096."""
    import torch

    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    dev = torch.device("cuda", 0)
    for U, I, dens, seed in ((9000, 30000, 0.002, 92), (5000, 3000, 0.004, 93)):
        ip, ix = powerlaw_csr(U, I, dens, seed=seed, device=dev)
        assert ix.data_ptr() % 16 == 0
        holder = torch.empty(ix.numel() + 1, dtype=torch.int32, device=dev)
        holder[1:] = ix
        ix_off = holder[1:]
        assert ix_off.data_ptr() % 16 == 4
        a = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
        b = _lib.DeviceCSR.from_device(ctx, ip, ix_off, (U, I))
        np.testing.assert_array_equal(a.item_degrees(), b.item_degrees())
        assert a.num_products() == b.num_products()
        ra = _lib.itemknn_fit(ctx, a, 40).topk_download()
        rb = _lib.itemknn_fit(ctx, b, 40).topk_download()
        for x, y in zip(ra, rb):
            np.testing.assert_array_equal(x, y)
        # against scipy's transpose
        Xs = sp.csr_matrix((np.ones(ix.numel(), dtype=np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
        np.testing.assert_array_equal(b.item_degrees(), np.asarray(Xs.sum(axis=0)).ravel().astype(np.int32))
        assert b.num_products() == int((np.diff(ip.cpu().numpy()).astype(np.int64) ** 2).sum())


def test_limits_and_dense_corners(hip):
    """The corners of the supported range: K and Kp at their maxima, a column every user has, a user with every item,
    more neighbours requested than items exist."""
    _lib, ctx = hip
    rng = np.random.default_rng(8)
    U, I = 600, 2300
    X = sp.random(U, I, density=0.01, format="csr", random_state=7)
    X.data[:] = 1
    X = X.tolil()
    X[:, 5] = 1      # an item every user has
    X[17, :] = 1     # a user with every item
    X = sp.csr_matrix(X)
    X.sort_indices()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    for K in (2048, 5000):  # RP_K_MAX, and a request beyond the number of items (clipped to I - 1 = 2299: one column tile takes any K)
        m = _lib.itemknn_fit(ctx, dX, K)
        assert m.K == min(K, I - 1)
        idx, val, nnz = m.topk_download()
        assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, m.K))
        S = onp.topk_to_csr(idx, val, nnz, I)
        scores = onp.predict_full(X[:100], S)
        for Kp in (384, 385, 2048):  # wave kernel's largest pool, first value of the workgroup kernel, RP_K_MAX
            pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=True, user_begin=0, user_end=100)
            check_topk_scores(scores, pi, pv, pn, Kp, exclude=X[:100], hist=X[:100] if Kp <= 384 else None, S=S)
        # (round 6) beyond RP_K_MAX: predict()'s full scores per chunk of users, cut to Kp on the device - exact float64 sums (hist=None),
        # with and without history removal, with an item filter; Kp past the number of scores returns them all
        for Kp in (2049, 2200, I):
            for excl in (True, False):
                pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl, user_begin=0, user_end=100)
                assert pi.shape == (100, min(Kp, I))
                check_topk_scores(scores, pi, pv, pn, min(Kp, I), exclude=X[:100] if excl else None, S=S)
        keep = (np.arange(I) % 3 != 0).astype(np.uint8)
        pi, pv, pn = _lib.predict_topk(ctx, m, dX, 2100, exclude_history=True, item_keep=keep, user_begin=0, user_end=100)
        check_topk_scores(onp.pipeline_scores(scores, item_keep=keep), pi, pv, pn, 2100, exclude=X[:100], S=S)
        assert not np.isin(pi[pi >= 0] % 3, [0]).any()


def test_fit_any_k_on_one_tile(hip):
    """ItemKNN(K=5000) on 20 000 x 8 000: up to RP_TILE_MAX items the fit takes any K, as the reference
    does (recpack/algorithms/base.py:302-304, recpack/util.py:63-66); with several column tiles K stays <= 2048."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(20000, 8000, 0.004, seed=41).to_scipy()
    _, m, idx, val, nnz = fit_arrays(hip, X, 5000)
    assert m.K == 5000
    assert_same_as_oracle(idx, val, nnz, onp.itemknn_fit(X, 5000))
    # the drop-in class and predict() on such a model
    import recpack_amd

    algo = recpack_amd.ItemKNN(K=5000).fit(X)
    S = algo.similarity_matrix_
    assert S.shape == (8000, 8000) and S.getnnz(axis=1).max() <= 5000
    got = algo.predict(X[:300])
    want = onp.predict_full(X[:300], S)
    assert abs(got - want).max() <= 2e-7 * abs(want).max()
    Xw = powerlaw_csr(200, 26000, 0.002, seed=42).to_scipy()  # two column tiles: K x tiles >= 32 768 takes the large-K merge (round 6) ...
    Sw = recpack_amd.ItemKNN(K=17000).fit(Xw).similarity_matrix_
    rw = onp.itemknn_fit(Xw, 17000)
    assert (Sw != onp.topk_to_csr(rw["idx"], rw["val"], rw["nnz"], 26000)).nnz == 0
    # ... and so does normalize_X (three tiles of its float64 fit; checked tie-aware in test_fit_normalize_X_any_k): no ValueError left
    Sn = recpack_amd.ItemKNN(K=17000, normalize_X=True).fit(Xw).similarity_matrix_
    assert Sn.shape == (26000, 26000) and Sn.nnz == Sw.nnz


def test_fit_any_k_over_several_tiles(hip):
    """ItemKNN(K=5000) on 60 000 x 40 000: two column tiles, a row's per-tile lists (2 x 5 000 candidates)
    merged where they lie in memory (k_itemknn_merge<true>); K = 2 500 on the same matrix merges in LDS; both equal the oracle
    (recpack/algorithms/base.py:302-304, recpack/util.py:63-66: the reference takes any K)."""
    from recpack_amd.synth import powerlaw_csr

    X = powerlaw_csr(60000, 40000, 0.002, seed=43).to_scipy()
    rows = np.sort(np.random.default_rng(3).choice(40000, size=2500, replace=False))  # (the oracle's Gram of all 40 000 rows takes minutes)
    rows = np.unique(np.r_[rows, np.argsort(-np.diff(X.tocsc().indptr))[:200]])       # ... and the 200 longest rows
    for K in (5000, 2500):
        _, m, idx, val, nnz = fit_arrays(hip, X, K)
        assert m.K == K
        ref = onp.itemknn_fit(X, K, rows=rows)
        assert_same_as_oracle(idx[rows], val[rows], nnz[rows], ref)
    import recpack_amd

    S = recpack_amd.ItemKNN(K=5000).fit(X).similarity_matrix_
    assert S.shape == (40000, 40000) and 2048 < S.getnnz(axis=1).max() <= 5000


def test_fit_any_k_beyond_the_numbered_merges(hip):
    """ItemKNN(K=20000) and K = n_items on 60 000 x 40 000: K x tiles >= 32 768, per-tile lists of min(K, tile)
    entries merged by k_itemknn_merge_big - rows with fewer candidates than K copied out, the popular rows through the MSD radix select on
    the 64-bit key with ties to the smaller column; a tie-heavy matrix (few distinct keys: whole bins tie at the K-th key) on two tiles."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(60000, 40000, 0.004, seed=44).to_scipy()
    rows = np.unique(np.r_[np.sort(np.random.default_rng(5).choice(40000, size=1500, replace=False)), np.argsort(-np.diff(X.tocsc().indptr))[:300]])
    for K in (17000, 40000):
        _, m, idx, val, nnz = fit_arrays(hip, X, K)
        assert m.K == min(K, 39999)
        ref = onp.itemknn_fit(X, K, rows=rows)
        assert_same_as_oracle(idx[rows], val[rows], nnz[rows], ref)
        if K == 17000:
            assert (nnz == 17000).sum() > 50  # (rows that really cut at K: the radix select ran)
        del m, idx, val, nnz
    # ties: 300 users, every item in one of 12 groups of identical columns -> a handful of distinct keys per row
    rng = np.random.default_rng(9)
    base = (rng.random((300, 12)) < 0.4)
    grp = rng.integers(0, 12, size=27000)
    Xt = sp.csr_matrix(base[:, grp].astype(np.float64))
    for K in (17000, 20000):
        _, m, idx, val, nnz = fit_arrays(hip, Xt, K)
        rr = np.sort(rng.choice(27000, size=300, replace=False))
        assert_same_as_oracle(idx[rr], val[rr], nnz[rr], onp.itemknn_fit(Xt, K, rows=rr))


def test_fit_normalize_X_any_k(hip):
    """normalize_X=True with K beyond 2 048 (round 5): 25 000 items = three column tiles of the float64 fit, K = 7 000 merges a row's
    lists where they lie in memory (k_itemknn_merge_weighted<true>; the LDS merge is the path of every smaller K); one tile (3 000
    items) takes any K.  Tie-aware
    against the oracle (the float64 sums depend on the order of the adds in the last bits, as in scipy)."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    for (U, I, dens, Ks) in ((10000, 25000, 0.0015, (7000, 12000)), (6000, 3000, 0.01, (2500,))):  # (12 000 x 3 tiles: the large-K merge, round 6)
        X = powerlaw_csr(U, I, dens, seed=47).to_scipy()
        dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
        cand = onp.itemknn_fit_normalized(X, None)  # every candidate of the similarity (CSR)
        for K in Ks:
            m = _lib.itemknn_fit(ctx, dX, K, normalize_X=True)
            idx, val, nnz = m.topk_download()
            assert m.K == K and nnz.max() > 2048 and (K < 12000 or nnz.max() > 7000)
            live = np.arange(idx.shape[1])[None, :] < nnz[:, None]
            assert ((idx[:, 1:] > idx[:, :-1]) | ~live[:, 1:]).all()  # canonical rows
            o = onp.itemknn_fit_normalized(X, m.K)
            osel = onp.topk_to_csr(o["idx"], o["val"], o["nnz"], X.shape[1])
            assert (np.diff(osel.indptr) == nnz).all()
            bad, exact = tie_aware_topk(cand, osel, idx, nnz, eps=1e-9)
            assert not bad, (I, K, bad[:5])
            topk_vals_close(osel, idx, val, nnz, rtol=1e-11)


@pytest.mark.parametrize("nproc,shape", [(2, (20000, 30000, 0.0015, 50)), (3, (6000, 900, 0.01, 20))])
def test_sharded_fit_on_hip_multi_rank(hip, nproc, shape):
    """The multi-GPU ItemKNN path with world > 1 ON THE KERNELS (ranks share this box's GPU over gloo): K0 with the degree pass
    sharded + the all-reduce of the degree vector, the shard's rows fitted by the HIP kernels, all-gather, user-sharded
    predict - model and predictions bit-identical to the single-GPU fit on every rank (tools/dist_itemknn_check.py; the
    first shape has two column tiles, the second one)."""
    import os
    import subprocess
    import sys

    from conftest import free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RECPACK_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "tools", "dist_itemknn_check.py"), *map(str, shape)]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("model bit-identical: True") == 2 * nproc  # through the library's communicator and through torch.distributed
