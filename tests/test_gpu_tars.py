"""GPU parity: the time-aware ItemKNN family (recpack/algorithms/time_aware_item_knn/) through the C ABI.

Fit: float64 Gram of the weighted matrix by LDS atomics - the same terms as scipy's product in another order.  Bar:
tie-aware top-K sets with eps = 1e-9 against the reference's goldens (candidates from the oracle, which the CPU suite
pins to the same goldens), common values within 1e-11 relative.
Predict: weighted histories times f32 similarities as exact fixed-point sums: 2e-7 of the largest score.
"""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from, load_golden
from oracle import oracle_np as onp
from parity import tie_aware_topk

pytestmark = pytest.mark.gpu

DECAYS = {"exponential": 0.004, "convex": 0.995, "concave": 0.3, "log": 5.0, "linear": 0.8, "inverse": 0.5}
SIMS = ("cosine", "conditional_probability", "pearson")
VAL_RTOL = 1e-11
PRED_TOL = 2e-7


@pytest.fixture(scope="module")
def hip():
    from recpack_amd import _lib, get_context

    return _lib, get_context(0)


def events(g):
    from recpack_amd.time_aware import TimedInteractions

    return TimedInteractions(g["uid"], g["iid"], g["ts"], tuple(int(x) for x in g["shape"]))


def dense_topk(S: sp.csr_matrix):
    S = sp.csr_matrix(S)
    S.sort_indices()
    n, k = S.shape[0], max(1, int(np.diff(S.indptr).max()))
    idx = np.full((n, k), -1, np.int32)
    val = np.zeros((n, k))
    nnz = np.diff(S.indptr).astype(np.int32)
    for r in range(n):
        idx[r, : nnz[r]] = S.indices[S.indptr[r]: S.indptr[r + 1]]
        val[r, : nnz[r]] = S.data[S.indptr[r]: S.indptr[r + 1]]
    return idx, val, nnz


def check_similarity(algo, W_fit, S_ref):
    """algo.similarity_matrix_ against the reference's: tie-aware sets, equal values where both selected an entry."""
    S = sp.csr_matrix(algo.similarity_matrix_)
    S.sort_indices()
    full = onp.tars_similarity(W_fit, algo.similarity)
    idx, _, nnz = dense_topk(S)
    bad, exact = tie_aware_topk(full, S_ref, idx, nnz, positive_only=False)
    assert not bad, bad[:5]
    a, b = S.multiply(S_ref != 0), S_ref.multiply(S != 0)
    assert abs(a - b).max() <= VAL_RTOL * max(1e-300, abs(S_ref).max())
    return exact == S.shape[0]


def check_predict(algo, X, W_pred, P_ref=None, same_S=False):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P = sp.csr_matrix(algo.predict(X))
    S32 = sp.csr_matrix(algo.similarity_matrix_).astype(np.float32).astype(np.float64)
    want = onp.tars_predict(W_pred, S32)
    scale = max(1e-300, abs(want).max())
    assert abs(P - want).max() <= PRED_TOL * scale
    if same_S and P_ref is not None:  # no tie was broken differently: the reference's own scores
        assert abs(P - P_ref).max() <= 1e-6 * max(1e-300, abs(P_ref).max())
    return P


def run_case(g, algo, tag, wtag=None):
    X = events(g)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        algo.fit(X)
    W_fit = algo._add_decay_to_fit_matrix(X)
    if wtag is not None:  # the host weights themselves against the reference's
        assert abs(W_fit - csr_from(g, f"Wfit_{wtag}")).max() <= 1e-15
    same = check_similarity(algo, W_fit, csr_from(g, f"S_{tag}"))
    P_ref = csr_from(g, f"P_{tag}") if f"P_{tag}_indptr" in g else None
    check_predict(algo, X, algo._add_decay_to_predict_matrix(X), P_ref, same)


@pytest.mark.parametrize("sim", SIMS)
def test_tars_reference_fixture(hip, sim):
    """recpack/tests/conftest.py:68-77 (``mat``) with the parameters of test_time_aware_item_knn_base.py:14-17."""
    from recpack_amd.time_aware import TARSItemKNN

    g = load_golden("ref_tars_mat")
    run_case(g, TARSItemKNN(K=2, fit_decay=0.5, predict_decay=0.5, similarity=sim), sim)


@pytest.mark.parametrize("sim", SIMS)
@pytest.mark.parametrize("fn", list(DECAYS))
def test_tars_decay_functions(hip, fn, sim):
    from recpack_amd.time_aware import TARSItemKNN

    g = load_golden("syn_tars_400x150")
    K, interval = int(g["K"][0]), int(g["decay_interval"][0])
    algo = TARSItemKNN(K=K, fit_decay=DECAYS[fn], predict_decay=DECAYS[fn], decay_interval=interval, similarity=sim, decay_function=fn)
    run_case(g, algo, f"{fn}_{sim}", wtag=f"{fn}_cosine")


def test_tars_variants(hip):
    from recpack_amd import time_aware as ta

    g = load_golden("syn_tars_400x150")
    K = int(g["K"][0])
    run_case(g, ta.TARSItemKNNDing(K=K, predict_decay=1 / (24 * 3600), similarity="conditional_probability"), "ding")
    run_case(g, ta.TARSItemKNNLiu(K=K, fit_decay=1 / (24 * 3600), predict_decay=1 / (7 * 24 * 3600)), "liu")
    run_case(g, ta.TARSItemKNNVaz(K=K, fit_decay=1 / (24 * 3600), predict_decay=1 / (24 * 3600)), "vaz")
    run_case(g, ta.TARSItemKNNLiu2012(K=K, decay=3.0), "liu2012", wtag="liu2012")
    for w in (3, 5):
        for sim in ("cosine", "pearson"):
            run_case(g, ta.TARSItemKNNLee(K=K, w=w, similarity=sim), f"lee{w}_{sim}", wtag=f"lee{w}_cosine")


@pytest.mark.parametrize("sim", SIMS)
def test_tars_more_items_than_one_tile(hip, sim):
    """13 000 items: two column tiles of the float64 accumulator, per-tile candidates and the merge (signed for pearson)."""
    from recpack_amd.time_aware import TARSItemKNN

    g = load_golden("syn_tars_3000x13000")
    K = int(g["K"][0])
    algo = TARSItemKNN(K=K, fit_decay=1 / (24 * 3600), predict_decay=1 / (24 * 3600), similarity=sim)
    X = events(g)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        algo.fit(X)
    check_similarity(algo, algo._add_decay_to_fit_matrix(X), csr_from(g, f"S_{sim}"))
    check_predict(algo, X, algo._add_decay_to_predict_matrix(X))


def test_weighted_predict_from_reference_similarity(hip):
    """The reference's similarity matrix loaded as a model, the reference's weighted history: its own scores, and the fused
    top-K of them with and without history removal."""
    _lib, ctx = hip
    g = load_golden("syn_tars_400x150")
    for tag, wtag in (("exponential_cosine", "exponential_cosine"), ("log_pearson", "log_cosine"), ("liu2012", "liu2012")):
        S_ref, P_ref, Wp = csr_from(g, f"S_{tag}"), csr_from(g, f"P_{tag}"), csr_from(g, f"Wpred_{wtag}")
        Wp.eliminate_zeros()
        Wp.sort_indices()
        P_ref.eliminate_zeros()
        model = _lib.Model.from_topk_host(ctx, *dense_topk(S_ref))
        dW = _lib.DeviceCSR.from_host_valued(ctx, Wp)
        ip, ind, dat = _lib.predict_full(ctx, model, dW)
        P = sp.csr_matrix((dat, ind, ip), shape=P_ref.shape)
        scale = abs(P_ref).max()
        assert abs(P - P_ref).max() <= PRED_TOL * scale
        for excl in (False, True):
            Kp = 7
            idx, val, nnz = _lib.predict_topk(ctx, model, dW, Kp, exclude_history=excl)
            cand = P_ref.copy()
            if excl:
                cand = cand - cand.multiply(Wp != 0)
                cand.eliminate_zeros()
            ridx, rval, rnnz = onp.topk_rows(cand, Kp)
            bad, _ = tie_aware_topk(cand, onp.topk_to_csr(ridx, rval, rnnz, cand.shape[1]), idx, nnz, eps=1e-6, positive_only=False)
            assert not bad, (tag, excl, bad[:5])
            got = onp.topk_to_csr(idx, val.astype(np.float64), nnz, cand.shape[1])
            assert abs(got - cand.multiply(got != 0)).max() <= PRED_TOL * scale
        # detaching the values gives the binary product again
        dW.set_values(None)
        ip, ind, dat = _lib.predict_full(ctx, model, dW)
        Pb = sp.csr_matrix((dat, ind, ip), shape=P_ref.shape)
        want = onp.tars_predict((Wp != 0).astype(np.float64), S_ref.astype(np.float32).astype(np.float64))
        assert abs(Pb - want).max() <= PRED_TOL * max(1e-300, abs(want).max())
        dW.close()
        model.close()


def test_valued_api_errors(hip):
    from recpack_amd.time_aware import TARSItemKNN, TimedInteractions

    _lib, ctx = hip
    X = sp.random(50, 20, density=0.2, random_state=np.random.default_rng(0), format="csr")
    X.sort_indices()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    with pytest.raises(_lib.RecpackHipError, match="carries no values"):
        _lib.itemknn_fit_values(ctx, dX, 5)
    with pytest.raises(ValueError):
        dX.set_values(np.ones(3))
    bad = X.data.copy()
    bad[0] = np.nan
    with pytest.raises(_lib.RecpackHipError, match="not finite"):
        dX.set_values(bad)
    dX.set_values(X.data)
    dense = _lib.ease_fit(ctx, dX, 10.0)  # EASE binarises: a dense model refuses a valued prediction matrix
    with pytest.raises(_lib.RecpackHipError, match="valued prediction matrix"):
        _lib.predict_topk(ctx, dense, dX, 5, False)
    with pytest.raises(ValueError, match="Pearson similarity can not be computed on a binary matrix"):
        TARSItemKNN(K=3, fit_decay=0, similarity="pearson").fit(TimedInteractions([0, 0, 1], [0, 1, 1], [5, 6, 7], (2, 2)))
    empty = TARSItemKNN(K=3).fit(TimedInteractions([], [], [], (4, 6)))
    assert empty.similarity_matrix_.nnz == 0


COOC_DECAYS = {"exponential": 0.02, "convex": 0.98, "concave": 0.3, "log": 5.0, "linear": 0.8, "inverse": 0.5}


def check_cooc(algo, g, tag, with_predict=False):
    X = events(g)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        algo.fit(X)
    S_ref = csr_from(g, f"S_{tag}")
    S = sp.csr_matrix(algo.similarity_matrix_)
    S.sort_indices()
    full = onp.tars_cooc_similarity(X.last_timestamps_matrix, algo.similarity, algo.decay_function, algo.fit_decay, algo.decay_interval)
    idx, _, nnz = dense_topk(S)
    bad, exact = tie_aware_topk(full, S_ref, idx, nnz)
    assert not bad, (tag, bad[:5])
    assert abs(S.multiply(S_ref != 0) - S_ref.multiply(S != 0)).max() <= VAL_RTOL * max(1e-300, abs(S_ref).max())
    if with_predict:  # binary histories (predict_decay = 0)
        Wp = algo._add_decay_to_predict_matrix(X)
        assert np.all(Wp.data == 1.0)
        check_predict(algo, X, Wp, csr_from(g, f"P_{tag}"), exact == S.shape[0])


def test_tars_cooc_reference_fixture(hip):
    """recpack/tests/conftest.py:80-89 and the expected arrays of test_time_aware_item_knn_base.py:112-150."""
    from recpack_amd.time_aware import TARSItemKNNCoocDistance

    g = load_golden("ref_tars_cooc_mat")
    e = np.exp
    want = np.array([[0, e(-0.5) * 3, 0, 0, 0], [e(-0.5) * 3, 0, 0, 0, 0], [0, 0, 0, e(-1.5), e(-1.0)], [0, 0, e(-1.5), 0, 0],
                     [0, 0, e(-1.0), 0, 0]])
    algo = TARSItemKNNCoocDistance(K=2, fit_decay=0.5, similarity="cooc")
    check_cooc(algo, g, "cooc", with_predict=True)
    np.testing.assert_array_almost_equal(algo.similarity_matrix_.toarray(), want)
    check_cooc(TARSItemKNNCoocDistance(K=2, fit_decay=0.5, similarity="conditional_probability"), g, "conditional_probability", with_predict=True)


@pytest.mark.parametrize("fn", list(COOC_DECAYS))
def test_tars_cooc_decay_functions(hip, fn):
    from recpack_amd.time_aware import TARSItemKNNCoocDistance

    g = load_golden("syn_tars_cooc_300x120")
    K, interval = int(g["K"][0]), int(g["decay_interval"][0])
    for sim in ("cooc", "conditional_probability"):
        check_cooc(TARSItemKNNCoocDistance(K=K, fit_decay=COOC_DECAYS[fn], decay_interval=interval, similarity=sim, decay_function=fn), g,
                   f"{fn}_{sim}")


def test_tars_cooc_no_decay_and_xia(hip):
    from recpack_amd.time_aware import TARSItemKNNCoocDistance, TARSItemKNNXia

    g = load_golden("syn_tars_cooc_300x120")
    K, interval = int(g["K"][0]), int(g["decay_interval"][0])
    check_cooc(TARSItemKNNCoocDistance(K=K, fit_decay=0, decay_interval=interval, similarity="cooc"), g, "none_cooc")
    check_cooc(TARSItemKNNXia(K=K, fit_decay=0.6, decay_function="concave", decay_interval=24 * 3600), g, "xia", with_predict=True)


def test_tars_hermann(hip):
    """hermann_2010.py: the reference's fixture and the seeded log (two time units), then predict on binary histories."""
    from recpack_amd.time_aware import TARSItemKNNHermann

    def check(g, algo, tag, P_tag=None):
        X = events(g)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            algo.fit(X)
        S_ref = csr_from(g, f"S_{tag}")
        S = sp.csr_matrix(algo.similarity_matrix_)
        S.sort_indices()
        full = onp.tars_hermann_similarity(X.last_timestamps_matrix, algo.decay_interval)
        idx, _, nnz = dense_topk(S)
        bad, exact = tie_aware_topk(full, S_ref, idx, nnz)
        assert not bad, (tag, bad[:5])
        assert abs(S.multiply(S_ref != 0) - S_ref.multiply(S != 0)).max() <= VAL_RTOL * abs(S_ref).max()
        if P_tag:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                P = sp.csr_matrix(algo.predict(X))
            want = onp.predict_full(X.binary_values, S.astype(np.float32).astype(np.float64))
            assert abs(P - want).max() <= PRED_TOL * abs(want).max()
            if exact == S.shape[0]:
                P_ref = csr_from(g, f"P_{P_tag}")
                assert abs(P - P_ref).max() <= 1e-6 * abs(P_ref).max()

    check(load_golden("ref_tars_cooc_mat"), TARSItemKNNHermann(K=2), "hermann")
    g = load_golden("syn_tars_cooc_300x120")
    K = int(g["K"][0])
    check(g, TARSItemKNNHermann(K=K, decay_interval=1), "hermann1")
    check(g, TARSItemKNNHermann(K=K, decay_interval=3600), "hermann3600", P_tag="hermann3600")


def same_rows(full, part, b, e, name):
    """Float64 sums by atomics differ in the last bits from run to run (and may flip an exact tie): same row sizes, the
    same selected values up to 1e-12, the same columns wherever the values are not tied."""
    (fi, fv, fn), (pi, pv, pn) = full, part
    np.testing.assert_array_equal(fn[b:e], pn, err_msg=name)
    np.testing.assert_allclose(np.sort(fv[b:e], axis=1), np.sort(pv, axis=1), rtol=1e-12, err_msg=name)
    assert e == b or (fi[b:e] == pi).mean() > 0.98, name


def test_valued_fits_shard_by_item_rows(hip):
    """SURVEY 8(e) for the time-aware fits: item rows are independent - any split of the rows gives the rows of the full fit,
    and the sharded driver (world size 1: RCCL all-gather of one part) assembles a predict-ready model."""
    import torch
    import torch.distributed as dist

    from recpack_amd import dist as rdist

    _lib, ctx = hip
    g = load_golden("syn_tars_cooc_300x120")
    T = sp.csr_matrix(events(g).last_timestamps_matrix, dtype=np.float64)
    T.eliminate_zeros()
    T.sort_indices()
    dT = _lib.DeviceCSR.from_host_valued(ctx, T)
    n = T.shape[1]
    fits = {
        "values": lambda b, e: _lib.itemknn_fit_values(ctx, dT, 8, row_begin=b, row_end=e),
        "timedist": lambda b, e: _lib.itemknn_fit_timedist(ctx, dT, 8, "exponential", 1e-6, float(T.data.max() - T.data.min()), row_begin=b, row_end=e),
        "hermann": lambda b, e: _lib.itemknn_fit_hermann(ctx, dT, 8, float(T.data.max() + 1), row_begin=b, row_end=e),
    }
    for name, fit in fits.items():
        full = fit(0, n).topk_download()
        for b, e in ((0, 37), (37, 38), (38, n), (50, 50)):
            part = fit(b, e).topk_download()
            same_rows(full, part, b, e, name)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29573", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m = rdist.fit_itemknn_sharded(ctx, dT, 8, [(0, n)], torch.device("cuda", 0), fit_fn=fits["hermann"])
        same_rows(fits["hermann"](0, n).topk_download(), m.topk_download(), 0, n, "sharded")
        idx, val, nnz = _lib.predict_topk(ctx, m, dT, 5, False)
        assert (nnz > 0).any()
    finally:
        dist.destroy_process_group()


def test_time_distance_fits_beyond_one_tile(hip):
    """13 000 items: light rows go through the hash-table kernel, the rest through column tiles and the merge; against the
    oracle (pinned to the reference by the CPU suite) for the time-distance and Hermann similarities."""
    from recpack_amd.time_aware import TARSItemKNNCoocDistance, TARSItemKNNHermann

    g = load_golden("syn_tars_3000x13000")
    X = events(g)
    K = 6
    T = X.last_timestamps_matrix
    cases = [
        (TARSItemKNNCoocDistance(K=K, fit_decay=0.3, decay_interval=24 * 3600, similarity="conditional_probability", decay_function="concave"),
         lambda: onp.tars_cooc_fit(T, K, "conditional_probability", "concave", 0.3, 24 * 3600)),
        (TARSItemKNNHermann(K=K, decay_interval=3600), lambda: onp.tars_hermann_fit(T, K, 3600)),
    ]
    for algo, oracle in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            algo.fit(X)
        r = oracle()
        S_ref = onp.topk_to_csr(r["idx"], r["val"], r["nnz"], T.shape[1])
        S = sp.csr_matrix(algo.similarity_matrix_)
        S.sort_indices()
        idx, _, nnz = dense_topk(S)
        bad, _ = tie_aware_topk(r["full"], S_ref, idx, nnz)
        assert not bad, (algo.name, bad[:5])
        assert abs(S.multiply(S_ref != 0) - S_ref.multiply(S != 0)).max() <= VAL_RTOL * abs(S_ref).max()
