"""The CPU oracle against the reference's outputs (tests/golden, made by oracle/gen_golden.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from, load_golden
from oracle import oracle_np as onp
from parity import tie_aware_topk, topk_vals_close

ITEMKNN_CASES = [
    "ref_itemknn_data",
    "ref_itemknn_empty_col",
    "ref_itemknn_x_in",
    "ref_itemknn_larger_matrix",
    "syn_itemknn_2000x500",
]


@pytest.mark.parametrize("name", ITEMKNN_CASES)
def test_itemknn_fit_matches_reference(name):
    g = load_golden(name)
    X = csr_from(g, "X")
    cand = csr_from(g, "cos")
    for K in g["Ks"]:
        K = int(K)
        ref = csr_from(g, f"S{K}")
        out = onp.itemknn_fit(X, K)
        bad, exact = tie_aware_topk(cand, ref, out["idx"], out["nnz"], eps=1e-9)
        assert not bad, (name, K, bad[:5])
        topk_vals_close(ref, out["idx"], out["val"], out["nnz"], rtol=1e-11)
        # no explicit zeros, no diagonal
        for r in range(out["idx"].shape[0]):
            assert r not in out["idx"][r, : out["nnz"][r]]
        assert (np.diff(ref.indptr) == out["nnz"]).all()


def test_itemknn_analytic_values():
    # recpack/tests/test_algorithms/test_nearest_neighbour.py:44-56
    g = load_golden("ref_itemknn_data")
    out = onp.itemknn_fit(csr_from(g, "X"), 2)
    S = onp.topk_to_csr(out["idx"], out["val"], out["nnz"], 3).toarray()
    e = 2 / np.sqrt(6)
    np.testing.assert_almost_equal(S, [[0, 0.5, e], [0.5, 0, e], [e, e, 0]])


def test_itemknn_normalize_sim():
    g = load_golden("ref_itemknn_data_normsim")
    out = onp.itemknn_fit(csr_from(g, "X"), 2)
    S = onp.topk_to_csr(out["idx"], out["val"], out["nnz"], 3, normalize_l1=True)
    np.testing.assert_allclose(S.toarray(), csr_from(g, "S2").toarray(), rtol=1e-12)
    np.testing.assert_allclose(np.asarray(S.sum(axis=1)).ravel(), 1.0)


def test_itemknn_big_thresholds():
    g = load_golden("syn_itemknn_10000x2000")
    X = csr_from(g, "X")
    ref = csr_from(g, "S100")
    out = onp.itemknn_fit(X, 100)
    assert (np.diff(ref.indptr) == out["nnz"]).all()
    # candidates recomputed by the oracle's closed form (|rel diff| to the reference <= 6e-13)
    G = onp.gram_counts(onp.binarize(X))
    c = onp.item_degrees(onp.binarize(X)).astype(np.float64)
    r = np.repeat(np.arange(G.shape[0]), np.diff(G.indptr))
    vals = G.data / np.sqrt(c[r] * c[G.indices])
    vals[r == G.indices] = 0
    cand = sp.csr_matrix((vals, G.indices, G.indptr), shape=G.shape)
    bad, exact = tie_aware_topk(cand, ref, out["idx"], out["nnz"], eps=1e-9)
    assert not bad, bad[:5]
    topk_vals_close(ref, out["idx"], out["val"], out["nnz"], rtol=1e-11)


@pytest.mark.parametrize("name", ["ref_itemknn_data", "ref_itemknn_x_in", "ref_itemknn_larger_matrix", "syn_itemknn_2000x500"])
def test_itemknn_predict_matches_reference(name):
    g = load_golden(name)
    n_items = int(g["X_shape"][1])
    for K in g["Ks"]:
        K = int(K)
        S = csr_from(g, f"S{K}")  # predict from the reference's own similarity matrix
        n = 0
        while f"Pin{n}_indptr" in g:
            out = onp.predict_full(csr_from(g, f"Pin{n}"), S)
            ref = csr_from(g, f"Pout{K}_{n}")
            np.testing.assert_allclose(out.toarray(), ref.toarray(), rtol=1e-12, atol=1e-15)
            n += 1
        assert n > 0 or name == "ref_itemknn_empty_col"


def test_topk_data_knn():
    # recpack/tests/test_util.py:49-62
    g = load_golden("ref_topk_data_knn")
    X = csr_from(g, "X")
    idx, val, nnz = onp.topk_rows(X, 2)
    out = onp.topk_to_csr(idx, val, nnz, X.shape[1])
    np.testing.assert_almost_equal(out.toarray(), csr_from(g, "top2").toarray())


@pytest.mark.parametrize("name", ["ref_ease_data", "ref_ease_larger_matrix", "syn_ease_2000x500"])
def test_ease_matches_reference(name):
    g = load_golden(name)
    X = csr_from(g, "X")
    for n, (l2, alpha) in enumerate(zip(g["l2"], g["alpha"])):
        B = onp.ease_fit(X, float(l2), float(alpha))
        ref = g[f"B{n}"]
        assert np.isfinite(ref).all()
        np.testing.assert_allclose(B, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
        if "Pin_indptr" in g:
            out = onp.predict_full(csr_from(g, "Pin"), B).toarray()
            np.testing.assert_allclose(out, g[f"Pout{n}"], rtol=1e-9, atol=1e-12 * np.abs(ref).max())


def test_ease_reference_known_values():
    # values the reference returns on recpack/tests/test_algorithms/test_ease.py:16-35
    g = load_golden("ref_ease_data")
    B = onp.ease_fit(csr_from(g, "X"), 0.03)
    np.testing.assert_allclose(B[0], [0, 0.12453300125572914, 0.9901932783260373], rtol=1e-10)
    B200 = onp.ease_fit(csr_from(g, "X"), 200.0)
    np.testing.assert_allclose(B200[1], [0.00487769187620418, 0, 0.00487769187620418], rtol=1e-10)
    # alpha relations of test_ease.py:66-87 (c(0) = 4)
    B1 = onp.ease_fit(csr_from(g, "X"), 0.03, 1)
    B2 = onp.ease_fit(csr_from(g, "X"), 0.03, 2)
    np.testing.assert_almost_equal(B1[1, 0], B[1, 0] / 4)
    np.testing.assert_almost_equal(B1[2, 1], B[2, 1])
    np.testing.assert_almost_equal(B1[1, 0] / 4, B2[1, 0])


def test_ease_prune():
    g = load_golden("ref_ease_prune")
    out = onp.ease_prune(g["before"], float(g["density"]))
    np.testing.assert_array_equal(out, g["after"])


CONDPROB_CASES = ["ref_itemknn_data_condprob", "ref_itemknn_data_condprob_pd1.0", "ref_itemknn_data_condprob_pd0.2",
                  "ref_itemknn_data_condprob_pd0.5", "syn_itemknn_condprob_2000x500", "syn_itemknn_condprob0_2000x500"]


@pytest.mark.parametrize("name", CONDPROB_CASES)
def test_itemknn_conditional_probability_matches_reference(name):
    # recpack/algorithms/nearest_neighbour.py:22-66 through ItemKNN(similarity="conditional_probability")
    g = load_golden(name)
    X = csr_from(g, "X")
    cand = csr_from(g, "cos")
    pd = float(g["pop_discount"])
    for K in g["Ks"]:
        K = int(K)
        ref = csr_from(g, f"S{K}")
        out = onp.itemknn_fit(X, K, similarity="conditional_probability", pop_discount=pd)
        bad, exact = tie_aware_topk(cand, ref, out["idx"], out["nnz"], eps=1e-9)
        assert not bad, (name, K, bad[:5])
        topk_vals_close(ref, out["idx"], out["val"], out["nnz"], rtol=1e-11)


def test_conditional_probability_analytic():
    # recpack/tests/test_algorithms/test_nearest_neighbour.py:127-152
    g = load_golden("ref_itemknn_data_condprob")
    out = onp.itemknn_fit(csr_from(g, "X"), 2, similarity="conditional_probability")
    S = onp.topk_to_csr(out["idx"], out["val"], out["nnz"], 3).toarray()
    np.testing.assert_almost_equal(S, [[0, 1 / 2, 2 / 2], [1 / 2, 0, 2 / 2], [2 / 3, 2 / 3, 0]])


NORMX_CASES = ["ref_itemknn_data_normX", "syn_itemknn_normX_2000x500", "syn_itemknn_normX_condprob_2000x500"]


def normx_candidates(X, similarity, pd):
    """Every candidate of the normalize_X similarity from the oracle's closed form (for goldens without ``cos``)."""
    return onp.itemknn_fit_normalized(X, None, similarity=similarity, pop_discount=pd)


@pytest.mark.parametrize("name", NORMX_CASES + ["syn_itemknn_normX_6000x14000"])
def test_itemknn_normalize_X_matches_reference(name):
    # nearest_neighbour.py:207-210 (Normalizer l1 on the binarised rows) followed by :69-84 or :22-66
    g = load_golden(name)
    X = csr_from(g, "X")
    sim, pd = str(g["similarity"]), float(g["pop_discount"])
    cand = csr_from(g, "cos") if "cos_indptr" in g else normx_candidates(X, sim, pd)
    for K in g["Ks"]:
        K = int(K)
        ref = csr_from(g, f"S{K}")
        out = onp.itemknn_fit_normalized(X, K, similarity=sim, pop_discount=pd)
        assert (np.diff(ref.indptr) == out["nnz"]).all()
        bad, exact = tie_aware_topk(cand, ref, out["idx"], out["nnz"], eps=1e-9)
        assert not bad, (name, K, bad[:5])
        topk_vals_close(ref, out["idx"], out["val"], out["nnz"], rtol=1e-11)


def test_itemknn_normalize_X_analytic():
    # recpack/tests/test_algorithms/test_nearest_neighbour.py:73-107
    import math

    g = load_golden("ref_itemknn_data_normX")
    out = onp.itemknn_fit_normalized(csr_from(g, "X"), 2)
    S = onp.topk_to_csr(out["idx"], out["val"], out["nnz"], 3).toarray()
    a = 1 / (3 * 3)
    b = 1 / (3 * 3) + 1 / (2 * 2)
    d = math.sqrt((1 / 2) ** 2 + (1 / 3) ** 2)
    f = math.sqrt(2 * (1 / 2) ** 2 + (1 / 3) ** 2)
    np.testing.assert_almost_equal(S, [[0, a / (d * d), b / (d * f)], [a / (d * d), 0, b / (d * f)], [b / (d * f), b / (d * f), 0]])


@pytest.mark.parametrize("name", ["ref_ease_xy_doc", "syn_ease_xy_1500x300"])
def test_ease_xy_matches_reference(name):
    # recpack/algorithms/experimental/ease_xy.py:61-96
    g = load_golden(name)
    X, Y = csr_from(g, "X"), csr_from(g, "Y")
    for n, (l2, alpha) in enumerate(zip(g["l2"], g["alpha"])):
        B = onp.ease_xy_fit(X, Y, float(l2), float(alpha))
        ref = g[f"B{n}"]
        assert np.abs(B - ref).max() <= 1e-10 * np.abs(ref).max()
        out = onp.predict_full(X[:50], sp.csr_matrix(B)).toarray()
        assert np.abs(out - g[f"Pout{n}"]).max() <= 1e-9 * max(1e-300, np.abs(g[f"Pout{n}"]).max())


@pytest.mark.parametrize("name", ["ref_itemknn_data", "ref_itemknn_larger_matrix", "syn_itemknn_2000x500"])
def test_reference_shaped_fit_matches_reference(name):
    """The restatement bench.py times as "reference shaped" (sklearn cosine + per-row argpartition loop) against the
    same goldens, tie-aware (its tie order is numpy's, like the reference's)."""
    g = load_golden(name)
    X = csr_from(g, "X")
    cand = csr_from(g, "cos")
    K = int(g["Ks"][-1])
    ref = csr_from(g, f"S{K}")
    S = onp.itemknn_fit_reference_shaped(X, K)
    S.eliminate_zeros()
    S.sort_indices()
    n = np.diff(S.indptr)
    idx = np.full((S.shape[0], max(1, n.max())), -1, np.int64)
    for r in range(S.shape[0]):
        idx[r, : n[r]] = S.indices[S.indptr[r]: S.indptr[r + 1]]
    bad, _ = tie_aware_topk(cand, ref, idx, n, eps=1e-9)
    assert not bad, (name, bad[:5])
    rows = np.array([0, X.shape[1] // 2, X.shape[1] - 1])
    sub = onp.itemknn_fit_reference_shaped(X, K, rows=rows)
    sub.eliminate_zeros()
    assert abs(sub - S[rows]).max() == 0


def test_reference_shaped_predict_topk():
    g = load_golden("syn_itemknn_2000x500")
    X = csr_from(g, "X")
    out = onp.itemknn_fit(X, 50)
    S = onp.topk_to_csr(out["idx"], out["val"], out["nnz"], X.shape[1])
    scores = onp.predict_full(X[:300], S)
    got = onp.predict_topk_reference_shaped(X[:300], S, 20)
    got.sort_indices()
    ridx, rval, rnnz = onp.topk_rows(scores, 20)
    n = np.diff(got.indptr)
    assert (n == rnnz).all()
    idx = np.full((300, 20), -1, np.int64)
    for r in range(300):
        idx[r, : n[r]] = got.indices[got.indptr[r]: got.indptr[r + 1]]
    bad, _ = tie_aware_topk(scores, onp.topk_to_csr(ridx, rval, rnnz, X.shape[1]), idx, n, eps=1e-12)
    assert not bad
