"""Pipeline-level parity on C1 (BASELINE.json configs[0]): the reference's MovieLens-100K sample split with
WeakGeneralization(0.75, seed=42), ItemKNN(K=50) / EASE(l2=20), history removal, an ExcludeItems post-filter and
NDCGK / RecallK / CalibratedRecallK / CoverageK at K = 10 (golden: oracle/gen_golden.py::pipeline_case).

CPU part: the oracle's metric restatement and the product's top-K metric harness against the reference's values.
GPU part: fit -> fused predict_topk (history removal + item filter + top-K in the kernel) -> metrics, against the same
goldens.  NDCG depends on the order of equal scores inside the top-K, which the reference leaves to numpy's
argpartition: it is compared at 5e-4, coverage within two items (ties at rank K), recall exactly.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from, load_golden
from oracle import oracle_np as onp

NAMES = ["ndcg", "recall", "calibrated_recall", "coverage"]
K = 10


@pytest.fixture(scope="module")
def g():
    return load_golden("c1_pipeline_ml100k")


def keep_mask(g, n_items):
    m = np.ones(n_items, dtype=bool)
    m[g["excluded_items"]] = False
    return m


@pytest.mark.parametrize("algo", ["ItemKNN", "EASE"])
def test_oracle_metrics_match_reference(g, algo):
    y_true = csr_from(g, "test_out")
    scores = csr_from(g, f"scores_{algo}")
    for tag, sc in (("", scores), ("_filtered", onp.pipeline_scores(scores, item_keep=keep_mask(g, scores.shape[1])))):
        ref = dict(zip(NAMES, g[f"metrics_{algo}{tag}"]))
        got = onp.metrics_topk(y_true, sc, K)
        assert abs(got["ndcg"] - ref["ndcg"]) <= 5e-4
        for k in ("recall", "calibrated_recall"):
            assert abs(got[k] - ref[k]) <= 1e-12, (algo, tag, k)
        assert abs(got["coverage"] - ref["coverage"]) <= 2.0 / scores.shape[1]  # ties at rank K pick other items


@pytest.mark.parametrize("algo", ["ItemKNN", "EASE"])
def test_topk_metric_harness_equals_oracle(g, algo):
    """recpack_amd.metrics works on the K best items per user only and must give what the full matrix gives."""
    from recpack_amd import metrics as M

    y_true = csr_from(g, "test_out")
    scores = csr_from(g, f"scores_{algo}")
    ridx, rval, rnnz = onp.topk_rows(scores, K)
    topk = onp.topk_to_csr(ridx, rval, rnnz, scores.shape[1])
    want = onp.metrics_topk(y_true, scores, K)
    got = {"ndcg": M.ndcg_k(y_true, topk, K), "recall": M.recall_k(y_true, topk, K),
           "calibrated_recall": M.calibrated_recall_k(y_true, topk, K), "coverage": M.coverage_k(y_true, topk, K)}
    for k in NAMES:
        assert abs(got[k] - want[k]) <= 1e-12, (k, got[k], want[k])
    # the reference's checks: shapes must agree, users without true interactions do not count
    with pytest.raises(AssertionError):
        M.ndcg_k(y_true[:10], topk, K)
    y2 = y_true.tolil()
    y2[0] = 0
    assert M.recall_k(y2.tocsr(), topk, K) != got["recall"] or y_true[0].nnz == 0


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["ItemKNN", "EASE"])
def test_gpu_pipeline_step_matches_reference(g, algo):
    import warnings

    import recpack_amd
    from recpack_amd import metrics as M

    train, tin, y_true = csr_from(g, "train"), csr_from(g, "test_in"), csr_from(g, "test_out")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = recpack_amd.ItemKNN(K=50).fit(train) if algo == "ItemKNN" else recpack_amd.EASE(l2=20).fit(train)
        # 1. the reference's own flow on the drop-in class: predict, remove history, metrics from the full matrix
        full = onp.pipeline_scores(model.predict(tin), history=tin)
        # 2. the fused step
        top = model.predict_topk(tin, K, exclude_history=True)
        top_f = model.predict_topk(tin, K, exclude_history=True, exclude_items=g["excluded_items"])
    ref_scores = csr_from(g, f"scores_{algo}")
    if algo == "EASE":  # (ItemKNN: ties at the 50-th neighbour are resolved differently, so single scores may differ)
        assert abs(full - ref_scores).max() <= 1e-4 * abs(ref_scores).max()
    # fused == unfused: the kernel's history removal + filter + top-K give exactly the metrics of the drop-in class's
    # own full score matrix pushed through the reference's steps
    keep = keep_mask(g, train.shape[1])
    for t, sc in ((top, full), (top_f, onp.pipeline_scores(full, item_keep=keep))):
        want = onp.metrics_topk(y_true, sc, K)
        got = {"ndcg": M.ndcg_k(y_true, t, K), "recall": M.recall_k(y_true, t, K),
               "calibrated_recall": M.calibrated_recall_k(y_true, t, K), "coverage": M.coverage_k(y_true, t, K)}
        for k in NAMES:
            assert abs(got[k] - want[k]) <= 1e-12, (algo, k, got[k], want[k])
    # against the reference's numbers.  EASE has no ties: tight.  ItemKNN on this small sample is full of equal
    # similarities (co-occurrence counts of 1-3): which neighbours survive at rank 50 and which items at rank 10 is the
    # reference's numpy-version-dependent choice, worth a few 1e-4 on every metric.
    tol = {"ItemKNN": (2e-3, 2e-3, 4.0 / train.shape[1]), "EASE": (1e-6, 1e-6, 1e-9)}[algo]
    for tag, t in (("", top), ("_filtered", top_f)):
        ref = dict(zip(NAMES, g[f"metrics_{algo}{tag}"]))
        got = {"ndcg": M.ndcg_k(y_true, t, K), "recall": M.recall_k(y_true, t, K),
               "calibrated_recall": M.calibrated_recall_k(y_true, t, K), "coverage": M.coverage_k(y_true, t, K)}
        assert abs(got["ndcg"] - ref["ndcg"]) <= tol[0], (tag, got, ref)
        assert abs(got["recall"] - ref["recall"]) <= tol[1], (tag, got, ref)
        assert abs(got["calibrated_recall"] - ref["calibrated_recall"]) <= tol[1]
        assert abs(got["coverage"] - ref["coverage"]) <= tol[2]
    assert not np.isin(top_f.indices, g["excluded_items"]).any()
    assert top_f.multiply(tin).nnz == 0 and top.multiply(tin).nnz == 0
    # select_items is the complement filter; the argument checks are the reference's (filters.py:69-70)
    keep = np.setdiff1d(np.arange(train.shape[1]), g["excluded_items"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top_s = model.predict_topk(tin, K, exclude_history=True, select_items=keep)
        with pytest.raises(ValueError):
            model.predict_topk(tin, K, exclude_items=np.array([], dtype=np.int64))
    assert (top_s != top_f).nnz == 0
