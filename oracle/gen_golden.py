"""Generate tests/golden/*.npz by IMPORTING the reference (LienM/recpack v0.3.6) from /root/reference.

Run in the build container only (the reference does not travel to the GPU box):

    python oracle/gen_golden.py

Every file holds inputs (CSR arrays) and the outputs the reference computed for them:
``ItemKNN(...).fit(X).similarity_matrix_``, ``EASE(...).fit(X).similarity_matrix_`` and ``predict``.
The fixtures named ``ref_*`` re-create the reference's own test fixtures (values written out here as
data; file:line given per case), the ``syn_*`` ones are seeded synthetic power-law matrices.
"""
import os
import sys
import warnings

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

if not hasattr(np, "product"):  # recpack/algorithms/ease.py:105 uses the alias removed in numpy 2
    np.product = np.prod

from recpack.algorithms import EASE, ItemKNN  # noqa: E402
from recpack.util import get_top_K_values  # noqa: E402

from recpack_amd.synth import powerlaw_csr  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
warnings.simplefilter("ignore")


def csr_parts(prefix, M):
    M = sp.csr_matrix(M)
    return {
        f"{prefix}_indptr": M.indptr.astype(np.int64),
        f"{prefix}_indices": M.indices.astype(np.int32),
        f"{prefix}_data": M.data.astype(np.float64),
        f"{prefix}_shape": np.asarray(M.shape, dtype=np.int64),
    }


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def cosine_full(X):
    from recpack.algorithms.nearest_neighbour import compute_cosine_similarity
    from recpack.matrix import to_csr_matrix

    S = compute_cosine_similarity(to_csr_matrix(X, binary=True))
    S = sp.csr_matrix(S)
    S.sort_indices()
    return S


def condprob_full(X, pop_discount):
    from recpack.algorithms.nearest_neighbour import compute_conditional_probability
    from recpack.matrix import to_csr_matrix

    S = sp.csr_matrix(compute_conditional_probability(to_csr_matrix(X, binary=True), pop_discount or 0))
    S.sort_indices()
    return S


def itemknn_case(name, X, Ks, predict_in=None, **kw):
    arrays = dict(csr_parts("X", X))
    arrays["Ks"] = np.asarray(Ks, dtype=np.int64)
    if kw.get("similarity") == "conditional_probability":
        arrays.update(csr_parts("cos", condprob_full(X, kw.get("pop_discount"))))  # all candidates of this similarity
        arrays["pop_discount"] = np.asarray(kw.get("pop_discount") or 0.0)
    else:
        arrays.update(csr_parts("cos", cosine_full(X)))  # all candidates incl. explicit diagonal zeros
    for K in Ks:
        algo = ItemKNN(K=K, **kw).fit(X)
        S = sp.csr_matrix(algo.similarity_matrix_)
        S.sort_indices()
        arrays.update(csr_parts(f"S{K}", S))
        if predict_in is not None:
            for n, P in enumerate(predict_in):
                arrays.update(csr_parts(f"Pin{n}", P))
                out = algo.predict(P)
                out.sort_indices()
                arrays.update(csr_parts(f"Pout{K}_{n}", out))
    save(name, **arrays)


def ease_case(name, X, params, predict_in=None):
    arrays = dict(csr_parts("X", X))
    arrays["l2"] = np.asarray([p[0] for p in params], dtype=np.float64)
    arrays["alpha"] = np.asarray([p[1] for p in params], dtype=np.float64)
    for n, (l2, alpha) in enumerate(params):
        algo = EASE(l2=l2, alpha=alpha).fit(X)
        arrays[f"B{n}"] = algo.similarity_matrix_.toarray()
        if predict_in is not None:
            arrays.update(csr_parts("Pin", predict_in))
            arrays[f"Pout{n}"] = algo.predict(predict_in).toarray()
    save(name, **arrays)


def main():
    # --- reference fixture: recpack/tests/test_algorithms/test_nearest_neighbour.py:24-31 (data), :61, :67 (predict inputs)
    data = sp.csr_matrix(([1] * 7, ([0, 0, 1, 1, 2, 2, 2], [1, 2, 0, 2, 0, 1, 2])), shape=(4, 3))
    p1 = sp.csr_matrix(([1, 1, 1], ([0, 1, 2], [0, 1, 2])), shape=(3, 3))
    p2 = sp.csr_matrix(([1, 1], ([0, 0], [0, 1])), shape=(1, 3))
    itemknn_case("ref_itemknn_data", data, [1, 2, 10], predict_in=[p1, p2])
    # normalize_sim on the same data (test_nearest_neighbour.py:110-116)
    itemknn_case("ref_itemknn_data_normsim", data, [2], normalize_sim=True)
    # conditional probability with and without pop_discount (test_nearest_neighbour.py:127-181)
    itemknn_case("ref_itemknn_data_condprob", data, [2], similarity="conditional_probability")
    for pd in (1.0, 0.2, 0.5):
        itemknn_case(f"ref_itemknn_data_condprob_pd{pd}", data, [2], similarity="conditional_probability", pop_discount=pd)
    # --- test_nearest_neighbour.py:34-41 (data_empty_col)
    dec = sp.csr_matrix(([1] * 5, ([0, 0, 1, 1, 2], [1, 2, 2, 1, 2])))
    itemknn_case("ref_itemknn_empty_col", dec, [2])
    # --- recpack/tests/test_algorithms/conftest.py:24-34 (X_in, counts > 1) with column 4 emptied
    # as in test_algorithms_base.py:49-67
    x_in = sp.csr_matrix(([1, 2, 1, 1, 1, 1], ([0, 0, 0, 2, 2, 2], [0, 2, 3, 1, 3, 1])), shape=(10, 5))
    itemknn_case("ref_itemknn_x_in", x_in, [2], predict_in=[x_in])
    # --- recpack/tests/test_algorithms/conftest.py:42-58 (larger_matrix, seed 400; duplicates -> counts > 1)
    np.random.seed(400)
    us = [np.random.randint(0, 500) for _ in range(2000)]
    it = [np.random.randint(0, 500) for _ in range(2000)]
    larger = sp.csr_matrix(([1] * 2000, (us, it)), shape=(700, 500))
    itemknn_case("ref_itemknn_larger_matrix", larger, [3, 200], predict_in=[larger[:50]])
    ease_case("ref_ease_larger_matrix", larger, [(10.0, 0.0)], predict_in=larger[:20])
    # --- recpack/tests/test_algorithms/test_ease.py:16-35 (data) + :44 predict input, alphas of :66-87
    ed = sp.csr_matrix(([1] * 9, ([0, 0, 1, 1, 2, 2, 2, 3, 3], [0, 2, 0, 2, 0, 1, 2, 0, 2])), shape=(5, 3))
    ease_case("ref_ease_data", ed, [(0.03, 0), (200.0, 0), (0.03, 1), (0.03, 2)], predict_in=p1)
    # prune (test_ease.py:90-105): matrix before/after _prune at density 2/9
    algo = EASE(l2=0.03, density=2 / 9)
    algo.similarity_matrix_ = sp.csr_matrix(
        ([0.9, 0.8, 0.7, 0.2, -0.9, -0.6], ([0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1]))
    )
    before = algo.similarity_matrix_.toarray()
    algo._prune()
    save("ref_ease_prune", before=before, after=algo.similarity_matrix_.toarray(), density=np.asarray(2 / 9))
    # --- recpack/tests/conftest.py:55-66 (data_knn) + recpack/tests/test_util.py:49-62
    dk = sp.csr_matrix(([0.3, 0.2, 0.1, 0.23, 0.3, 0.5], ([0, 0, 0, 2, 2, 2], [0, 2, 3, 1, 3, 4])), shape=(10, 5))
    save("ref_topk_data_knn", **csr_parts("X", dk), **csr_parts("top2", get_top_K_values(dk, 2)))

    # --- seeded synthetic power-law matrices
    m = powerlaw_csr(2000, 500, 0.02, seed=20260101).to_scipy()
    # duplicate a few interactions (counts > 1 must binarise to the same result)
    m_dup = m.copy().astype(np.int32)
    m_dup.data[::7] = 3
    itemknn_case("syn_itemknn_2000x500", m_dup, [10, 50, 600], predict_in=[m[:300]])
    ease_case("syn_ease_2000x500", m_dup, [(200.0, 0.0), (5.0, 0.5)], predict_in=m[:100])
    itemknn_case("syn_itemknn_condprob_2000x500", m_dup, [20], similarity="conditional_probability", pop_discount=0.5)
    itemknn_case("syn_itemknn_condprob0_2000x500", m_dup, [20], similarity="conditional_probability")
    m2 = powerlaw_csr(10000, 2000, 0.004, seed=20260102).to_scipy()
    algo = ItemKNN(K=100).fit(m2)
    S = sp.csr_matrix(algo.similarity_matrix_)
    S.sort_indices()
    cosf = cosine_full(m2)
    # per-row K'-th reference value (threshold of the tie-aware rule) instead of the full 2000^2 candidates
    save("syn_itemknn_10000x2000", **csr_parts("X", m2), **csr_parts("S100", S), Ks=np.asarray([100]),
         cos_row_nnz=np.diff(cosf.indptr).astype(np.int64))


def normalized_cases():
    """normalize_X=True (nearest_neighbour.py:207-210): the reference's own fixture (test_nearest_neighbour.py:73-107)
    and seeded synthetic matrices, cosine and conditional probability.  ``full_*`` holds every candidate of the
    similarity (before the top-K) for the tie-aware rule."""
    from sklearn.preprocessing import Normalizer

    from recpack.algorithms.nearest_neighbour import compute_conditional_probability, compute_cosine_similarity
    from recpack.matrix import to_csr_matrix

    def full(X, similarity, pd):
        Xn = Normalizer(norm="l1", copy=True).transform(to_csr_matrix(X, binary=True))
        S = compute_cosine_similarity(Xn) if similarity == "cosine" else compute_conditional_probability(Xn, pd or 0)
        S = sp.csr_matrix(S)
        S.sort_indices()
        return S

    data = sp.csr_matrix(([1] * 7, ([0, 0, 1, 1, 2, 2, 2], [1, 2, 0, 2, 0, 1, 2])), shape=(4, 3))
    m = powerlaw_csr(2000, 500, 0.02, seed=20260101).to_scipy()
    m_dup = m.copy().astype(np.int32)
    m_dup.data[::7] = 3
    m3 = powerlaw_csr(6000, 14000, 0.0015, seed=20260105).to_scipy()  # two weighted column tiles
    cases = [("ref_itemknn_data_normX", data, [2], "cosine", None), ("syn_itemknn_normX_2000x500", m_dup, [10, 50], "cosine", None),
             ("syn_itemknn_normX_condprob_2000x500", m_dup, [20], "conditional_probability", 0.5),
             ("syn_itemknn_normX_6000x14000", m3, [10], "cosine", None)]
    for name, X, Ks, sim, pd in cases:
        arrays = dict(csr_parts("X", X))
        arrays["Ks"] = np.asarray(Ks, dtype=np.int64)
        arrays["pop_discount"] = np.asarray(pd or 0.0)
        arrays["similarity"] = np.asarray(sim)
        if X.shape[1] <= 1000:
            arrays.update(csr_parts("cos", full(X, sim, pd)))
        else:
            arrays["cos_row_nnz"] = np.diff(full(X, sim, pd).indptr).astype(np.int64)
        for K in Ks:
            kw = {"similarity": sim, "normalize_X": True}
            if pd is not None:
                kw["pop_discount"] = pd
            S = sp.csr_matrix(ItemKNN(K=K, **kw).fit(X).similarity_matrix_)
            S.sort_indices()
            arrays.update(csr_parts(f"S{K}", S))
        save(name, **arrays)


def ease_xy_cases():
    """EASE_XY (experimental/ease_xy.py): the docstring example and a seeded synthetic pair (y = a later, overlapping draw)."""
    from recpack.algorithms.experimental.ease_xy import EASE_XY

    Xd = sp.csr_matrix(np.array([[1, 0, 1], [1, 0, 1], [1, 1, 1]]))
    yd = sp.csr_matrix(np.array([[0, 0, 1], [0, 0, 1], [1, 0, 0]]))
    m = powerlaw_csr(1500, 300, 0.03, seed=20260112).to_scipy()
    m2 = powerlaw_csr(1500, 300, 0.02, seed=20260111).to_scipy()
    for name, X, y, params in (("ref_ease_xy_doc", Xd, yd, [(1.0, 0.0), (1000.0, 0.0)]),
                               ("syn_ease_xy_1500x300", m, m2, [(200.0, 0.0), (5.0, 0.5)])):
        arrays = dict(csr_parts("X", X))
        arrays.update(csr_parts("Y", y))
        arrays["l2"] = np.asarray([p[0] for p in params], dtype=np.float64)
        arrays["alpha"] = np.asarray([p[1] for p in params], dtype=np.float64)
        for n, (l2, alpha) in enumerate(params):
            algo = EASE_XY(l2=l2, alpha=alpha).fit(X, y)
            arrays[f"B{n}"] = algo.similarity_matrix_.toarray()
            arrays[f"Pout{n}"] = algo.predict(X[:50]).toarray()
        save(name, **arrays)


def pipeline_case():
    """C1 (BASELINE.json configs[0]): the MovieLens-100K sample that ships with the reference's tests
    (recpack/tests/test_datasets/datasets/ml-100k_sample.data, default filters -> 373 x 390, 4620 interactions),
    WeakGeneralization(0.75, seed=42), ItemKNN(K=50) and EASE(l2=20), history removal as in
    pipelines/pipeline.py:174-175, NDCGK / RecallK / CalibratedRecallK / CoverageK at K=10.  Stored: the split's CSR
    arrays, the reference's score matrices and metric values (data, not the sample file)."""
    import logging

    logging.disable(logging.CRITICAL)
    from recpack.datasets import MovieLens100K
    from recpack.metrics import NDCGK, CalibratedRecallK, CoverageK, RecallK
    from recpack.postprocessing.filters import ExcludeItems
    from recpack.scenarios import WeakGeneralization

    d = MovieLens100K(path="/root/reference/recpack/tests/test_datasets/datasets", filename="ml-100k_sample.data")
    im = d.load()
    sc = WeakGeneralization(0.75, seed=42, validation=False)
    sc.split(im)
    tr, tin, tout = sc.full_training_data, sc.test_data_in, sc.test_data_out
    arrays = {}
    arrays.update(csr_parts("train", tr.values))
    arrays.update(csr_parts("test_in", tin.values))
    arrays.update(csr_parts("test_out", tout.values))
    # a post-filter as Pipeline would apply it: the 20 most popular training items are not to be recommended
    pop = np.argsort(-np.asarray(tr.binary_values.sum(axis=0)).ravel(), kind="stable")[:20]
    arrays["excluded_items"] = pop.astype(np.int64)
    names = []
    for algo in (ItemKNN(K=50), EASE(l2=20)):
        algo.fit(tr)
        Xp = algo.predict(tin)
        Xp = Xp - Xp.multiply(tin.binary_values)
        Xp = sp.csr_matrix(Xp)
        Xp.eliminate_zeros()
        Xp.sort_indices()
        arrays.update(csr_parts(f"scores_{algo.name}", Xp))
        Xf = ExcludeItems(pop).apply(Xp)
        for tag, M in (("", Xp), ("_filtered", Xf)):
            vals = []
            for metric in (NDCGK(10), RecallK(10), CalibratedRecallK(10), CoverageK(10)):
                metric.calculate(tout.binary_values, M)
                vals.append(float(metric.value))
            arrays[f"metrics_{algo.name}{tag}"] = np.asarray(vals)
            print(algo.name + tag, vals)
        names.append(algo.name)
    arrays["metric_names"] = np.asarray(["NDCGK_10", "RecallK_10", "CalibratedRecallK_10", "CoverageK_10"])
    save("c1_pipeline_ml100k", **arrays)


def _tars_events(seed, U, I, n_events, t_max=2_000_000):
    """Seeded event log with repeated (user, item) pairs: power-law users and items, integer timestamps >= 1."""
    rng = np.random.default_rng(seed)
    pu = 1.0 / np.arange(1, U + 1) ** 0.7
    pi = 1.0 / np.arange(1, I + 1) ** 0.9
    uid = rng.choice(U, size=n_events, p=pu / pu.sum())
    iid = rng.permutation(I)[rng.choice(I, size=n_events, p=pi / pi.sum())]
    ts = rng.integers(1, t_max, size=n_events)
    return uid.astype(np.int64), iid.astype(np.int64), ts.astype(np.int64)


def _tars_run(algo, im, arrays, tag, weights=False):
    if weights:  # the weighted matrices the algorithm builds (host logic of the drop-in classes)
        arrays.update(csr_parts(f"Wfit_{tag}", algo._add_decay_to_fit_matrix(im)))
        arrays.update(csr_parts(f"Wpred_{tag}", algo._add_decay_to_predict_matrix(im)))
    algo.fit(im)
    arrays.update(csr_parts(f"S_{tag}", algo.similarity_matrix_))
    arrays.update(csr_parts(f"P_{tag}", algo.predict(im)))


def tars_cases():
    """Time-aware ItemKNN (recpack/algorithms/time_aware_item_knn/): the reference's own fixture
    (recpack/tests/conftest.py:68-77 ``mat``, used by test_time_aware_item_knn_base.py:54-110) and seeded event logs;
    every decay function x similarity of TARSItemKNN plus the weight-only variants.  Stored per case: the events,
    ``similarity_matrix_`` and ``predict`` of the same interactions."""
    import pandas as pd
    from recpack.algorithms.time_aware_item_knn import (TARSItemKNN, TARSItemKNNDing, TARSItemKNNLee, TARSItemKNNLiu,
                                                        TARSItemKNNLiu2012, TARSItemKNNVaz)
    from recpack.matrix import InteractionMatrix

    def matrix(uid, iid, ts, shape):
        df = pd.DataFrame({"uid": uid, "iid": iid, "ts": ts})
        im = InteractionMatrix(df, "iid", "uid", timestamp_ix="ts", shape=shape)
        return im

    # --- reference fixture
    uid = np.array([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5])
    iid = np.array([0, 1, 2, 3, 0, 1, 2, 4, 0, 1, 2])
    ts = np.array([3, 2, 1, 4, 0, 1, 2, 4, 0, 1, 2])
    im = matrix(uid, iid, ts, (6, 5))
    arrays = {"uid": uid, "iid": iid, "ts": ts, "shape": np.array([6, 5]), "K": np.array([2])}
    arrays.update(csr_parts("T", im.last_timestamps_matrix))
    for sim in ("cosine", "conditional_probability", "pearson"):
        _tars_run(TARSItemKNN(K=2, fit_decay=0.5, predict_decay=0.5, similarity=sim), im, arrays, sim)
    save("ref_tars_mat", **arrays)

    # --- seeded log, every decay function x similarity
    U, I, K = 400, 150, 10
    uid, iid, ts = _tars_events(20260701, U, I, 7000)
    im = matrix(uid, iid, ts, (U, I))
    arrays = {"uid": uid, "iid": iid, "ts": ts, "shape": np.array([U, I]), "K": np.array([K]), "decay_interval": np.array([3600])}
    decays = {"exponential": 0.004, "convex": 0.995, "concave": 0.3, "log": 5.0, "linear": 0.8, "inverse": 0.5}
    arrays["decay_names"] = np.array(list(decays.keys()))
    arrays["decay_values"] = np.array(list(decays.values()))
    for fn, d in decays.items():
        for sim in ("cosine", "conditional_probability", "pearson"):
            _tars_run(TARSItemKNN(K=K, fit_decay=d, predict_decay=d, decay_interval=3600, similarity=sim, decay_function=fn), im, arrays,
                      f"{fn}_{sim}", weights=(sim == "cosine"))
    # weight-only variants
    _tars_run(TARSItemKNNDing(K=K, predict_decay=1 / (24 * 3600), similarity="conditional_probability"), im, arrays, "ding")
    _tars_run(TARSItemKNNLiu(K=K, fit_decay=1 / (24 * 3600), predict_decay=1 / (7 * 24 * 3600)), im, arrays, "liu")
    _tars_run(TARSItemKNNVaz(K=K, fit_decay=1 / (24 * 3600), predict_decay=1 / (24 * 3600)), im, arrays, "vaz")
    _tars_run(TARSItemKNNLiu2012(K=K, decay=3.0), im, arrays, "liu2012", weights=True)
    for w in (3, 5):
        for sim in ("cosine", "pearson"):
            _tars_run(TARSItemKNNLee(K=K, w=w, similarity=sim), im, arrays, f"lee{w}_{sim}", weights=(sim == "cosine"))
    save("syn_tars_400x150", **arrays)

    # --- more items than one weighted LDS tile (12288 columns): per-tile candidates + merge
    U, I, K = 3000, 13000, 5
    uid, iid, ts = _tars_events(20260702, U, I, 60000)
    im = matrix(uid, iid, ts, (U, I))
    arrays = {"uid": uid, "iid": iid, "ts": ts, "shape": np.array([U, I]), "K": np.array([K])}
    for sim in ("cosine", "pearson", "conditional_probability"):
        algo = TARSItemKNN(K=K, fit_decay=1 / (24 * 3600), predict_decay=1 / (24 * 3600), similarity=sim).fit(im)
        arrays.update(csr_parts(f"S_{sim}", algo.similarity_matrix_))
    save("syn_tars_3000x13000", **arrays)


def tars_cooc_cases():
    """TARSItemKNNCoocDistance / TARSItemKNNXia (time_aware_item_knn/base.py:204-294, xia_2010.py): the reference's fixture with
    its known answers (recpack/tests/conftest.py:80-89 ``mat_no_zero_timestamp``, test_time_aware_item_knn_base.py:112-150) and a
    seeded event log with some timestamps of 0 and repeated timestamps, every decay function x {cooc, conditional_probability}."""
    import pandas as pd
    from recpack.algorithms.time_aware_item_knn import TARSItemKNNCoocDistance, TARSItemKNNXia
    from recpack.matrix import InteractionMatrix

    def matrix(uid, iid, ts, shape):
        return InteractionMatrix(pd.DataFrame({"uid": uid, "iid": iid, "ts": ts}), "iid", "uid", timestamp_ix="ts", shape=shape)

    uid = np.array([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5])
    iid = np.array([0, 1, 2, 3, 0, 1, 2, 4, 0, 1, 2])
    ts = np.array([4, 3, 2, 5, 1, 2, 3, 5, 1, 2, 3])
    im = matrix(uid, iid, ts, (6, 5))
    arrays = {"uid": uid, "iid": iid, "ts": ts, "shape": np.array([6, 5]), "K": np.array([2])}
    for sim in ("cooc", "conditional_probability"):
        algo = TARSItemKNNCoocDistance(K=2, fit_decay=0.5, similarity=sim).fit(im)
        arrays.update(csr_parts(f"S_{sim}", algo.similarity_matrix_))
        arrays.update(csr_parts(f"P_{sim}", algo.predict(im)))
    from recpack.algorithms.time_aware_item_knn import TARSItemKNNHermann as _H

    arrays.update(csr_parts("S_hermann", _H(K=2).fit(im).similarity_matrix_))
    save("ref_tars_cooc_mat", **arrays)

    U, I, K = 300, 120, 8
    uid, iid, ts = _tars_events(20260703, U, I, 4500, t_max=40 * 24 * 3600)
    ts = (ts // 3600) * 3600  # whole hours: users with two events in the same hour (distance 0 adds nothing)
    ts[::97] = 0  # some timestamps of 0: excluded by the `> 0` masks
    im = matrix(uid, iid, ts, (U, I))
    arrays = {"uid": uid, "iid": iid, "ts": ts, "shape": np.array([U, I]), "K": np.array([K]), "decay_interval": np.array([3600])}
    decays = {"exponential": 0.02, "convex": 0.98, "concave": 0.3, "log": 5.0, "linear": 0.8, "inverse": 0.5}
    arrays["decay_names"] = np.array(list(decays.keys()))
    arrays["decay_values"] = np.array(list(decays.values()))
    for fn, d in decays.items():
        for sim in ("cooc", "conditional_probability"):
            algo = TARSItemKNNCoocDistance(K=K, fit_decay=d, decay_interval=3600, similarity=sim, decay_function=fn).fit(im)
            arrays.update(csr_parts(f"S_{fn}_{sim}", algo.similarity_matrix_))
    algo = TARSItemKNNCoocDistance(K=K, fit_decay=0, decay_interval=3600, similarity="cooc").fit(im)  # no decay: co-occurrence counts
    arrays.update(csr_parts("S_none_cooc", algo.similarity_matrix_))
    algo = TARSItemKNNXia(K=K, fit_decay=0.6, decay_function="concave", decay_interval=24 * 3600).fit(im)
    arrays.update(csr_parts("S_xia", algo.similarity_matrix_))
    arrays.update(csr_parts("P_xia", algo.predict(im)))
    from recpack.algorithms.time_aware_item_knn import TARSItemKNNHermann

    for interval in (1, 3600):
        algo = TARSItemKNNHermann(K=K, decay_interval=interval).fit(im)
        arrays.update(csr_parts(f"S_hermann{interval}", algo.similarity_matrix_))
    arrays.update(csr_parts("P_hermann3600", algo.predict(im)))
    save("syn_tars_cooc_300x120", **arrays)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tars_cooc":
        tars_cooc_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tars":
        tars_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pipeline":
        pipeline_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "normalized":
        normalized_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "ease_xy":
        ease_xy_cases()
    else:
        main()
        normalized_cases()
        ease_xy_cases()
        pipeline_case()
