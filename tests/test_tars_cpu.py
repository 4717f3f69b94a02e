"""Time-aware ItemKNN on the CPU: the oracle restatement against the reference's goldens (oracle/gen_golden.py tars) and
the host logic of the drop-in classes (weights per interaction, argument checks).  No GPU, no compute through the library."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from, load_golden
from oracle import oracle_np as onp
from parity import tie_aware_topk

DECAYS = {"exponential": 0.004, "convex": 0.995, "concave": 0.3, "log": 5.0, "linear": 0.8, "inverse": 0.5}
SIMS = ("cosine", "conditional_probability", "pearson")


def last_timestamps(g):
    uid, iid, ts = g["uid"], g["iid"], g["ts"]
    shape = tuple(int(x) for x in g["shape"])
    key = uid * shape[1] + iid
    order = np.lexsort((ts, key))
    last = np.ones(order.shape[0], dtype=bool)
    last[:-1] = key[order][1:] != key[order][:-1]
    sel = order[last]
    return sp.csr_matrix((ts[sel], (uid[sel], iid[sel])), shape=shape)


def check_fit_against_golden(W, K, sim, S_ref):
    r = onp.tars_fit(W, K, sim)
    bad, exact = tie_aware_topk(r["full"], S_ref, r["idx"], r["nnz"], positive_only=False)
    assert not bad, (sim, bad[:5])
    S = onp.topk_to_csr(r["idx"], r["val"], r["nnz"], W.shape[1])
    common = S.multiply(S_ref != 0)
    ref_common = S_ref.multiply(S != 0)
    assert abs(common - ref_common).max() <= 1e-12 * max(1.0, abs(S_ref).max())
    return S, exact


def test_oracle_tars_reference_fixture():
    g = load_golden("ref_tars_mat")
    T = csr_from(g, "T")
    assert T.nnz == 11  # the two timestamps of 0 are stored entries (conftest.py:68-77 of the reference)
    assert (last_timestamps(g) != T).nnz == 0
    W = onp.tars_weights(T, 0.5, "exponential")
    now = 5
    np.testing.assert_array_equal(W.toarray()[0, :2], [np.exp(-(now - 3) / 2), np.exp(-(now - 2) / 2)])  # test_..._base.py:54-72
    for sim in SIMS:
        S_ref = csr_from(g, f"S_{sim}")
        S, _ = check_fit_against_golden(W, 2, sim, S_ref)
        P = onp.tars_predict(W, S_ref)
        P_ref = csr_from(g, f"P_{sim}")
        P_ref.eliminate_zeros()
        assert abs(P - P_ref).max() <= 1e-14


@pytest.mark.parametrize("fn", list(DECAYS))
def test_oracle_tars_synthetic(fn):
    g = load_golden("syn_tars_400x150")
    T = last_timestamps(g)
    K, interval = int(g["K"][0]), int(g["decay_interval"][0])
    W = onp.tars_weights(T, DECAYS[fn], fn, interval)
    W_ref = csr_from(g, f"Wfit_{fn}_cosine")
    assert abs(W - W_ref).max() <= 1e-15
    for sim in SIMS:
        S_ref = csr_from(g, f"S_{fn}_{sim}")
        check_fit_against_golden(W, K, sim, S_ref)
        P = onp.tars_predict(W, S_ref)
        P_ref = csr_from(g, f"P_{fn}_{sim}")
        P_ref.eliminate_zeros()
        assert abs(P - P_ref).max() <= 1e-12 * max(1.0, abs(P_ref).max())


def test_host_weights_match_reference():
    """The per-interaction weights the drop-in classes compute on the host, against the reference's weighted matrices."""
    from recpack_amd import time_aware as ta

    g = load_golden("syn_tars_400x150")
    X = ta.TimedInteractions(g["uid"], g["iid"], g["ts"], tuple(int(x) for x in g["shape"]))
    assert (X.last_timestamps_matrix != last_timestamps(g)).nnz == 0
    K, interval = int(g["K"][0]), int(g["decay_interval"][0])
    for fn, d in DECAYS.items():
        algo = ta.TARSItemKNN(K=K, fit_decay=d, predict_decay=d, decay_interval=interval, decay_function=fn)
        assert abs(algo._add_decay_to_fit_matrix(X) - csr_from(g, f"Wfit_{fn}_cosine")).max() <= 1e-15
        assert abs(algo._add_decay_to_predict_matrix(X) - csr_from(g, f"Wpred_{fn}_cosine")).max() <= 1e-15
    liu = ta.TARSItemKNNLiu2012(K=K, decay=3.0)
    assert abs(liu._add_decay_to_fit_matrix(X) - csr_from(g, "Wfit_liu2012")).max() <= 1e-15
    assert abs(liu._add_decay_to_predict_matrix(X) - csr_from(g, "Wpred_liu2012")).max() <= 1e-15
    for w in (3, 5):
        lee = ta.TARSItemKNNLee(K=K, w=w)
        assert abs(lee._add_decay_to_fit_matrix(X) - csr_from(g, f"Wfit_lee{w}_cosine")).max() == 0
        Wp = lee._add_decay_to_predict_matrix(X)
        assert abs(Wp - csr_from(g, f"Wpred_lee{w}_cosine")).max() == 0 and np.all(Wp.data == 1.0)


def test_constructor_checks():
    """Argument validation and texts of time_aware_item_knn/base.py:94-125 and decay_functions.py."""
    from recpack_amd import time_aware as ta

    with pytest.raises(ValueError, match="decay_interval needs to be a positive integer"):
        ta.TARSItemKNN(decay_interval=0)
    with pytest.raises(ValueError, match="decay_interval needs to be a positive integer"):
        ta.TARSItemKNN(decay_interval=2.0)
    with pytest.raises(ValueError, match="Similarity jaccard is not supported"):
        ta.TARSItemKNN(similarity="jaccard")
    with pytest.raises(ValueError, match="Decay function step is not supported"):
        ta.TARSItemKNN(decay_function="step")
    with pytest.raises(ValueError, match=r"not in the supported range: \[0, 1\]"):
        ta.TARSItemKNN(fit_decay=2, decay_function="exponential")
    with pytest.raises(ValueError, match=r"not in the supported range: \]1, inf\["):
        ta.TARSItemKNN(fit_decay=0.5, predict_decay=0.5, decay_function="log")
    ta.TARSItemKNN(fit_decay=0, predict_decay=0, decay_function="log")  # 0 = no decay: not range-checked
    with pytest.raises(ValueError, match="Only 3 and 5 are allowed"):
        ta.TARSItemKNNLee(w=4)
    with pytest.raises(ValueError, match=r"\]1, inf \["):
        ta.TARSItemKNNLiu2012(decay=1.0)
    with pytest.raises(ValueError, match="Similarity pearson is not supported"):
        ta.TARSItemKNNDing(similarity="pearson")
    a = ta.TARSItemKNNDing(K=7, predict_decay=0.25)
    assert a.get_params() == {"K": 7, "predict_decay": 0.25, "similarity": "cosine"}
    assert a.identifier == "TARSItemKNNDing(K=7,predict_decay=0.25,similarity=cosine)"
    with pytest.raises(TypeError, match="requires Interaction Matrix as input"):
        ta.TARSItemKNN()._transform_fit_input(sp.csr_matrix((2, 2)))


COOC_DECAYS = {"exponential": 0.02, "convex": 0.98, "concave": 0.3, "log": 5.0, "linear": 0.8, "inverse": 0.5}


def check_cooc(T, K, sim, fn, d, interval, S_ref):
    r = onp.tars_cooc_fit(T, K, sim, fn, d, interval)
    bad, _ = tie_aware_topk(r["full"], S_ref, r["idx"], r["nnz"])
    assert not bad, (fn, sim, bad[:5])
    S = onp.topk_to_csr(r["idx"], r["val"], r["nnz"], T.shape[1])
    assert abs(S.multiply(S_ref != 0) - S_ref.multiply(S != 0)).max() <= 1e-12 * max(1.0, abs(S_ref).max())


def test_oracle_tars_cooc_known_answers():
    """test_time_aware_item_knn_base.py:112-150 of the reference (expected arrays written out there) and the golden."""
    g = load_golden("ref_tars_cooc_mat")
    T = last_timestamps(g)
    e = np.exp
    expected = {
        "cooc": np.array([[0, e(-0.5) * 3, 0, 0, 0], [e(-0.5) * 3, 0, 0, 0, 0], [0, 0, 0, e(-1.5), e(-1.0)], [0, 0, e(-1.5), 0, 0],
                          [0, 0, e(-1.0), 0, 0]]),
        "conditional_probability": np.array([[0, e(-0.5), 0, 0, 0], [e(-0.5), 0, 0, 0, 0], [0, 0, 0, e(-1.5) / 3, e(-1.0) / 3],
                                             [0, 0, e(-1.5), 0, 0], [0, 0, e(-1.0), 0, 0]]),
    }
    for sim, want in expected.items():
        r = onp.tars_cooc_fit(T, 2, sim, "exponential", 0.5)
        S = onp.topk_to_csr(r["idx"], r["val"], r["nnz"], 5)
        np.testing.assert_array_almost_equal(S.toarray(), want)
        assert abs(S - csr_from(g, f"S_{sim}")).max() <= 1e-15


def test_oracle_tars_cooc_synthetic():
    g = load_golden("syn_tars_cooc_300x120")
    T = last_timestamps(g)
    K, interval = int(g["K"][0]), int(g["decay_interval"][0])
    assert (T.data == 0).sum() > 0  # timestamps of 0 are in the fixture
    for fn, d in COOC_DECAYS.items():
        for sim in ("cooc", "conditional_probability"):
            check_cooc(T, K, sim, fn, d, interval, csr_from(g, f"S_{fn}_{sim}"))
    check_cooc(T, K, "cooc", "exponential", 0, interval, csr_from(g, "S_none_cooc"))
    check_cooc(T, K, "cooc", "concave", 0.6, 24 * 3600, csr_from(g, "S_xia"))


def test_cooc_constructor_checks():
    from recpack_amd import time_aware as ta

    with pytest.raises(ValueError, match="Similarity cosine is not supported"):
        ta.TARSItemKNNCoocDistance(similarity="cosine")
    with pytest.raises(ValueError, match="decay_function exponential not supported"):
        ta.TARSItemKNNXia(decay_function="exponential")
    a = ta.TARSItemKNNXia(K=5)
    assert a.get_params() == {"K": 5, "decay_function": "convex", "decay_interval": 86400, "fit_decay": 0.5}
    assert a.predict_decay == 0 and a.similarity == "cooc"


def test_oracle_tars_hermann():
    """TARSItemKNNHermann (hermann_2010.py) against the reference's output on its own fixture and on the seeded log."""
    g = load_golden("ref_tars_cooc_mat")
    r = onp.tars_hermann_fit(last_timestamps(g), 2)
    assert abs(onp.topk_to_csr(r["idx"], r["val"], r["nnz"], 5) - csr_from(g, "S_hermann")).max() <= 1e-15
    g = load_golden("syn_tars_cooc_300x120")
    T, K = last_timestamps(g), int(g["K"][0])
    for interval in (1, 3600):
        r = onp.tars_hermann_fit(T, K, interval)
        S_ref = csr_from(g, f"S_hermann{interval}")
        bad, _ = tie_aware_topk(r["full"], S_ref, r["idx"], r["nnz"])
        assert not bad
        S = onp.topk_to_csr(r["idx"], r["val"], r["nnz"], T.shape[1])
        assert abs(S.multiply(S_ref != 0) - S_ref.multiply(S != 0)).max() <= 1e-12 * abs(S_ref).max()
