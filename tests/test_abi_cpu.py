"""CPU-side checks of the boundary: the C-ABI library loads and exports every declared symbol; host logic."""
import os
import re
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT


def _declared_symbols(header="recpack_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # (comments name functions of other headers)
    return sorted(set(re.findall(r"\b(rphip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from recpack_amd import _lib

    lib = _lib.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/recpack_hip.h but not exported"
    assert set(declared) == set(_lib.PROTOTYPES), set(declared) ^ set(_lib.PROTOTYPES)
    # the measuring instruments and test hooks have a header of their own (include/recpack_hip_diag.h): not the drop-in boundary
    diag = _declared_symbols("recpack_hip_diag.h")
    for name in diag:
        assert hasattr(lib, name), f"{name} declared in include/recpack_hip_diag.h but not exported"
    assert set(diag) == set(_lib.DIAG_PROTOTYPES), set(diag) ^ set(_lib.DIAG_PROTOTYPES)
    assert not set(diag) & set(declared)
    assert lib.rphip_version() == 100


def test_context_options_documented():
    """Every name rphip_ctx_set_option accepts is described in the header, and nothing else is."""
    api = open(os.path.join(ROOT, "recpack_amd", "csrc", "rp_api.hip")).read()
    body = api[api.index("extern \"C\" int rphip_ctx_set_option"):]
    body = body[: body.index("\n}\n")]
    accepted = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', body))
    header = open(os.path.join(ROOT, "include", "recpack_hip.h")).read()
    documented = set(re.findall(r'^ \*   "([a-z0-9_]+)"', header, flags=re.M))
    assert accepted and accepted == documented, accepted ^ documented


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, not compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from recpack_amd import ItemKNN
    from recpack_amd._lib import RecpackHipError

    X = sp.csr_matrix(([1] * 7, ([0, 0, 1, 1, 2, 2, 2], [1, 2, 0, 2, 0, 1, 2])), shape=(4, 3))
    with pytest.raises(RecpackHipError):
        ItemKNN(K=2).fit(X)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "recpack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_itemknn_params_identifier_and_validation():
    # recpack/algorithms/base.py:45-61, nearest_neighbour.py:172-202
    from recpack_amd import EASE, ItemKNN

    a = ItemKNN(K=10, normalize_sim=True)
    assert a.name == "ItemKNN"
    assert a.identifier == "ItemKNN(K=10,normalize_X=False,normalize_sim=True,pop_discount=None,similarity=cosine)"
    assert a.get_params() == {"K": 10, "normalize_X": False, "normalize_sim": True, "pop_discount": None, "similarity": "cosine"}
    assert EASE(l2=200).identifier == "EASE(alpha=0,density=None,l2=200)"
    with pytest.raises(ValueError):
        ItemKNN(similarity="jaccard")
    with pytest.raises(ValueError):
        ItemKNN(similarity="conditional_probability", pop_discount=1.5)
    with pytest.warns(UserWarning, match="pop_discount"):
        ItemKNN(similarity="cosine", pop_discount=0.5)
    a.set_params(K=3)
    assert a.K == 3


def test_to_binary_csr_semantics():
    from recpack_amd.algorithms import UnsupportedTypeError, to_binary_csr

    X = sp.csr_matrix(([1, 2, 0, 1], ([0, 0, 1, 2], [0, 2, 1, 1])), shape=(3, 3))
    B = to_binary_csr(X)
    assert B.nnz == 3 and B.has_sorted_indices
    # duplicates merged
    M = sp.csr_matrix((np.array([1, 1, 1]), np.array([2, 2, 0]), np.array([0, 3])), shape=(1, 3))
    assert to_binary_csr(M).nnz == 2

    class FakeIM:  # recpack InteractionMatrix exposes .values (interaction_matrix.py:202-217)
        values = X

    assert to_binary_csr(FakeIM()).nnz == 3
    with pytest.raises(UnsupportedTypeError):
        to_binary_csr(np.zeros((2, 2)))


def test_post_condition_warnings():
    # recpack/tests/test_algorithms/test_algorithms_base.py:33-47
    from recpack_amd.algorithms import Algorithm

    a = np.ones(5)
    b = a.copy()
    b[2] = 0
    X_pred = sp.diags(b).tocsr()
    X = sp.diags(a).tocsr()
    alg = Algorithm()
    with pytest.warns(UserWarning, match="1 users"):
        alg._check_prediction(X_pred, X)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        alg._check_prediction(X, X)


def test_no_host_side_k_limit_on_the_fits():
    """The reference takes any K (recpack/util.py:63-66, base.py:302-304); since round 6 so does every fit here - no host-side ValueError is
    left: the calls reach the device (without one they fail there: RecpackHipError).  Only the `predict_topk` extension keeps its K."""
    import scipy.sparse as sp
    import torch

    import recpack_amd
    from recpack_amd._lib import RecpackHipError
    from recpack_amd.algorithms import RP_K_MAX, RP_TILE_MAX, RP_WTILE_MAX

    X = sp.random(50, RP_TILE_MAX + 400, density=0.002, format="csr", random_state=0)
    X.data[:] = 1
    Xw = sp.random(50, 2 * RP_WTILE_MAX + 400, density=0.002, format="csr", random_state=0)
    Xw.data[:] = 1
    if not torch.cuda.is_available():
        for algo, M in ((recpack_amd.ItemKNN(K=16384), X), (recpack_amd.ItemKNN(K=RP_K_MAX + 1), X), (recpack_amd.ItemKNN(K=16384, normalize_X=True), Xw)):
            with pytest.raises(RecpackHipError):
                algo.fit(M)


def test_fitted_models_pickle_round_trip(tmp_path):
    """A fitted drop-in estimator goes through pickle / joblib as a BaseEstimator user expects (recpack/algorithms/base.py:33):
    the device handle stays behind, the model travels as host arrays and is uploaded again on first use (here: without a GPU that
    first use raises - there is no CPU fallback - but the round trip itself needs no device)."""
    import pickle

    import joblib
    import torch

    import recpack_amd
    from recpack_amd._lib import RecpackHipError

    idx = np.array([[1, 2], [0, 2], [0, 1]], dtype=np.int32)
    val = np.array([[0.5, 0.8], [0.5, 0.8], [0.8, 0.8]])
    nnz = np.array([2, 2, 2], dtype=np.int32)
    knn = recpack_amd.ItemKNN(K=2)
    knn._model_host = ("topk", idx, val, nnz)
    knn._empty_rows_ = 0
    knn.similarity_matrix_ = recpack_amd.algorithms.topk_arrays_to_csr(idx, val, nnz, 3)
    B = np.array([[0.0, 0.1, 0.9], [0.2, 0.0, 0.2], [0.9, 0.1, 0.0]], dtype=np.float32)
    ease = recpack_amd.EASE(l2=0.03)
    ease._model_host = ("dense", B)
    ease.n_items_ = 3
    ease._empty_rows_ = 0
    for algo, how in ((knn, "pickle"), (ease, "joblib")):
        if how == "pickle":
            back = pickle.loads(pickle.dumps(algo))
        else:
            joblib.dump(algo, tmp_path / "m.joblib")
            back = joblib.load(tmp_path / "m.joblib")
        assert type(back) is type(algo) and back.get_params() == algo.get_params() and back.identifier == algo.identifier
        assert back._model_dev is None and back._model_host[0] == algo._model_host[0]
        for a, b in zip(back._model_host[1:], algo._model_host[1:]):
            np.testing.assert_array_equal(a, b)
        assert back._empty_rows_ == 0
        if not torch.cuda.is_available():
            with pytest.raises(RecpackHipError):
                back._model  # the first use uploads: no device, no model - never a host computation
    assert (pickle.loads(pickle.dumps(knn)).similarity_matrix_ != knn.similarity_matrix_).nnz == 0
    # an unfitted estimator round-trips too
    fresh = pickle.loads(pickle.dumps(recpack_amd.ItemKNN(K=7)))
    assert fresh.K == 7 and fresh._model_host is None


def test_host_code_under_asan():
    """sanitizers belong on the CPU build.  `make -C recpack_amd/csrc asan` compiles the HOST side of EVERY source
    with -fsanitize=address,undefined against a stub HIP runtime (tests/host_asan/hip_stub.cpp: calloc'ed "device" memory, kernels are
    no-ops) and tests/host_asan/driver.cpp exercises it: rp_api.hip's context, block cache, scratch, options and timing bookkeeping,
    and (round 6) the host orchestration of the fits and predicts - K0, the ItemKNN plans (any K), rp_spd_inverse's streams / events /
    plane cache, the distributed solve and its three predict routes on a loop-back communicator with two and three ranks, COO
    ingestion.  Exit code 0 = every check held, ASan, UBSan and LeakSanitizer silent, no "device" block left behind."""
    import shutil
    import subprocess

    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "recpack_amd", "csrc")
    b = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    r = subprocess.run([os.path.join(csrc, "rp_host_asan")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host asan driver: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_wrapper_teardown_in_any_order():
    """The ctypes wrappers' native objects point into their context.  `Context.close()` closes live matrices, models and communicators
    first (communicators last); and when the garbage collector finalises the context of an unreachable cycle BEFORE its children - it
    may, and the weak references are gone by then - a child must not hand the library its dangling object (a core dump at interpreter
    exit, seen after a failed GPU test whose traceback kept a model alive)."""
    import weakref

    from recpack_amd import _lib

    calls = []

    class FakeLib:
        def __getattr__(self, name):
            return lambda *a: calls.append(name) or 0

    def make(cls, ctx):
        o = object.__new__(cls)
        o.ctx, o.handle, o._keep = ctx, object(), None
        ctx._children.add(o)
        return o

    def context():
        c = object.__new__(_lib.Context)
        c.lib, c.handle, c._children = FakeLib(), object(), weakref.WeakSet()
        return c

    ctx = context()
    comm, m, x = make(_lib.Comm, ctx), make(_lib.Model, ctx), make(_lib.DeviceCSR, ctx)
    ctx.close()
    assert sorted(calls[:2]) == ["rphip_csr_destroy", "rphip_model_destroy"] and calls[2:] == ["rphip_comm_destroy", "rphip_ctx_destroy"], calls
    for o in (m, x, comm, ctx):
        o.close()
        assert o.handle is None
    assert len(calls) == 4
    # the collector's order: the context first, its weak set already empty
    del calls[:]
    ctx = context()
    m = make(_lib.Model, ctx)
    ctx._children = weakref.WeakSet()
    ctx.close()
    m.close()
    assert calls == ["rphip_ctx_destroy"] and m.handle is None
    # the usual order: children first
    del calls[:]
    ctx = context()
    m = make(_lib.Model, ctx)
    m.close()
    ctx.close()
    assert calls == ["rphip_model_destroy", "rphip_ctx_destroy"]
