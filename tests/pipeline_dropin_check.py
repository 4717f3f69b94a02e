"""Drive the reference's REAL ``recpack.pipelines.Pipeline`` through the drop-in classes.

Runs in the BUILD container only, as its own process (tests/test_reference_pipeline.py): /root/reference goes on sys.path BEFORE
recpack_amd is imported, so that ``recpack_amd.algorithms.Algorithm`` derives from ``recpack.algorithms.base.Algorithm``; the missing
``hyperopt`` package is the stub of tests/stubs; the HIP entry points are the oracle-backed stand-ins of tests/fake_lib.py (no GPU
here).  What is checked is the PROTOCOL the north star names - "drops into recpack.pipelines unchanged" - on the reference's own code:

  registry      ALGORITHM_REGISTRY.register(<new name>, <subclass>) as INTEGRATION.md section 2 says; PipelineBuilder.add_algorithm
                resolves the name (pipelines/registries.py:50-75, pipeline_builder.py:70-77, 140);
  run           Pipeline.run(): ``ALGORITHM_REGISTRY.get(name)(**params)``, ``fit(InteractionMatrix)``, ``predict(InteractionMatrix)``,
                history removal on the returned CSR, the post-filter, ``metric.calculate`` and the accumulator keyed by
                ``algorithm.identifier`` (pipelines/pipeline.py:135-179);
  numbers       the metric table against tests/golden/c1_pipeline_ml100k.npz (the reference's own values) and against the stock
                recpack.algorithms.ItemKNN / EASE run in the SAME pipeline;
  surface       isinstance(recpack Algorithm), not a TorchMLAlgorithm, get_params / identifier texts, the two warnings, str().
Prints one line per check and "PIPELINE DROP-IN OK" at the end; any failure raises."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "tests", "stubs"), REF, ROOT, os.path.join(ROOT, "tests")]

import logging  # noqa: E402

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

logging.disable(logging.CRITICAL)
import recpack  # noqa: E402,F401
from recpack.algorithms import EASE as RefEASE, ItemKNN as RefItemKNN  # noqa: E402
from recpack.algorithms.base import Algorithm as RefAlgorithm, TorchMLAlgorithm  # noqa: E402
from recpack.datasets import MovieLens100K  # noqa: E402
from recpack.pipelines import ALGORITHM_REGISTRY, PipelineBuilder  # noqa: E402
from recpack.postprocessing.filters import ExcludeItems  # noqa: E402
from recpack.scenarios import WeakGeneralization  # noqa: E402

import fake_lib  # noqa: E402
import recpack_amd  # noqa: E402
from recpack_amd import _lib  # noqa: E402

fake_lib.install(_lib)
recpack_amd.algorithms._contexts.clear()


class ItemKNNHip(recpack_amd.ItemKNN):
    pass


class EASEHip(recpack_amd.EASE):
    pass


ALGORITHM_REGISTRY.register("ItemKNNHip", ItemKNNHip)
ALGORITHM_REGISTRY.register("EASEHip", EASEHip)
try:
    ALGORITHM_REGISTRY.register("ItemKNN", recpack_amd.ItemKNN)
    raise SystemExit("re-registering a stock name should raise (registries.py:73-74)")
except KeyError:
    print("ok: stock names cannot be re-registered, new names can")

g = np.load(os.path.join(ROOT, "tests", "golden", "c1_pipeline_ml100k.npz"))


def csr(prefix):
    return sp.csr_matrix((g[f"{prefix}_data"], g[f"{prefix}_indices"], g[f"{prefix}_indptr"]), shape=tuple(int(x) for x in g[f"{prefix}_shape"]))


d = MovieLens100K(path=os.path.join(REF, "recpack/tests/test_datasets/datasets"), filename="ml-100k_sample.data")
scenario = WeakGeneralization(0.75, seed=42, validation=False)
scenario.split(d.load())
assert (scenario.full_training_data.values != csr("train")).nnz == 0 and (scenario.test_data_in.values != csr("test_in")).nnz == 0
print("ok: the scenario's split is the golden file's", scenario.full_training_data.shape)

NAMES = ["NDCGK_10", "RecallK_10", "CalibratedRecallK_10", "CoverageK_10"]
TOL = {"ItemKNN": (2e-3, 2e-3, 2e-3, 4.0 / 390), "EASE": (1e-6, 1e-6, 1e-6, 1e-9)}  # (ItemKNN: ties at rank K are numpy's choice in the reference)


def run(post_filter=None):
    b = PipelineBuilder(folder_name="unused", base_path="/tmp")
    b.set_data_from_scenario(scenario)
    for name, params in (("ItemKNN", {"K": 50}), ("ItemKNNHip", {"K": 50}), ("EASE", {"l2": 20}), ("EASEHip", {"l2": 20})):
        b.add_algorithm(name, params=params)
    b.add_algorithm(ItemKNNHip, params={"K": 50, "normalize_sim": True})  # (by class: PipelineBuilder turns it into cls.__name__)
    for m in ("NDCGK", "RecallK", "CalibratedRecallK", "CoverageK"):
        b.add_metric(m, 10)
    if post_filter is not None:
        b.add_post_filter(post_filter)
    p = b.build()
    fake_lib.CALLS.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p.run()
    return p.get_metrics(), [str(x.message) for x in w], list(fake_lib.CALLS)


for tag, flt in (("", None), ("_filtered", ExcludeItems(g["excluded_items"]))):
    df, warns, calls = run(flt)
    print(df.to_string())
    ids = list(df.index)
    assert len(ids) == 5, ids
    knn_ref = RefItemKNN(K=50).identifier
    assert knn_ref in ids and knn_ref.replace("ItemKNN(", "ItemKNNHip(") in ids, (knn_ref, ids)  # same parameter string, the class's own name
    ease_ref = RefEASE(l2=20).identifier
    assert ease_ref in ids and ease_ref.replace("EASE(", "EASEHip(") in ids, (ease_ref, ids)
    for algo in ("ItemKNN", "EASE"):
        ref_id = knn_ref if algo == "ItemKNN" else ease_ref
        hip_id = ref_id.replace(algo + "(", algo + "Hip(")
        want = g[f"metrics_{algo}{tag}"]
        for name, w_, tol in zip(NAMES, want, TOL[algo]):
            stock, ours = float(df.loc[ref_id, name]), float(df.loc[hip_id, name])
            assert abs(stock - w_) <= 1e-12, ("the stock class no longer reproduces the golden", algo, name, stock, w_)
            assert abs(ours - w_) <= tol, (algo, tag, name, ours, w_)
    # the calls went through the library's entry points: fit + predict per drop-in entry (3 of them), nothing else fitted there
    assert [c[0] for c in calls].count("itemknn_fit") == 2 and [c[0] for c in calls].count("ease_fit") == 1, calls
    assert [c[0] for c in calls].count("predict_full") == 3, calls
    # the reference's two warnings come from the drop-in classes with the reference's texts (base.py:120-127, 272-279)
    assert any(x.startswith("ItemKNNHip missing similar items for") for x in warns) == any(x.startswith("ItemKNN missing similar items for") for x in warns), warns
    print(f"ok: Pipeline.run(){' + ExcludeItems' if flt is not None else ''}: metric table equals the golden and the stock classes' rows")

# ---- the class surface the pipeline and the registry rely on
a, r = ItemKNNHip(K=50), RefItemKNN(K=50)
assert isinstance(a, RefAlgorithm) and not isinstance(a, TorchMLAlgorithm) and isinstance(EASEHip(), RefAlgorithm)
assert a.get_params() == r.get_params(), (a.get_params(), r.get_params())
assert EASEHip(l2=7, alpha=0.5, density=0.1).get_params() == RefEASE(l2=7, alpha=0.5, density=0.1).get_params()
assert str(a) == "ItemKNNHip" and a.name == "ItemKNNHip"
tr = scenario.full_training_data
a.fit(tr)  # an InteractionMatrix, as Pipeline._train passes it
r.fit(tr)
assert a.similarity_matrix_.shape == r.similarity_matrix_.shape and a.similarity_matrix_.nnz == r.similarity_matrix_.nnz
empty_in = sp.csr_matrix(tr.values.shape)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    P = a.predict(scenario.test_data_in)
    a.predict(empty_in)
assert isinstance(P, sp.csr_matrix) and P.shape == tr.shape
lone = sp.lil_matrix(tr.shape)
lone[0, int(np.argmin(np.asarray(tr.binary_values.sum(axis=0)).ravel()))] = 1  # a user whose only item nobody else has
with warnings.catch_warnings(record=True) as w2:
    warnings.simplefilter("always")
    a.predict(lone.tocsr())
    r.predict(lone.tocsr())
msgs = sorted(str(x.message) for x in w2)
assert any(m.startswith("ItemKNNHip failed to recommend any items for") for m in msgs) == any(m.startswith("ItemKNN failed to recommend any items for") for m in msgs), msgs
for bad in ({"K": 50, "similarity": "nope"}, {"K": 50, "pop_discount": 2.0, "similarity": "conditional_probability"}):
    for cls in (ItemKNNHip, RefItemKNN):
        try:
            cls(**bad)
            raise SystemExit(f"{cls.__name__}({bad}) should raise")
        except ValueError:
            pass
print("ok: isinstance / get_params / identifier / str / fit(InteractionMatrix) / warnings / constructor validation as the reference's")
print("PIPELINE DROP-IN OK")
