"""Randomised parity of the ItemKNN fit against the oracle over shapes that cross the kernels' regime boundaries (one / several
column tiles, K around the wave merge's limit of 384, empty rows and columns, cosine and conditional probability with
pop_discount): neighbour sets and values bit-identical to the oracle's deterministic order.  Found while it was written:
numpy's vectorised power and libm's pow differ in the last bit of c_j^-alpha, which decides exact ties of the selection key
(the oracle now goes through libm like the library's host code)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n_cases", [(7, 30), (99, 20)])
def test_fit_fuzz_against_oracle(seed, n_cases):
    from recpack_amd import _lib, get_context

    ctx = get_context(0)
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        I = int(rng.choice([3, 17, 300, 2500, 26000, 40000, 70000]))
        U = int(rng.integers(1, 6000))
        nnz_target = int(rng.integers(1, 400000))
        dens = min(0.5, nnz_target / (U * I))
        K = int(rng.choice([1, 2, 7, 64, 200, 383, 384, 385, 500]))
        sim = str(rng.choice(["cosine", "conditional_probability"]))
        pd = float(rng.choice([0.0, 0.3, 1.0])) if sim != "cosine" else None
        # power-law-ish columns: some heavy rows, many empty ones
        cols = np.minimum((rng.pareto(1.2, size=max(1, int(dens * U * I))) * I / 50).astype(np.int64), I - 1)
        rows = rng.integers(0, U, size=cols.shape[0])
        X = sp.csr_matrix((np.ones(cols.shape[0]), (rows, cols)), shape=(U, I))
        X.sum_duplicates()
        X.data[:] = 1.0
        X.sort_indices()
        ref = onp.itemknn_fit(X, K, similarity=sim, pop_discount=pd)
        dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
        if sim == "cosine":
            m = _lib.itemknn_fit(ctx, dX, K)
        else:
            m = _lib.itemknn_fit(ctx, dX, K, similarity=_lib.SIM_CONDITIONAL_PROBABILITY, pop_discount=pd)
        idx, val, nnz = m.topk_download()
        what = f"case {case}: U={U} I={I} nnz={X.nnz} K={K} {sim} pd={pd}"
        np.testing.assert_array_equal(nnz, ref["nnz"], err_msg=what)
        k = min(idx.shape[1], ref["idx"].shape[1])
        order = np.argsort(np.where(ref["idx"][:, :k] >= 0, ref["idx"][:, :k], np.iinfo(np.int32).max), axis=1, kind="stable")
        ridx = np.take_along_axis(ref["idx"][:, :k], order, axis=1)
        rval = np.take_along_axis(ref["val"][:, :k], order, axis=1)
        np.testing.assert_array_equal(idx[:, :k], ridx, err_msg=what)
        np.testing.assert_allclose(val[:, :k], rval, rtol=1e-12, err_msg=what)
        dX.close()
        m.close()
