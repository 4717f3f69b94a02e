"""GPU parity: EASE fit (dense Gram, MFMA Cholesky/inverse, scaling) and dense predict through the C ABI.

Bar (BASELINE.json north_star): EASE scores within 1e-4 relative in fp32.  Written here as
  max|B - B_ref| <= 1e-4 * max|B_ref|   and   max|scores - scores_ref| <= 1e-4 * max|scores_ref|
against the reference's float64 goldens and, at sizes without goldens, the float64 oracle.
The fp32 MFMA GEMM itself is checked against float64 numpy at rtol 2e-5 (K = 512 fp32 accumulations).
"""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from, free_port, load_golden
from oracle import oracle_np as onp
from parity import tie_aware_topk

pytestmark = pytest.mark.gpu

EASE_TOL = 1e-4


@pytest.fixture(scope="module")
def hip():
    from recpack_amd import _lib, get_context

    return _lib, get_context(0)


def rel_max(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def test_gemm_nt_mfma(hip):
    import torch

    _lib, ctx = hip
    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, N, K) in [(128, 128, 16), (256, 384, 512), (128, 256, 2048)]:
        A = torch.rand((M, K), generator=g) * 2 - 1
        B = torch.rand((N, K), generator=g) * 2 - 1  # asymmetric operands
        C0 = torch.rand((M, N), generator=g)
        Ad, Bd, Cd = A.cuda(), B.cuda(), C0.clone().cuda()
        _lib.gemm_nt(ctx, Ad, Bd, Cd, alpha=1.0, beta=0.0)
        ctx.sync()
        ref = A.double().numpy() @ B.double().numpy().T
        np.testing.assert_allclose(Cd.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
        Cd = C0.clone().cuda()
        _lib.gemm_nt(ctx, Ad, Bd, Cd, alpha=-0.5, beta=2.0)
        ctx.sync()
        ref2 = -0.5 * ref + 2.0 * C0.double().numpy()
        np.testing.assert_allclose(Cd.cpu().numpy(), ref2, rtol=2e-5, atol=2e-5 * np.abs(ref2).max())
    # A = I picks rows of an asymmetric B: catches a transposed C write
    I = torch.eye(128).cuda()
    Bd = (torch.arange(128 * 128, dtype=torch.float32).reshape(128, 128) / 1000).cuda()
    Cd = torch.zeros((128, 128)).cuda()
    _lib.gemm_nt(ctx, I, Bd, Cd)
    ctx.sync()
    # (the default two-plane fp16 product holds an operand to 2^-24 of its value - fp32's own rounding - not to the last bit)
    np.testing.assert_allclose(Cd.cpu().numpy(), Bd.cpu().numpy().T, rtol=1.2e-7, atol=0)
    ctx.set_option("gemm_mode", 0)  # the fp32-input MFMA (the cross-check mode): exact
    try:
        _lib.gemm_nt(ctx, I, Bd, Cd)
        ctx.sync()
    finally:
        ctx.set_option("gemm_mode", 2)
    np.testing.assert_array_equal(Cd.cpu().numpy(), Bd.cpu().numpy().T)
    with pytest.raises(_lib.RecpackHipError):
        ctx.set_option("gemm_mode", 1)  # (round 2's three-bf16-plane mode was removed in round 4)


def test_gemm_plane_workspace_paths(hip):
    """The large-product route of the GEMM (operands cut into fp16 planes once per K chunk, LDS-DMA product kernel) against float64:
    several K chunks (small workspace), B = a row range of A (one set of planes serves both, one scale),
    B = A, and the in-kernel conversion (gemm_presplit = 0) - all within the error of the plain kernel."""
    import torch

    _lib, ctx = hip
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 3072, 2944, 4096  # 23 x 24 tiles >= 512: the workspace route
    # rows of different size, within the 2^18 of the largest magnitude that keeps both fp16 planes normal (below that the planes'
    # error becomes absolute: 2^-40 of the operand's largest magnitude - see include/recpack_hip.h, gemm_mode)
    A = torch.randn((M, K), generator=g) * torch.exp(1.0 * torch.randn((M, 1), generator=g))
    B = torch.randn((N, K), generator=g) * 37.0
    C0 = torch.randn((M, M), generator=g)
    Ad, Bd = A.cuda(), B.cuda()

    def check(Ax, Bx, alpha, beta, what):
        ref = alpha * (Ax.double().cpu() @ Bx.double().cpu().T) + beta * C0[:, : Bx.shape[0]].double()
        bound = abs(alpha) * (Ax.double().cpu().abs() @ Bx.double().cpu().abs().T) + abs(beta) * C0[:, : Bx.shape[0]].double().abs()
        Cd = C0[:, : Bx.shape[0]].contiguous().cuda()
        _lib.gemm_nt(ctx, Ax, Bx, Cd, alpha=alpha, beta=beta)
        ctx.sync()
        err = ((Cd.cpu().double() - ref).abs() / bound).max().item()
        assert err <= 5e-7, (what, err)

    try:
        ctx.set_option("gemm_workspace_bytes", 64 << 20)  # K chunks of 5120 -> one chunk here ...
        check(Ad, Bd, 0.5, -1.0, "one chunk")
        big = torch.randn((2048, 16384), generator=g).cuda()  # ... and 16 x 16 tiles < 512: stays on the plain kernel
        Cd = torch.zeros((2048, 2048)).cuda()
        _lib.gemm_nt(ctx, big, big, Cd)
        ctx.sync()
        ref = big.double().cpu() @ big.double().cpu().T
        assert ((Cd.cpu().double() - ref).abs() / (big.double().cpu().abs() @ big.double().cpu().abs().T)).max().item() <= 5e-7
        A2 = torch.randn((3072, 16384), generator=g).cuda()  # 24 x 23 tiles, K in four chunks of 5120 (later chunks: beta = 1)
        check(A2, A2[128:128 + N], -1.0, 1.0, "B = rows 128.. of A (aliased planes), four K chunks")
        check(A2, A2, 1.0, 0.0, "B = A")
        ctx.set_option("gemm_presplit", 0)
        check(Ad, Bd, 0.5, -1.0, "conversion inside the product kernel")
    finally:
        ctx.set_option("gemm_presplit", 1)
        ctx.set_option("gemm_workspace_bytes", 4 << 30)
        ctx.trim()


def ease_B(hip, X, l2, alpha=0.0):
    _lib, ctx = hip
    from recpack_amd.algorithms import to_binary_csr

    X = to_binary_csr(sp.csr_matrix(X))
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    m = _lib.ease_fit(ctx, dX, l2, alpha)
    return dX, m, m.dense_download()


@pytest.mark.parametrize("name", ["ref_ease_data", "ref_ease_larger_matrix", "syn_ease_2000x500"])
def test_ease_fit_golden(hip, name):
    _lib, ctx = hip
    g = load_golden(name)
    X = csr_from(g, "X")
    for n, (l2, alpha) in enumerate(zip(g["l2"], g["alpha"])):
        dX, m, B = ease_B(hip, X, float(l2), float(alpha))
        ref = g[f"B{n}"]
        assert rel_max(B, ref) <= EASE_TOL, (name, l2, alpha, rel_max(B, ref))
        assert (np.diag(B) == 0).all()
        if "Pin_indptr" in g:
            from recpack_amd.algorithms import to_binary_csr

            Pin = to_binary_csr(csr_from(g, "Pin"))
            dP = _lib.DeviceCSR.from_host(ctx, Pin.indptr, Pin.indices, Pin.shape)
            ip, ind, dat = _lib.predict_full(ctx, m, dP)
            out = sp.csr_matrix((dat, ind, ip), shape=Pin.shape).toarray()
            assert rel_max(out, g[f"Pout{n}"]) <= EASE_TOL


def test_ease_reference_test_cases(hip):
    """recpack/tests/test_algorithms/test_ease.py:38-52, :66-87 through the drop-in class."""
    from recpack_amd import EASE

    g = load_golden("ref_ease_data")
    data = csr_from(g, "X")
    algo = EASE(l2=0.03).fit(data)
    _in = sp.csr_matrix(([1, 1, 1], ([0, 1, 2], [0, 1, 2])), shape=(3, 3))
    result = algo.predict(_in)
    np.testing.assert_almost_equal(result[2, 0], 1, decimal=1)
    np.testing.assert_almost_equal(result[0, 2], 1, decimal=1)
    a1, a2, a3 = EASE(l2=0.03, alpha=1).fit(data), EASE(l2=0.03, alpha=0).fit(data), EASE(l2=0.03, alpha=2).fit(data)
    np.testing.assert_almost_equal(a1.similarity_matrix_[1, 0], a2.similarity_matrix_[1, 0] / 4, decimal=6)
    np.testing.assert_almost_equal(a1.similarity_matrix_[2, 1], a2.similarity_matrix_[2, 1], decimal=6)
    np.testing.assert_almost_equal(a1.similarity_matrix_[1, 0] / 4, a3.similarity_matrix_[1, 0], decimal=6)
    # known values of SURVEY.md §8(c)
    np.testing.assert_allclose(algo.similarity_matrix_.toarray()[0], [0, 0.12453300125572914, 0.9901932783260373], rtol=1e-4)


def test_ease_prune_dropin(hip):
    # recpack/tests/test_algorithms/test_ease.py:90-105
    from recpack_amd import EASE

    g = load_golden("ref_ease_prune")
    algo = EASE(l2=0.03, density=float(g["density"]))
    algo.similarity_matrix_ = sp.csr_matrix(g["before"])
    algo._prune()
    np.testing.assert_allclose(algo.similarity_matrix_.toarray(), g["after"], rtol=1e-6)
    S = algo.similarity_matrix_
    np.testing.assert_almost_equal(S.nnz / (S.shape[0] * S.shape[1]), 2 / 9)


@pytest.mark.parametrize("shape,l2,alpha,seed", [((20000, 3000, 0.004), 200.0, 0.0, 31), ((8000, 1100, 0.01), 5.0, 0.5, 32),
                                                 ((4000, 700, 0.02), 0.5, 1.0, 33),
                                                 ((15000, 4400, 0.003), 50.0, 0.0, 34)])  # 35 tiles: three panels, the last one short
def test_ease_fit_synthetic_vs_oracle(hip, shape, l2, alpha, seed):
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(shape[0], shape[1], shape[2], seed=seed).to_scipy()
    dX, m, B = ease_B(hip, X, l2, alpha)
    ref = onp.ease_fit(X, l2, alpha)
    err = rel_max(B, ref)
    assert err <= EASE_TOL, err
    # fused top-Kp on the dense model vs float64 scores (tie-aware, fp32 eps)
    users = slice(0, 500)
    scores = sp.csr_matrix(onp.predict_full(X[users], ref))
    for excl in (False, True):
        pi, pv, pn = _lib.predict_topk(ctx, m, dX, 50, exclude_history=excl, user_begin=0, user_end=500)
        ridx, rval, rnnz = onp.topk_rows(scores, 50, exclude=X[users] if excl else None)
        cand = scores
        if excl:
            cand = scores - scores.multiply(onp.binarize(X[users]))
            cand.eliminate_zeros()
        ref_sel = onp.topk_to_csr(ridx, rval, rnnz, X.shape[1])
        smax = np.abs(scores.data).max()
        # values: absolute 1e-4 of the largest score
        lut = sp.csr_matrix(cand)
        for r in range(0, 500, 25):
            n = pn[r]
            want = np.asarray(lut[r, pi[r, :n]].todense()).ravel()
            assert np.abs(pv[r, :n] - want).max() <= EASE_TOL * smax
        # sets: every selected score is within tolerance of the reference's K-th score or above it
        for r in range(500):
            n = pn[r]
            assert n == rnnz[r]
            if n == 0:
                continue
            t = rval[r, : rnnz[r]].min()
            got = np.asarray(lut[r, pi[r, :n]].todense()).ravel()
            assert (got >= t - 2 * EASE_TOL * smax).all()


@pytest.mark.parametrize("n_items,n_users,dens,l2,alpha", [(8192, 100000, 0.004, 200.0, 0.0), (16384, 200000, 0.002, 200.0, 0.0),
                                                         (8192, 100000, 0.004, 10.0, 1.0)])
def test_ease_vs_float64_inverse_8k_16k(hip, n_items, n_users, dens, l2, alpha):
    """The largest sizes a host inverts in reasonable time: all of B against the reference's float64
    inverse (ease.py:81-86 through the oracle), globally and per column; the achieved errors are printed for the trend
    (tests/test_gpu_fullsize.py continues it to 50K, 100K and 200K items with certified float64 columns).  The third case is the
    wide-dynamic-range one: a small l2 (the Gram's own conditioning) and alpha = 1 (columns scaled by 1 / c_j over three decades of
    item popularity, ease.py:88-90) - every column still to the same relative bar."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(n_users, n_items, dens, seed=20260100 + n_items).to_scipy()
    dX, m, B = ease_B(hip, X, l2, alpha)
    ref = onp.ease_fit(X, l2, alpha)
    err = rel_max(B, ref)
    col = np.abs(B - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1e-300)
    c_max = int(np.asarray(X.sum(axis=0)).max())
    print(f"EASE vs float64 inverse: n_items={n_items} nnz={X.nnz} c_max={c_max} err_global={err:.3e} err_col_max={col.max():.3e} "
          f"err_col_median={np.median(col):.3e}")
    assert err <= EASE_TOL, err
    assert col.max() <= EASE_TOL, col.max()
    m.close()
    dX.close()
    ctx.trim()


def test_potrf_trtri_block(hip):
    """The 128 x 128 diagonal block kernel of the blocked Cholesky (float64 factor + inverse in LDS, blocked in 16 columns)
    against numpy float64 for condition numbers up to 1e5; the upper triangles are zero; a non-positive pivot is reported."""
    import torch

    _lib, ctx = hip
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    W = torch.zeros(128, 128, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    for cond in (1e1, 1e3, 1e5):
        Q, _ = np.linalg.qr(rng.standard_normal((128, 128)))
        S = (Q * np.logspace(0, np.log10(cond), 128)) @ Q.T
        S = ((S + S.T) / 2).astype(np.float32)
        A = torch.tensor(S, device=dev)
        _lib.potrf_trtri_128(ctx, A, W, info)
        torch.cuda.synchronize()
        L = np.linalg.cholesky(S.astype(np.float64))
        Wr = np.linalg.inv(L)
        assert rel_max(A.cpu().numpy(), L) < 2e-7
        assert rel_max(W.cpu().numpy(), Wr) < 2e-7
        assert np.triu(A.cpu().numpy(), 1).max() == 0.0 and np.triu(W.cpu().numpy(), 1).max() == 0.0
        assert int(info.item()) == 0
    S = np.eye(128, dtype=np.float32)
    S[40, 40] = -1.0
    _lib.potrf_trtri_128(ctx, torch.tensor(S, device=dev), W, info)
    torch.cuda.synchronize()
    assert int(info.item()) == 41


def test_ease_solve_driver_options(hip):
    """The solve driver's switches give the same B (to the rounding of the plane products): the inverse's plane cache off / whole /
    a prefix of the panels (a byte limit: the products then come in a cached and a freshly cut part), the Cholesky's sub-panels
    off / 8 / 4 tiles.  (The absolute accuracy of the default is what the float64 tests above and the full-size tests measure.)"""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(40000, 9000, 0.004, seed=7).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    try:
        ref = _lib.ease_fit(ctx, dX, 50.0, 0.0).dense_download()
        assert np.isfinite(ref).all()
        for opt, values, back in (("gemm_plane_cache", (0, 4 * 4096 * 4096 + (64 << 20)), 1), ("chol_sub_panel", (0, 4), 8)):
            for v in values:
                ctx.set_option(opt, v)
                try:
                    B = _lib.ease_fit(ctx, dX, 50.0, 0.0).dense_download()
                finally:
                    ctx.set_option(opt, back)
                assert rel_max(B, ref) <= 2e-5, (opt, v, rel_max(B, ref))
        # (round 6) WHEN the inverse's diagonal-block chains are issued - during the factorisation (the first 4 panels by default), all of
        # them there, none of them - changes no bit: the same kernels on the same operands
        for v in (0, 1000):
            ctx.set_option("inv_n22_early", v)
            try:
                B = _lib.ease_fit(ctx, dX, 50.0, 0.0).dense_download()
            finally:
                ctx.set_option("inv_n22_early", 4)
            np.testing.assert_array_equal(B.view(np.uint32), ref.view(np.uint32), err_msg=f"inv_n22_early={v}")
    finally:
        dX.close()
        ctx.trim()


def test_ease_edge_cases(hip):
    _lib, ctx = hip
    from recpack_amd import EASE

    # an item nobody interacted with: its row and column of B stay zero, a warning names it
    X = sp.csr_matrix(([1, 1, 1, 1], ([0, 0, 1, 1], [0, 1, 0, 1])), shape=(3, 3))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        algo = EASE(l2=1.0).fit(X)
        assert any("missing similar items for 1 items" in str(x.message) for x in w)
    B = algo.similarity_matrix_.toarray()
    assert (B[2] == 0).all() and (B[:, 2] == 0).all()
    ref = onp.ease_fit(X, 1.0)
    assert rel_max(B, ref) <= EASE_TOL
    with pytest.raises(_lib.RecpackHipError):
        EASE(l2=0.0).fit(X)
    # users without history predict nothing
    out = algo.predict(sp.csr_matrix((2, 3)))
    assert out.nnz == 0


def test_ease_prune_on_device(hip):
    """ease.py:97-109 on the device: exact count, the same survivors as the oracle's rule when there are no ties, and with
    ties exactly `keep` entries none of which is smaller than a dropped one."""
    from recpack_amd import EASE
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(8000, 1100, 0.01, seed=32).to_scipy()
    dX, m, B = ease_B(hip, X, 5.0, 0.5)
    n = B.shape[0]
    density = 0.03
    want = onp.ease_prune(B.astype(np.float64), density)
    before, after = m.dense_prune(int(density * n * n))
    assert before == np.count_nonzero(B) and after == int(density * n * n)
    got = m.dense_download()
    assert np.count_nonzero(got) == after
    np.testing.assert_array_equal(got != 0, want != 0)
    np.testing.assert_array_equal(got[got != 0], B[got != 0])
    # pruning to more than there is changes nothing; to zero clears the matrix
    assert m.dense_prune(n * n) == (after, after)
    # through the drop-in class, with the warning bookkeeping still working
    algo = EASE(l2=5.0, alpha=0.5, density=density).fit(X)
    S = algo.similarity_matrix_
    assert S.nnz == after
    assert (S.toarray().astype(np.float32) != 0).sum() == after
    # ties: values from a tiny set
    rng = np.random.default_rng(0)
    T = rng.choice(np.array([0.0, 0.25, -0.25, 0.5, -0.5, 1.0], dtype=np.float32), size=(300, 300))
    mt = _lib.Model.from_dense_host(ctx, T)
    keep = int((np.abs(T) >= 0.5).sum()) + 1234  # all of the 0.5 and 1.0 entries plus 1234 of the 0.25 ones
    b, a = mt.dense_prune(keep)
    G = mt.dense_download()
    assert b == np.count_nonzero(T) and a == keep and np.count_nonzero(G) == keep
    assert (G[np.abs(T) >= 0.5] == T[np.abs(T) >= 0.5]).all()
    assert (np.abs(G[G != 0]) >= 0.25).all()
    assert mt.dense_prune(0)[1] == 0 and np.count_nonzero(mt.dense_download()) == 0


def _run_dist_check(nproc, args=()):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RECPACK_DIST_BACKEND="gloo")  # the ranks share this box's GPU(s); RCCL needs one GPU per rank
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "tools", "dist_ease_check.py"), *map(str, args)]
    return subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("nproc,shape", [(2, (4000, 700, 0.02, 2, 1)), (3, (3000, 1000, 0.01, 3, 2)), (2, (3000, 1000, 0.01, 16, 3)),
                                         (3, (2000, 200, 0.03, 1, 1)),      # two panels for three ranks: one rank owns nothing
                                         (3, (3000, 1000, 0.01, 4, 4)),     # two 512-column panels for three ranks, enough items for the owner route
                                         (3, (12000, 3000, 0.005, 4, 4)),     # 6 panels of 512 columns, then the library's own choice
                                         (2, (60000, 20000, 0.002, 16, 16))])  # 2048-column panels: the plane-product kernel with the block-cyclic tile rules
def test_ease_fit_sharded_multi_rank(hip, nproc, shape):
    """The multi-GPU routes of recpack_amd.dist with several ranks (gloo, sharing this box's GPU).
    ease_fit_sharded - the library's distributed solve (rphip_ease_fit_dist: block-cyclic panels, symmetric block sweeps, the
    collectives behind the communicator's callback transport) - gives every rank its COLUMNS of B (1e-4 of the single-GPU B)
    without a single torch allocation, and predict_topk_sharded (local top-Kp per rank, all-gather, merge) reproduces the
    single-GPU top-K.  Fifth number: panel width of the distributed solve's first pass (the second pass lets the library choose;
    the fourth was the panel width of round 2's replicated route, removed in round 4)."""
    r = _run_dist_check(nproc, shape)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("top-20 agreement") == 4 * nproc
    assert r.stdout.count("transport self-test: ok") == nproc  # rphip_comm_selftest: broadcast, all-gather, exchange, all-reduce
    # the owner route (rphip_predict_topk_owner): the gathered model's predict bit for bit, with and without history removal, for
    # both fits and both values of l2 - or refused by every rank alike where the first pass has too few items
    if shape[1] >= 512:
        assert r.stdout.count("identical to the gathered model's predict: True") == 8 * nproc, r.stdout[-3000:]
        import re

        unproven = [int(x) for x in re.findall(r"users without a proof: \d+ here, (\d+) on all ranks", r.stdout)]
        print("owner route, users without a proof per call:", unproven)
    else:
        assert r.stdout.count("owner route refused (expected") == 4 * nproc, r.stdout[-3000:]


def test_bench_two_ranks_share_the_gpu(hip):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank) - here over gloo with both ranks on
    this box's GPU: the transport self-test, the sharded ItemKNN half, the library's distributed EASE solve and the owner route of
    the predict all run, and rank 0 prints one JSON line with the contract's keys."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RECPACK_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "C2", "--cpu-baseline-seconds", "0",
           "--ease-config", "none", "--tars-config", "none"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["status"] == "ok" and d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 1 and d["value"] > 0
    assert d["transport_selftest"]["ok"] and d["transport_selftest"].get("ok_all_ranks", True)
    assert "exact sums by the columns' owners" in d["config"]["parallelism"]
    for k in ("metric", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d


@pytest.mark.parametrize("hard", [False, True])
def test_bench_rank_failure_reaches_rank_zero(hip, hard):
    """a rank != 0 that dies first must not leave rank 0 waiting in a collective without a word.  Rank 1 raises in its
    first warm-up step (test hook); rank 0 - by then inside the all-reduce of the item degrees - prints ONE line whose status names the
    rank, its reason and the phase, and the run exits non-zero within seconds, not at the 1 500 s watchdog.  The failing rank waits for
    rank 0's acknowledgement before it exits (torch.distributed.run terminates the other ranks the moment one exits non-zero); `hard`: the
    rank dies silently (os._exit) - rank 0's line then comes from its SIGTERM thread (sigwait: the main thread sits in C) or from its own
    failed collective, and names the collective it was in."""
    import json
    import os
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RECPACK_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "C2", "--cpu-baseline-seconds", "0",
           "--ease-config", "none", "--tars-config", "none", "--inject-failure-rank", "1"] + (["--inject-failure-hard"] if hard else [])
    t0 = time.time()
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert time.time() - t0 < 300
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    # (either rank 0's watcher saw rank 1's note first, or - over gloo - its own collective noticed the dead peer: the line carries rank 1's
    # reason both ways, and the collective rank 0 was in)
    if hard:  # a rank that crashes without a word: the launcher ends the job with SIGTERM, or rank 0's own collective notices the dead peer - the line says which
        assert d["n_gpus"] == 2 and (d["status"].startswith("terminated by the launcher") or d["status"].startswith("failed in the")) and "all-reduce of the item degrees" in d["status"], d
    else:
        assert d["status"].startswith("failed") and "rank 1" in d["status"] and "injected failure" in d["status"] and d["n_gpus"] == 2, d


def test_ease_fit_sharded_one_rank(hip):
    import torch

    from recpack_amd import dist as rdist
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(6000, 1500, 0.008, seed=77).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    ref = onp.ease_fit(X, 20.0, 0.0)
    dev = torch.device("cuda", 0)
    # the column-sharded route on one rank (one slab = the whole matrix): block sweeps, K7 on the slab, predict + merge
    single = _lib.ease_fit(ctx, dX, 20.0, 0.0)
    want = _lib.predict_topk(ctx, single, dX, 30, exclude_history=True)
    for panel_tiles in (8, 5, 1, 0):
        ms, lay = rdist.ease_fit_sharded(ctx, dX, 20.0, 0.0, dev, panel_tiles=panel_tiles)
        assert lay["n_local"] == 1500 and lay["world"] == 1 and (ms.local_columns() == np.arange(1500)).all()
        ptr, _, ld = ms.dense_device_ptr()
        B = torch.as_tensor(rdist._DevPtr(ptr, (1500, ld), "<f4", ms), device=dev)[:, :1500].cpu().numpy()
        assert rel_max(B, ref) <= EASE_TOL, (panel_tiles, rel_max(B, ref))
        users, pi, pv, pn = rdist.predict_topk_sharded(ctx, ms, dX, 30, dev, exclude_history=True, user_chunk=2500)
        assert (users.cpu().numpy() == np.arange(6000)).all()
        assert (pn.cpu().numpy() == want[2]).all()
        assert (pi.cpu().numpy() == want[0]).mean() > 0.999
        del ms
    with pytest.raises(_lib.RecpackHipError):
        rdist.ease_fit_sharded(ctx, dX, 0.0, 0.0, dev)


@pytest.mark.parametrize("Kp,excl,unsorted", [(200, False, False), (100, True, True), (100, True, False), (600, False, False)])
def test_gathered_model_predict_waits_for_the_f32_rows(hip, Kp, excl, unsorted):
    """a gathered model's f32 rows arrive on a stream of their own (rphip_ease_model_allgather, phase 2); only
    the per-user first pass may run beside them - every other predict route (Kc > 256: Kp = 200 / 600; history removal on unsorted
    rows) must wait for the gather before it reads B.  One rank of two on a loop-back communicator with a SLOW link (the strips
    take ~0.5 s): the predict issued at once must equal the predict after the gather has completed, bit for bit, and the host's
    float64 scores of the downloaded matrix."""
    import torch

    from recpack_amd import dist as rdist
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    n_items = 2048
    X = powerlaw_csr(1500, n_items, 0.01, seed=123).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    Xp = X[:600].copy()
    if unsorted:  # reverse every row's column order
        idx = Xp.indices.copy()
        for u in range(Xp.shape[0]):
            idx[Xp.indptr[u]:Xp.indptr[u + 1]] = idx[Xp.indptr[u]:Xp.indptr[u + 1]][::-1]
        Xp = sp.csr_matrix((Xp.data, idx, Xp.indptr), shape=Xp.shape)
    dXp = _lib.DeviceCSR.from_host(ctx, Xp.indptr, Xp.indices, Xp.shape)
    dev = torch.device("cuda", 0)
    comm = _lib.Comm.loopback(ctx, 2, 0, gbytes_per_s=0.02, latency_us=10.0)
    try:
        ms, lay = rdist.ease_fit_sharded(ctx, dX, 20.0, 0.0, dev, panel_tiles=4, comm=comm)
        # (a loop-back fit computes with noise in the peers' places: the rank's panels are overwritten with known numbers)
        ptr, _, ld = ms.dense_device_ptr()
        n_loc = lay["n_local"]
        Bl = torch.as_tensor(rdist._DevPtr(ptr, (n_items, ld), "<f4", ms), device=dev)
        g = torch.Generator(device="cpu").manual_seed(5)
        Bl[:, :n_loc] = (torch.randn((n_items, n_loc), generator=g) * 0.05).to(dev)
        torch.cuda.synchronize()
        full = rdist.ease_gather_model(ctx, ms, n_items, comm)
        first = _lib.predict_topk(ctx, full, dXp, Kp, exclude_history=excl)  # (the strips are still on their way)
        ctx.sync()
        B = full.dense_download().astype(np.float64)
        again = _lib.predict_topk(ctx, full, dXp, Kp, exclude_history=excl)
        for a, b in zip(first, again):
            np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
        assert np.abs(B).max() > 0.01
        S = np.asarray(Xp @ B)
        if excl:
            S[Xp.nonzero()] = -np.inf
        smax = np.abs(S[np.isfinite(S)]).max()
        pi, pv, pn = first
        for u in range(0, Xp.shape[0], 7):
            k = int(pn[u])
            if Xp.indptr[u + 1] == Xp.indptr[u]:
                assert k == 0
                continue
            want = np.sort(S[u][np.isfinite(S[u]) & (S[u] != 0.0)])[::-1][:Kp]
            assert k == len(want)
            assert np.abs(np.sort(pv[u, :k].astype(np.float64))[::-1] - want).max() <= 1e-5 * smax
            np.testing.assert_allclose(S[u][pi[u, :k]], pv[u, :k], atol=1e-5 * smax)
        del full, ms
    finally:
        comm.close()


def test_ease_fit_sharded_rccl_one_rank(hip):
    """The same code with a one-rank NCCL (= RCCL) group: every collective call of the multi-GPU path runs through RCCL."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RECPACK_FORCE_PG="1", RECPACK_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dist_ease_check.py"), "3000", "600", "0.02", "2"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("name", ["ref_ease_xy_doc", "syn_ease_xy_1500x300"])
def test_ease_xy_golden(hip, name):
    """EASE_XY (experimental/ease_xy.py:61-96): B and predict against the reference's goldens at the EASE tolerance."""
    from recpack_amd import EASE_XY
    from recpack_amd.algorithms import to_binary_csr

    _lib, ctx = hip
    g = load_golden(name)
    X, Y = csr_from(g, "X"), csr_from(g, "Y")
    for n, (l2, alpha) in enumerate(zip(g["l2"], g["alpha"])):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            algo = EASE_XY(l2=float(l2), alpha=float(alpha)).fit(X, Y)
            B = algo.similarity_matrix_.toarray()
            out = algo.predict(X[:50]).toarray()
        assert rel_max(B, g[f"B{n}"]) <= EASE_TOL, (name, l2, alpha, rel_max(B, g[f"B{n}"]))
        assert (np.diag(B) == 0).all()
        assert rel_max(out, g[f"Pout{n}"]) <= EASE_TOL
    with pytest.raises(ValueError):
        EASE_XY(l2=1.0).fit(X, Y[:, :-1])


def test_ease_xy_larger_vs_oracle(hip):
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(9000, 1300, 0.01, seed=51).to_scipy()
    Y = powerlaw_csr(9000, 1300, 0.006, seed=52).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    dY = _lib.DeviceCSR.from_host(ctx, Y.indptr, Y.indices, Y.shape)
    for l2, alpha in ((100.0, 0.0), (10.0, 0.3)):
        B = _lib.ease_xy_fit(ctx, dX, dY, l2, alpha).dense_download()
        ref = onp.ease_xy_fit(X, Y, l2, alpha)
        assert rel_max(B, ref) <= EASE_TOL, rel_max(B, ref)


def test_ease_l2_sweep_from_cached_gram(hip):
    """a sweep over l2 (and alpha) with the Gram computed once equals independent fits bit for bit, and the
    float64 oracle within the EASE tolerance; the Gram handle is refused by predict."""
    from recpack_amd import EASE, EASEGram

    _lib, ctx = hip
    rng = np.random.default_rng(77)
    X = sp.random(900, 300, density=0.03, random_state=rng, format="csr")
    X.data[:] = 1.0
    with EASEGram(X) as g:
        with pytest.raises(_lib.RecpackHipError):
            _lib.predict_topk(ctx, g._gram, g._dX, 5, False)
        for l2, alpha in [(10.0, 0.0), (100.0, 0.0), (1000.0, 0.5)]:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                swept = g.fit(EASE(l2=l2, alpha=alpha))
                alone = EASE(l2=l2, alpha=alpha).fit(X)
            B_s, B_a = swept._model.dense_download(), alone._model.dense_download()
            assert np.array_equal(B_s, B_a)
            B_ref = onp.ease_fit(X, l2=l2, alpha=alpha)
            assert rel_max(B_s.astype(np.float64), B_ref) <= EASE_TOL
    with pytest.raises(RuntimeError):
        g.fit(EASE(l2=1.0))


def test_predict_from_two_threads(hip):
    """Two threads predict on ONE freshly fitted model through one context at the same time: the model's acceleration structures
    (K9's quantised copy of B, K8's tile offsets) are built on first use - under the context's lock since round 5 (every entry
    point holds it: calls on a context are serialised) - and both threads must get the single-threaded result."""
    import threading

    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(7000, 2400, 0.01, seed=93).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    for fit in (lambda: _lib.ease_fit(ctx, dX, 30.0, 0.0), lambda: _lib.itemknn_fit(ctx, dX, 50)):
        want = _lib.predict_topk(ctx, fit(), dX, 40, exclude_history=True)
        for _ in range(3):
            m = fit()  # (fresh: nothing built yet)
            got, err = [None, None], []
            gate = threading.Barrier(2)

            def work(k):
                try:
                    gate.wait()
                    got[k] = _lib.predict_topk(ctx, m, dX, 40, exclude_history=True)
                except Exception as e:  # noqa: BLE001
                    err.append(e)

            ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert not err, err
            for g in got:
                for a, b in zip(g, want):
                    np.testing.assert_array_equal(a, b)


def test_predict_user_route_corner_cases(hip):
    """K9's first pass with one task per user (rp_predict_i8.hip, round 5) against the one-pass kernel, bit for bit, where its special
    paths run: histories longer than its LDS copy (2 048 items) under history removal, users without a history, a catalogue of one
    column tile and of three, Kp at the edge of the route (Kc = 256) and beyond it (the per-(user, tile) route takes over), and the
    per-(user, tile) route itself (`predict_user_tasks` = 0), which stays the fallback for filters and unsorted rows."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    rng = np.random.default_rng(17)
    for n_items in (3000, 40000):
        X = powerlaw_csr(2500, n_items, 40.0 / n_items, seed=94).to_scipy().tolil()
        X[3, :] = 0                                   # no history
        big = rng.choice(n_items, size=2600, replace=False)
        X[5, big] = 1                                 # longer than the LDS copy of a history
        X = X.tocsr()
        X.sort_indices()
        dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
        m = _lib.ease_fit(ctx, dX, 20.0, 0.0)
        try:
            for Kp, excl in ((100, True), (100, False), (164, True), (200, True), (7, True)):
                ctx.set_option("predict_half", 0)
                want = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl)
                ctx.set_option("predict_half", 1)
                for route in (1, 0):
                    ctx.set_option("predict_user_tasks", route)
                    got = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl)
                    for a, b in zip(got, want):
                        np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b,
                                                      err_msg=f"{n_items} items, Kp {Kp}, exclude {excl}, route {route}")
                assert want[2][3] == 0 and (want[2][[0, 1, 2, 5]] > 0).all()
            # an item filter (round 6: the per-user pass clears the filtered columns in its registers; until then a filter meant the
            # per-(user, tile) route): most items kept, a SelectItems of 150 and one of 30 items (fewer than Kp: short rows)
            ctx.set_option("predict_user_tasks", 1)
            for n_keep in (int(0.7 * n_items), 150, 30):
                keep = np.zeros(n_items, dtype=np.uint8)
                keep[rng.choice(n_items, size=n_keep, replace=False)] = 1
                for excl in (True, False):
                    ctx.set_option("predict_half", 0)
                    want = _lib.predict_topk(ctx, m, dX, 100, exclude_history=excl, item_keep=keep)
                    ctx.set_option("predict_half", 1)
                    got = _lib.predict_topk(ctx, m, dX, 100, exclude_history=excl, item_keep=keep)
                    assert _lib.predict_stats(ctx)["per_user_first_pass"]
                    for a, b in zip(got, want):
                        np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b,
                                                      err_msg=f"{n_items} items, {n_keep} kept, exclude {excl}")
                    assert keep[got[0][got[0] >= 0]].all() and got[2].max() <= min(100, n_keep)
        finally:
            ctx.set_option("predict_half", 1)
            ctx.set_option("predict_user_tasks", 1)


@pytest.mark.parametrize("n_items", [13344, 18560])
def test_predict_user_route_keeps_its_users_on_ragged_tiles(hip, n_items):
    """(round 6) The per-user first pass finds its first threshold from the best score of every thread.  Until round 5 every wave of the
    tile had to count down the same number of them: a tile whose last wave owns a handful of threads (200 000 items: 13 tiles of
    15 424 columns = 964 threads, four in the last wave) never got a threshold, handed on every column, overflowed its list - and ALL
    users went to the exact one-pass kernel (same results, 29 s instead of 17 s at 5M x 200K).  The results cannot show that, the
    counter of include/recpack_hip_diag.h does: catalogues whose tiles end in a wave of four threads (one tile of 836 threads, two
    tiles of 580) keep their users on the two-pass route as a catalogue of whole waves does (2 % without a proof at 13 312 items), and
    give the one-pass kernel's result bit for bit."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(40000, n_items, 40.0 / n_items, seed=95).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    m = _lib.ease_fit(ctx, dX, 100.0, 0.0)
    try:
        for excl in (True, False):
            ctx.set_option("predict_half", 0)
            want = _lib.predict_topk(ctx, m, dX, 100, exclude_history=excl)
            assert _lib.predict_stats(ctx)["users"] == -1
            ctx.set_option("predict_half", 1)
            got = _lib.predict_topk(ctx, m, dX, 100, exclude_history=excl)
            st = _lib.predict_stats(ctx)
            assert st["per_user_first_pass"] and st["users"] == X.shape[0] and st["candidates_per_user"] == 192, st
            assert 0 <= st["exact_pass_users"] <= 0.06 * X.shape[0], st
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
            # the second list (round 6): the unproven users once more with 256 candidates - it proves some of them, the exact pass takes
            # the rest, the results do not move; switched off, everybody without a proof takes the exact pass at once
            assert st["second_list_users"] > 0 and 0 < st["second_list_proven"] <= st["second_list_users"], st
            ctx.set_option("predict_second_list", 0)
            got0 = _lib.predict_topk(ctx, m, dX, 100, exclude_history=excl)
            st0 = _lib.predict_stats(ctx)
            ctx.set_option("predict_second_list", 1)
            assert st0["second_list_users"] == 0 and st0["exact_pass_users"] == st["exact_pass_users"] + st["second_list_proven"], (st0, st)
            for a, b in zip(got0, want):
                np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    finally:
        ctx.set_option("predict_half", 1)
        ctx.set_option("predict_second_list", 1)


def test_predict_topk_any_kp_dense(hip):
    """predict_topk with Kp beyond 2048 on an EASE model (round 6): predict()'s full float64 sums per chunk of users cut on the device -
    the top-Kp of X @ B (float64 from the downloaded f32 matrix), columns ascending, history removed, several chunks of users
    (6 000 items x 12 bytes: the 4 GB chunk holds every user here, so a second call on a sub-range checks the offsets)."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(3000, 6000, 0.006, seed=97).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    m = _lib.ease_fit(ctx, dX, 30.0, 0.0)
    B = m.dense_download().astype(np.float64)
    S = np.asarray(X[:400] @ B)
    for Kp, excl in ((2500, True), (5999, False), (6000, True)):
        pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl, user_begin=0, user_end=400)
        sub = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl, user_begin=150, user_end=170)
        np.testing.assert_array_equal(sub[0], pi[150:170])
        np.testing.assert_array_equal(sub[1].view(np.uint32), pv[150:170].view(np.uint32))
        smax = np.abs(S).max()
        for u in range(0, 400, 9):
            sc = S[u].copy()
            live = sc != 0.0
            if excl:
                live[X.indices[X.indptr[u]:X.indptr[u + 1]]] = False
            k = min(Kp, int(live.sum()))
            assert pn[u] == k
            got = pi[u, :k]
            assert (np.diff(got) > 0).all() and live[got].all()
            np.testing.assert_allclose(pv[u, :k], sc[got], rtol=2e-6, atol=2e-7 * smax)
            if k < int(live.sum()):  # nothing left out beats the worst one kept (up to the f32 rounding of the sums)
                out = live.copy()
                out[got] = False
                assert sc[out].max() <= sc[got].min() + 2e-6 * smax


def test_predict_two_pass_equals_one_pass(hip):
    """K9's two passes (fp16 or int8 rows for a candidate list, exact f32 scores of the candidates, a proof per user, the one-pass kernel
    for users without one - options predict_half, predict_int8) give the one-pass kernel's top-Kp bit for bit: fitted models, several Kp, history
    removal, an item filter, a pruned model (the fp16 copy is rebuilt), and a model of few distinct values whose ties at the
    boundary no list can prove (everybody takes the fallback)."""
    from recpack_amd.synth import powerlaw_csr

    _lib, ctx = hip
    X = powerlaw_csr(9000, 2600, 0.01, seed=91).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    m = _lib.ease_fit(ctx, dX, 30.0, 0.0)
    rng = np.random.default_rng(5)
    keep = (rng.random(X.shape[1]) < 0.7).astype(np.uint8)

    def both(model, Kp, excl, flt=None):
        out = []
        for half, i8 in ((0, 1), (1, 0), (1, 1)):  # one pass; first pass on fp16 rows; on int8 rows (the default)
            ctx.set_option("predict_half", half)
            ctx.set_option("predict_int8", i8)
            if flt is None:
                out.append(_lib.predict_topk(ctx, model, dX, Kp, exclude_history=excl))
            else:
                out.append(_lib.predict_topk(ctx, model, dX, Kp, exclude_history=excl, item_keep=flt))
        ctx.set_option("predict_half", 1)
        ctx.set_option("predict_int8", 1)
        i0, v0, n0 = out[0]
        for i1, v1, n1 in out[1:]:
            np.testing.assert_array_equal(n0, n1)
            np.testing.assert_array_equal(i0, i1)
            np.testing.assert_array_equal(v0.view(np.uint32), v1.view(np.uint32))
        return n0

    try:
        for Kp, excl in ((10, True), (100, True), (100, False), (300, True)):
            assert (both(m, Kp, excl) > 0).any()
        both(m, 50, True, keep)
        m.dense_prune(int(0.05 * 2600 * 2600))
        both(m, 100, True)
        T = rng.choice(np.array([0.0, 0.25, -0.25, 0.5, 1.0], dtype=np.float32), size=(2600, 2600), p=[0.5, 0.2, 0.1, 0.15, 0.05])
        both(_lib.Model.from_dense_host(ctx, T), 100, True)
        # rows of wildly different scale, all-zero rows, rows whose one outlier leaves every other entry below the int8 step: the
        # quantised copies prove little here, whoever is not proven takes the exact pass, the results stay the one-pass kernel's
        T2 = (rng.standard_normal((2600, 2600)) * 10.0 ** rng.uniform(-6, 6, size=(2600, 1))).astype(np.float32)
        T2[rng.random(2600) < 0.05] = 0.0
        out = rng.choice(2600, 200, replace=False)
        T2[out, rng.integers(0, 2600, 200)] = 1e9
        both(_lib.Model.from_dense_host(ctx, T2), 100, True)
        both(_lib.Model.from_dense_host(ctx, T2), 20, False)
        # a model that borrows its matrix, rewritten by its owner: dense_changed drops the quantised copy of the old values
        import torch

        Bt = torch.zeros((2600, 2624), dtype=torch.float32, device="cuda:0")
        Bt[:, :2600] = torch.as_tensor(T2, device="cuda:0")
        mb = _lib.Model.from_dense_cols(ctx, 2600, 0, 2600, Bt)
        both(mb, 50, True)
        Bt[:, :2600] = torch.as_tensor(rng.standard_normal((2600, 2600)).astype(np.float32), device="cuda:0")
        torch.cuda.synchronize()
        mb.dense_changed()
        both(mb, 50, True)
        mb.close()
        # tiles wide enough that every thread of the first pass owns columns: the pre-filter of the selection (a lower bound of
        # the Kc-th score from the threads' best scores) is active; and the narrow-tile instance (two workgroups per CU)
        X2 = powerlaw_csr(3000, 9000, 0.004, seed=92).to_scipy()
        dX = _lib.DeviceCSR.from_host(ctx, X2.indptr, X2.indices, X2.shape)
        m2 = _lib.ease_fit(ctx, dX, 30.0, 0.0)
        for tile in (0, 8960, 11200):
            ctx.set_option("predict_half_tile", tile)
            assert (both(m2, 100, True) > 0).any()
            both(m2, 20, False)
    finally:
        ctx.set_option("predict_half", 1)
        ctx.set_option("predict_int8", 1)
        ctx.set_option("predict_half_tile", 11200)
