"""The N > 1 path on CPU: world_size-2 gloo processes shard item rows / users exactly as the GPU ranks do.

The compute of a shard is injected (the CPU oracle stands in for the HIP kernels here - tests may use it);
what is under test is recpack_amd.dist: cost-balanced contiguous partitions, the padded all-gather that
assembles the full neighbour lists on every rank, and user sharding for predict.
"""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_np as onp
from recpack_amd import dist as rdist
from recpack_amd.synth import powerlaw_csr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, K, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = powerlaw_csr(3000, 700, 0.01, seed=99).to_scipy()
        deg = np.bincount(X.indices, minlength=X.shape[1])
        P = int((np.diff(X.indptr).astype(np.int64) ** 2).sum())
        parts = rdist.partition_by_cost(rdist.item_row_cost(deg, X.nnz, P), world)
        b, e = parts[rank]
        loc = onp.itemknn_fit(X, K, rows=np.arange(b, e))  # this rank's item rows
        # device order: columns ascending inside a row
        idx, val, nnz = loc["idx"].copy(), loc["val"].copy(), loc["nnz"]
        for r in range(idx.shape[0]):
            o = np.argsort(idx[r, : nnz[r]])
            idx[r, : nnz[r]], val[r, : nnz[r]] = idx[r, : nnz[r]][o], val[r, : nnz[r]][o]
        full = rdist.allgather_topk(torch.from_numpy(idx), torch.from_numpy(val), torch.from_numpy(nnz), parts)
        idx_f, val_f, nnz_f = (t.numpy() for t in full)
        # predict: users sharded, S replicated, no collective
        uparts = rdist.partition_users(X.indptr, world)
        ub, ue = uparts[rank]
        S = onp.topk_to_csr(idx_f, val_f, nnz_f, X.shape[1])
        scores = onp.predict_full(X[ub:ue], S)
        pidx, pval, pnnz = onp.topk_rows(scores, 20)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx_f, val=val_f, nnz=nnz_f, pidx=pidx, pval=pval, pnnz=pnnz,
                 ub=ub, ue=ue, parts=np.asarray(parts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fit_and_predict_match_single_process(tmp_path, world):
    K = 30
    port = _free_port()
    mp.spawn(_worker, args=(world, port, K, str(tmp_path)), nprocs=world, join=True)
    X = powerlaw_csr(3000, 700, 0.01, seed=99).to_scipy()
    ref = onp.itemknn_fit(X, K)
    S_ref = onp.topk_to_csr(ref["idx"], ref["val"], ref["nnz"], X.shape[1])
    ridx, rval, rnnz = onp.topk_rows(onp.predict_full(X, S_ref), 20)
    covered = np.zeros(X.shape[0], dtype=bool)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        S = onp.topk_to_csr(g["idx"], g["val"], g["nnz"], X.shape[1])
        assert (S != S_ref).nnz == 0, "every rank must hold the same full similarity matrix"
        ub, ue = int(g["ub"]), int(g["ue"])
        np.testing.assert_array_equal(g["pidx"], ridx[ub:ue])
        np.testing.assert_allclose(g["pval"], rval[ub:ue], rtol=1e-12)
        covered[ub:ue] = True
        parts = g["parts"]
        assert parts[0][0] == 0 and parts[-1][1] == X.shape[1] and (parts[1:, 0] == parts[:-1, 1]).all()
    assert covered.all()


def test_partition_by_cost_balances_power_law():
    rng = np.random.default_rng(0)
    cost = 1.0 / np.arange(1, 100001) ** 1.0
    rng.shuffle(cost)
    for n in (1, 2, 4, 8):
        parts = rdist.partition_by_cost(cost, n)
        assert len(parts) == n and parts[0][0] == 0 and parts[-1][1] == cost.shape[0]
        sums = np.array([cost[b:e].sum() for b, e in parts])
        assert sums.max() <= 1.25 * cost.sum() / n + cost.max()
    # degenerate inputs
    assert rdist.partition_by_cost(np.zeros(5), 2) == [(0, 2), (2, 5)]
    assert rdist.partition_by_cost(np.zeros(0), 3) == [(0, 0), (0, 0), (0, 0)]
