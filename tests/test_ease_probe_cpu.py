"""The whole-matrix EASE probe (oracle/ease_probe.py) on the CPU: the float64 oracle and the reference's own golden B pass, a
perturbed tile, a wrong column scale and a non-zero diagonal do not."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from oracle.ease_probe import ease_probe
from conftest import csr_from, load_golden


def _tensors(X):
    return torch.as_tensor(X.indptr.astype(np.int64)), torch.as_tensor(X.indices.astype(np.int64))


@pytest.mark.parametrize("l2,alpha", [(20.0, 0.0), (200.0, 0.0), (1000.0, 0.5), (10.0, 1.0)])
def test_probe_accepts_the_oracle_and_rejects_damage(l2, alpha):
    from recpack_amd.synth import powerlaw_csr

    X = powerlaw_csr(2500, 600, 0.02, seed=3).to_scipy()
    X = X[:, : X.shape[1] - 3].tocsr()  # (nothing forces every item to be seen)
    ip, ix = _tensors(X)
    B = np.nan_to_num(onp.ease_fit(X, l2, alpha))
    good = ease_probe(ip, ix, X.shape, torch.as_tensor(B.astype(np.float32)), l2, alpha)
    assert good["probe_rel"] <= 2e-6 and good["bounds_violation"] <= 1e-6 and good["diag_max"] == 0.0, good
    exact = ease_probe(ip, ix, X.shape, torch.as_tensor(B), l2, alpha)
    assert exact["probe_rel"] <= 1e-12, exact
    tile = B.astype(np.float32).copy()
    tile[64:80, 300:332] *= 1.02  # one small tile, 2 % off
    assert ease_probe(ip, ix, X.shape, torch.as_tensor(tile), l2, alpha)["probe_rel"] > 1e-4
    col = B.astype(np.float32).copy()
    col[:, 17] *= 1.01  # one column scaled by 1 %
    assert ease_probe(ip, ix, X.shape, torch.as_tensor(col), l2, alpha)["probe_rel"] > 1e-4
    other = np.nan_to_num(onp.ease_fit(X, l2 * 1.05, alpha)).astype(np.float32)  # the right form with another lambda
    r = ease_probe(ip, ix, X.shape, torch.as_tensor(other), l2, alpha)
    assert r["probe_rel"] > 1e-4, r
    d = B.astype(np.float32).copy()
    d[5, 5] = 1e-3
    assert ease_probe(ip, ix, X.shape, torch.as_tensor(d), l2, alpha)["diag_max"] > 0.0


def test_probe_accepts_the_reference_golden():
    g = load_golden("syn_ease_2000x500")
    X = csr_from(g, "X")
    ip, ix = _tensors(X)
    for n, (l2, alpha) in enumerate(zip(g["l2"], g["alpha"])):
        if f"B{n}" not in g:
            continue
        B = np.nan_to_num(np.asarray(g[f"B{n}"], dtype=np.float64))
        r = ease_probe(ip, ix, X.shape, torch.as_tensor(B), float(l2), float(alpha))
        assert r["probe_rel"] <= 1e-9 and r["bounds_violation"] <= 1e-9, (l2, alpha, r)
