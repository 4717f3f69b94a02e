"""Deterministic synthetic implicit-feedback matrices (power-law users and items).

This is the workload generator of SURVEY.md §8(d): the reference ships no benchmark
data, so every config in BASELINE.json is run on matrices produced here.

* user activity   d_u ~ lognormal(mu=0, sigma=1), rescaled so that sum(d_u) is about
  density*U*I, clipped to [1, I/4];
* item popularity Zipf(alpha=1) over a seeded random permutation of the item ids;
* every user draws d_u *distinct* items by inverse-CDF sampling (duplicates are redrawn);
* values are binary, rows are sorted by item id (what scipy's COO->CSR gives recpack in
  ``InteractionMatrix.values``, /root/reference recpack/matrix/interaction_matrix.py:202-217).

Only ``numpy.random.Generator(PCG64(seed)).random()`` uniforms are consumed, so the
stream of indices is stable across numpy versions.  Seeds: 20260000 + config index.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass

import numpy as np

CONFIG_SEED_BASE = 20260000

# BASELINE.json configs -> (users, items, density); C1 is ML-100K shaped (943 x 1682, ~55K positives).
CONFIGS = {
    "C1": (943, 1682, 55_000 / (943 * 1682)),
    "C2": (100_000, 20_000, 0.001),
    "C3": (200_000, 50_000, 0.001),
    "C4": (1_000_000, 100_000, 0.0005),
    "C5": (5_000_000, 200_000, 0.00025),
}


@dataclass
class SynthCSR:
    """Binary CSR interaction matrix: ``indptr`` int64[U+1] (int32 when it fits), ``indices`` int32[nnz]."""

    n_users: int
    n_items: int
    indptr: np.ndarray
    indices: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.indices.shape[0])

    def to_scipy(self):
        from scipy.sparse import csr_matrix

        data = np.ones(self.nnz, dtype=np.int32)
        return csr_matrix((data, self.indices, self.indptr), shape=(self.n_users, self.n_items))

    def crc32(self) -> int:
        return zlib.crc32(self.indices.tobytes()) & 0xFFFFFFFF

    def sum_sq_degrees(self) -> int:
        """P = sum_u d_u^2, the number of scalar products in the Gram (SURVEY.md §8 notation)."""
        d = np.diff(self.indptr).astype(np.int64)
        return int((d * d).sum())


def _normal_from_uniform(u1: np.ndarray, u2: np.ndarray) -> np.ndarray:
    # Box-Muller; u1 in (0,1]
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def powerlaw_csr(n_users: int, n_items: int, density: float, seed: int, sigma: float = 1.0,
                 zipf_alpha: float = 1.0) -> SynthCSR:
    """Generate the matrix described in the module docstring."""
    rng = np.random.Generator(np.random.PCG64(seed))
    target = int(round(density * n_users * n_items))

    # --- user degrees
    z = _normal_from_uniform(rng.random(n_users), rng.random(n_users))
    w = np.exp(sigma * z)
    d_max = max(1, n_items // 4)
    deg = np.clip(np.rint(w * (target / w.sum())), 1, d_max).astype(np.int64)

    # --- item popularity: Zipf over a random permutation (argsort of uniforms)
    perm = np.argsort(rng.random(n_items), kind="stable").astype(np.int32)
    pop = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_alpha)
    cdf = np.cumsum(pop)
    cdf /= cdf[-1]

    # draw with replacement, then redraw the duplicates until every user holds d_u distinct items
    need_users = np.repeat(np.arange(n_users, dtype=np.int64), deg)
    have = np.empty(0, dtype=np.int64)  # sorted unique keys user * n_items + item
    for _ in range(1000):
        r = rng.random(need_users.shape[0])
        rank = np.searchsorted(cdf, r, side="right")
        np.minimum(rank, n_items - 1, out=rank)
        new = np.unique(need_users * n_items + perm[rank])
        if have.size:
            pos = np.searchsorted(have, new)
            pos[pos == have.size] = have.size - 1
            new = new[have[pos] != new]
            have = np.sort(np.concatenate([have, new]), kind="stable")  # two sorted runs: a merge
        else:
            have = new
        miss = deg - np.bincount(have // n_items, minlength=n_users)
        mu = np.nonzero(miss)[0]
        if mu.size == 0:
            break
        need_users = np.repeat(mu, miss[mu])
    else:  # pragma: no cover - clipping to I/4 makes this practically unreachable
        raise RuntimeError("could not draw distinct items")
    # 'have' is sorted by key => rows sorted by item id
    indices = (have % n_items).astype(np.int32)
    cnt = np.bincount(have // n_items, minlength=n_users)
    indptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(cnt, out=indptr[1:])
    if indptr[-1] < 2**31:
        indptr = indptr.astype(np.int32)
    return SynthCSR(n_users, n_items, indptr, indices)


def config_csr(name: str, scale: float = 1.0) -> SynthCSR:
    """Matrix of a BASELINE.json config; ``scale`` < 1 shrinks users and items alike (tests)."""
    u, i, dens = CONFIGS[name]
    idx = int(name[1:])
    if scale != 1.0:
        u, i = max(8, int(u * scale)), max(8, int(i * scale))
    return powerlaw_csr(u, i, dens if scale == 1.0 else min(0.5, dens / scale), CONFIG_SEED_BASE + idx)
