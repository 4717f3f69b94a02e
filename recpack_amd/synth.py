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


class _NumpyOps:
    """numpy backend of the generator's integer steps."""

    @staticmethod
    def to_dev(a):
        return a

    @staticmethod
    def draw_keys(cdf, perm, need_users, r, n_items):
        rank = np.searchsorted(cdf, r, side="right")
        np.minimum(rank, n_items - 1, out=rank)
        return np.unique(need_users * n_items + perm[rank])

    @staticmethod
    def merge_new(have, new):
        if have.size:
            pos = np.searchsorted(have, new)
            pos[pos == have.size] = have.size - 1
            new = new[have[pos] != new]
            have = np.sort(np.concatenate([have, new]), kind="stable")  # two sorted runs: a merge
        else:
            have = new
        return have

    @staticmethod
    def missing(have, deg, n_items, n_users):
        miss = deg - np.bincount(have // n_items, minlength=n_users)
        mu = np.nonzero(miss)[0]
        return mu, (np.repeat(mu, miss[mu]) if mu.size else mu)

    @staticmethod
    def finish(have, n_items, n_users):
        indices = (have % n_items).astype(np.int32)
        cnt = np.bincount(have // n_items, minlength=n_users)
        indptr = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(cnt, out=indptr[1:])
        return indptr, indices


class _TorchOps:
    """torch backend (same integer steps on a GPU; the uniforms still come from numpy's PCG64 on the host)."""

    def __init__(self, device):
        import torch

        self.t = torch
        self.dev = device

    def to_dev(self, a):
        return self.t.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def draw_keys(self, cdf, perm, need_users, r, n_items):
        t = self.t
        rank = t.searchsorted(cdf, self.to_dev(r), right=True).clamp_(max=n_items - 1)
        return t.unique(need_users * n_items + perm[rank].to(t.int64))

    def merge_new(self, have, new):
        t = self.t
        if have.numel():
            pos = t.searchsorted(have, new).clamp_(max=have.numel() - 1)
            new = new[have[pos] != new]
            have = t.sort(t.cat([have, new]))[0]
        else:
            have = new
        return have

    def missing(self, have, deg, n_items, n_users):
        t = self.t
        miss = deg - t.bincount(t.div(have, n_items, rounding_mode="floor"), minlength=n_users)
        mu = t.nonzero(miss).flatten()
        return mu, (t.repeat_interleave(mu, miss[mu]) if mu.numel() else mu)

    def finish(self, have, n_items, n_users):
        t = self.t
        indices = (have % n_items).to(t.int32)
        cnt = t.bincount(t.div(have, n_items, rounding_mode="floor"), minlength=n_users)
        indptr = t.zeros(n_users + 1, dtype=t.int64, device=self.dev)
        t.cumsum(cnt, 0, out=indptr[1:])
        return indptr, indices


def powerlaw_csr(n_users: int, n_items: int, density: float, seed: int, sigma: float = 1.0,
                 zipf_alpha: float = 1.0, device=None):
    """Generate the matrix described in the module docstring.

    ``device=None``: numpy, returns :class:`SynthCSR`.  ``device="cuda:0"``: the sort/unique/search steps run
    in torch on that GPU (bit-identical result, seconds instead of minutes at 1M x 100K) and the function
    returns ``(indptr int32 CUDA tensor, indices int32 CUDA tensor)``.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    target = int(round(density * n_users * n_items))
    ops = _NumpyOps if device is None else _TorchOps(device)

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
    need_host = np.repeat(np.arange(n_users, dtype=np.int64), deg)
    n_need = need_host.shape[0]
    need_users = ops.to_dev(need_host)
    del need_host
    cdf_d, perm_d, deg_d = ops.to_dev(cdf), ops.to_dev(perm), ops.to_dev(deg)
    have = ops.to_dev(np.empty(0, dtype=np.int64))  # sorted unique keys user * n_items + item
    for _ in range(1000):
        r = rng.random(n_need)
        new = ops.draw_keys(cdf_d, perm_d, need_users, r, n_items)
        have = ops.merge_new(have, new)
        mu, need_users = ops.missing(have, deg_d, n_items, n_users)
        n_need = int(need_users.shape[0])
        if n_need == 0:
            break
    else:  # pragma: no cover - clipping to I/4 makes this practically unreachable
        raise RuntimeError("could not draw distinct items")
    # 'have' is sorted by key => rows sorted by item id
    indptr, indices = ops.finish(have, n_items, n_users)
    if device is not None:
        return indptr.to(indptr.new_empty(0, dtype=ops.t.int32).dtype), indices
    if indptr[-1] < 2**31:
        indptr = indptr.astype(np.int32)
    return SynthCSR(n_users, n_items, indptr, indices)


def config_csr(name: str, scale: float = 1.0, device=None):
    """Matrix of a BASELINE.json config; ``scale`` < 1 shrinks users and items alike (tests)."""
    u, i, dens = CONFIGS[name]
    idx = int(name[1:])
    if scale != 1.0:
        u, i = max(8, int(u * scale)), max(8, int(i * scale))
    return powerlaw_csr(u, i, dens if scale == 1.0 else min(0.5, dens / scale), CONFIG_SEED_BASE + idx, device=device)
