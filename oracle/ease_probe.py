"""Whole-matrix check of an EASE model without the inverse (TEST INFRASTRUCTURE ONLY: imported by ``tests/`` and by the parity
verdict ``bench.py`` prints after its timed steps - never by ``recpack_amd/``).

The reference computes B = I - P D^-1 with P = (X^T X + lambda I)^-1 in float64 and D = diag(P), the diagonal of B set to zero,
column j scaled by c_j^-alpha (/root/reference/recpack/algorithms/ease.py:81-90).  So, with A = X^T X + lambda I and
W = diag(c_j^-alpha):

    A (I - B W^-1) = D^-1                 (a DIAGONAL matrix),

and for any probe vector z without zeros   ( A (z - B (z / w)) )_j / z_j = 1 / P_jj   for every item j - the same number for
every probe.  Every entry of B takes part in B (z / w).  If a matrix B' with zero diagonal passes for two independent random
probes, A (B - B') W^-1 is diagonal, i.e. B' = I - P (D^-1 + Delta) W with Delta diagonal, and the zero diagonal forces Delta = 0:
B' is the reference's B.  The probe also checks lambda <= 1/P_jj <= c_j + lambda (P_jj >= 1/A_jj by the Schur complement,
P_jj <= 1/lambda_min(A) <= 1/lambda).

Cost: one float64 pass over B (read in strips of rows) and two sparse products per probe - seconds at 100 000 items.
"""
from __future__ import annotations

import numpy as np


class GramOps:
    """X v and X^T w for a binary CSR held as torch tensors, float64.  Preferred: two sparse-CSR x dense products (X and an
    explicitly built X^T, hipSPARSE on the GPU box); fallback where the sparse product is unavailable: gathers + index_add
    over chunks of ~2^24 entries."""

    def __init__(self, indptr_t, indices_t, shape, chunk_nnz=1 << 24, use_sparse=True):
        import torch

        self.torch = torch
        self.ip = indptr_t.to(torch.int64)
        self.ix = indices_t.to(torch.int64)
        self.U, self.I = shape
        dev = self.ix.device
        cnt = self.ip[1:] - self.ip[:-1]
        self.deg = torch.bincount(self.ix, minlength=self.I).to(torch.float64)
        self.A = self.At = None
        if use_sparse:
            try:
                ones = torch.ones(self.ix.numel(), dtype=torch.float64, device=dev)
                rows = torch.repeat_interleave(torch.arange(self.U, device=dev), cnt)
                perm = torch.argsort(self.ix, stable=True)
                tptr = torch.zeros(self.I + 1, dtype=torch.int64, device=dev)
                tptr[1:] = torch.cumsum(self.deg.to(torch.int64), 0)
                A = torch.sparse_csr_tensor(self.ip, self.ix, ones, size=(self.U, self.I))
                At = torch.sparse_csr_tensor(tptr, rows[perm], ones, size=(self.I, self.U))
                probe = torch.ones((self.I, 2), dtype=torch.float64, device=dev)
                got = At @ (A @ probe)
                assert torch.isfinite(got).all()
                self.A, self.At = A, At
                del rows, perm, probe, got
            except Exception as e:  # noqa: BLE001 - any failure of the sparse path: use the plain one
                print(f"ease_cg: sparse CSR products unavailable ({type(e).__name__}: {e}); using gathers + index_add")
                self.A = self.At = None
        if self.A is None:
            ipc = self.ip.cpu().numpy()
            cuts = [0]
            while cuts[-1] < self.U:
                nxt = int(np.searchsorted(ipc, ipc[cuts[-1]] + chunk_nnz, side="right")) - 1
                cuts.append(min(self.U, max(nxt, cuts[-1] + 1)))
            self.cuts = cuts
            self.ipc = ipc

    def rows_of(self, u0, u1):
        torch = self.torch
        cnt = (self.ip[u0 + 1:u1 + 1] - self.ip[u0:u1])
        return torch.repeat_interleave(torch.arange(u1 - u0, device=self.ix.device), cnt)

    def gram_apply(self, V):
        """(X^T X) V in float64, V [I, k]."""
        torch = self.torch
        if self.A is not None:
            return self.At @ (self.A @ V)
        out = torch.zeros_like(V)
        for u0, u1 in zip(self.cuts[:-1], self.cuts[1:]):
            a, b = int(self.ipc[u0]), int(self.ipc[u1])
            if a == b:
                continue
            cols = self.ix[a:b]
            rows = self.rows_of(u0, u1)
            W = torch.zeros((u1 - u0, V.shape[1]), dtype=V.dtype, device=V.device)
            W.index_add_(0, rows, V[cols])  # X v on this user chunk
            out.index_add_(0, cols, W[rows])  # X^T (.)
        return out


def ease_probe(indptr_t, indices_t, shape, B, l2: float, alpha: float = 0.0, n_probes: int = 2, seed: int = 7, strip_rows: int = 4096, ops=None):
    """``B``: torch float32 tensor [>= I, >= I] on the device of the CSR tensors (the model's matrix; rows / columns behind I are
    padding).  Returns a dict: ``probe_rel`` = max over items and probe pairs of |r_p[j] - r_0[j]| / |r_0[j]| with
    r_p = A (z_p - B (z_p / w)) / z_p; ``bounds_violation`` = the largest relative excursion of r_0 outside [lambda, c_j + lambda];
    ``diag_max`` = max |B_jj|; ``items_checked``."""
    import torch

    I = int(shape[1])
    dev = indices_t.device
    if ops is None:
        ops = GramOps(indptr_t, indices_t, shape)
    lam = float(np.float32(l2))  # (ease.py:82: lambda * identity(float32))
    deg = ops.deg
    if alpha != 0.0:
        w = torch.where(deg > 0, torch.clamp(deg, min=1.0) ** (-float(alpha)), torch.zeros_like(deg))
    else:
        w = torch.ones_like(deg)
    live = w > 0  # (alpha != 0: the reference's column of a never-seen item is NaN; it stays zero here - DESIGN section 4, deviation 1)
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    mag = 1.0 + torch.rand((I, n_probes), generator=g, dtype=torch.float64)
    sgn = (torch.rand((I, n_probes), generator=g, dtype=torch.float64) < 0.5).to(torch.float64) * 2.0 - 1.0
    Z = (mag * sgn).to(dev)
    Zw = torch.where(live[:, None], Z / torch.where(live, w, torch.ones_like(w))[:, None], torch.zeros_like(Z))
    BZ = torch.empty((I, n_probes), dtype=torch.float64, device=dev)
    for r0 in range(0, I, strip_rows):
        r1 = min(I, r0 + strip_rows)
        BZ[r0:r1] = B[r0:r1, :I].to(torch.float64) @ Zw
    diag_max = float(torch.diagonal(B)[:I].abs().max())
    Y = torch.where(live[:, None], Z, torch.zeros_like(Z)) - BZ
    R = (ops.gram_apply(Y) + lam * Y) / Z
    r0v = R[:, 0]
    rel = ((R[:, 1:] - r0v[:, None]).abs() / r0v.abs()[:, None]).amax(dim=1) if n_probes > 1 else torch.zeros_like(r0v)
    rel = torch.where(live, rel, torch.zeros_like(rel))
    lo, hi = torch.full_like(r0v, lam), deg + lam
    viol = torch.maximum((lo - r0v) / lo, (r0v - hi) / hi).clamp(min=0.0)
    viol = torch.where(live, viol, torch.zeros_like(viol))
    worst = int(torch.argmax(rel))
    return {"probe_rel": float(rel.max()), "probe_rel_median": float(rel.median()), "worst_item": worst, "worst_item_degree": int(deg[worst]),
            "bounds_violation": float(viol.max()), "diag_max": diag_max, "items_checked": int(live.sum()), "probes": int(n_probes)}
