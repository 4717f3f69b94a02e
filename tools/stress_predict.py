"""Repeat predict_topk on one model and compare the runs bit for bit (race hunting)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from recpack_amd import _lib
from recpack_amd.synth import powerlaw_csr

U, I, dens, K, Kp, seed = 2500, 33000, 0.001, 40, 20, 22
if len(sys.argv) > 1:
    U, I, dens, K, Kp = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
X = powerlaw_csr(U, I, dens, seed=seed).to_scipy()
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
m = _lib.itemknn_fit(ctx, dX, K)
bad = 0
for excl in (False, True):
    ref = None
    for it in range(40):
        pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl, user_begin=0, user_end=U)
        if ref is None:
            ref = (pi.copy(), pv.copy(), pn.copy())
            continue
        d = np.nonzero((pi != ref[0]).any(axis=1) | (pn != ref[2]) | (pv != ref[1]).any(axis=1))[0]
        if len(d):
            bad += 1
            u = d[0]
            print(f"excl={excl} run {it}: {len(d)} users differ; first user {u} deg {X.indptr[u+1]-X.indptr[u]} nnz {pn[u]} vs {ref[2][u]}")
            print("  idx", pi[u][:12], "\n  ref", ref[0][u][:12])
            print("  val", pv[u][:6], "\n  ref", ref[1][u][:6])
print("bad runs:", bad)
