import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
from recpack_amd import _lib
from recpack_amd.synth import powerlaw_csr
from oracle import oracle_np as onp
ctx = _lib.Context(0)
for (U, I, d, K) in [(4, 3, 0.6, 2), (2000, 500, 0.02, 10), (20000, 2000, 0.004, 200), (3000, 40000, 0.0008, 50)]:
    X = powerlaw_csr(U, I, d, seed=11).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    print("fit", U, I, K, flush=True)
    m = _lib.itemknn_fit(ctx, dX, K)
    idx, val, nnz = m.topk_download()
    ref = onp.itemknn_fit(X, m.K)
    bad = 0
    for r in range(I):
        o = np.argsort(ref["idx"][r, :ref["nnz"][r]])
        if nnz[r] != ref["nnz"][r] or not (idx[r, :nnz[r]] == ref["idx"][r, :nnz[r]][o]).all():
            bad += 1
            if bad < 4:
                print(" row", r, "deg", int(onp.item_degrees(onp.binarize(X))[r]), "nnz", nnz[r], ref["nnz"][r], idx[r, :8], ref["idx"][r, :ref["nnz"][r]][o][:8])
    print(" bad rows", bad, "counters", ctx.counters()[:2], flush=True)
