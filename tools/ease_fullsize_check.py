"""One-off parity check of the EASE fit at a large config (default C5: 200 000 items, 160 GB of B on one GPU): the optimality
condition of the closed form, (G + lambda I)(e_j - B[:, j]) = gamma_j e_j, on sampled columns in float64 on the host (the same
check tests/test_gpu_fullsize.py runs at C3).  usage: python tools/ease_fullsize_check.py [C5]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import scipy.sparse as sp
import torch

from recpack_amd import _lib
from recpack_amd import dist as rdist
from recpack_amd.synth import CONFIGS, config_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
ip, ix = config_csr(cfg, device=dev)
U, I, _ = CONFIGS[cfg]
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
l2 = 200.0
t0 = time.perf_counter()
m = _lib.ease_fit(ctx, dX, l2)
ctx.sync()
print(f"{cfg}: fit {time.perf_counter() - t0:.1f} s", {k: round(v["ms"] / 1e3, 2) for k, v in ctx.timings().items() if v["launches"]}, flush=True)
B = rdist.model_dense_tensor(m, dev)
assert float(torch.diagonal(B).abs().max()) == 0.0
indptr, indices = ip.cpu().numpy().astype(np.int64), ix.cpu().numpy()
X = sp.csr_matrix((np.ones(len(indices), dtype=np.float64), indices, indptr), shape=(U, I))
deg = np.bincount(indices, minlength=I)
rng = np.random.default_rng(3)
order = np.argsort(-deg)
cols = np.unique(np.concatenate([order[:4], order[I // 2 - 2:I // 2 + 2], rng.choice(I, 4, replace=False)]))
Bc = B[:, torch.as_tensor(cols, device=dev)].double().cpu().numpy()
Y = -Bc
Y[cols, np.arange(len(cols))] += 1.0
Z = X.T @ (X @ Y) + float(np.float32(l2)) * Y
diag = Z[cols, np.arange(len(cols))]
off = Z.copy()
off[cols, np.arange(len(cols))] = 0.0
rel = np.abs(off).max(axis=0) / diag
print(f"{cfg}: optimality residual of {len(cols)} sampled columns, max |off-diagonal| / gamma_j = {rel.max():.3e} (all gamma_j > 0: {(diag > 0).all()})")
