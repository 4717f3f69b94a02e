"""Gate for an 8-bit first pass of K9: rigorous error bound of int8 rows (one scale per row, round to nearest) against the score gaps.
python tools/k9_int8_gate.py [config [n_users]]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from recpack_amd import _lib
from recpack_amd import dist as rdist
from recpack_amd.synth import CONFIGS, config_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = torch.device("cuda", 0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
B = rdist.model_dense_tensor(m, dev)
rowmax = torch.empty(I, dtype=torch.float32, device=dev)
for r0 in range(0, I, 4096):
    rowmax[r0:r0 + 4096] = B[r0:r0 + 4096].abs().max(dim=1).values
delta = rowmax / 127.0 * 0.5  # round to nearest on a per-row grid of 127 steps
ipc = ip.cpu().numpy().astype(np.int64)
ixc = ix.cpu().numpy()
rng = np.random.default_rng(1)
users = rng.choice(U, nu, replace=False)
Kp = 100
kcs = (256, 384, 512, 768, 1024, 2048)
res = {kc: 0 for kc in kcs}
n = 0
for u in users:
    h = torch.as_tensor(ixc[ipc[u]:ipc[u + 1]], device=dev, dtype=torch.long)
    if h.numel() == 0:
        continue
    n += 1
    e = B[h].sum(dim=0)
    e[h] = -1e30
    Eu = float(delta[h].sum())
    top = torch.topk(e, 2048).values
    for kc in kcs:
        if float(top[Kp - 1]) > float(top[kc - 1]) + 2.0 * Eu:
            res[kc] += 1
print(f"{cfg}: {n} users, Kp = {Kp}, int8 rows with a per-row scale: share provable by Kc: " + ", ".join(f"{kc}: {v / n:.4f}" for kc, v in res.items()))
