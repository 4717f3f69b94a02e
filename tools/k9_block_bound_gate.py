"""Gate for a block upper-bound prefilter of K9: for sampled users, the share of column blocks whose bound sum_i max_{j in block} B_ij
reaches the user's Kp-th best exact score (only those blocks would have to be read).  python tools/k9_block_bound_gate.py [config]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from recpack_amd import _lib
from recpack_amd import dist as rdist
from recpack_amd.synth import CONFIGS, config_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda", 0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
B = rdist.model_dense_tensor(m, dev)
ipc = ip.cpu().numpy().astype(np.int64)
ixc = ix.cpu().numpy()
rng = np.random.default_rng(1)
users = rng.choice(U, 300, replace=False)
for bw in (16, 64, 256):
    nbk = (I + bw - 1) // bw
    pad = nbk * bw - I
    shares = []
    for u in users:
        h = torch.as_tensor(ixc[ipc[u]:ipc[u + 1]], device=dev, dtype=torch.long)
        if h.numel() == 0:
            continue
        rows = B[h]
        e = rows.sum(dim=0)
        e[h] = -1e30
        thr = float(torch.topk(e, 100).values[-1])
        rp = torch.nn.functional.pad(rows, (0, pad), value=-1e30) if pad else rows
        ub = rp.view(h.numel(), nbk, bw).max(dim=2).values.sum(dim=0)
        shares.append(float((ub >= thr).float().mean()))
    s = np.array(shares)
    print(f"{cfg} block {bw}: share of blocks to read: median {np.median(s):.3f}, mean {s.mean():.3f}, 90 % {np.quantile(s, 0.9):.3f}")
