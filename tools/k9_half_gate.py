"""Measurement gate for a half-precision first pass of K9 (EASE predict): for sampled users of a config, the gap between the
Kp-th and the Kc-th exact score against the rigorous error bound of fp16-rounded rows, E_u = sum over the history of the row's
largest rounding error.  python tools/k9_half_gate.py [config [n_users]]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from recpack_amd import _lib
from recpack_amd import dist as rdist
from recpack_amd.synth import CONFIGS, config_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda", 0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
B = rdist.model_dense_tensor(m, dev)
# per-row scale-free fp16 rounding: values scaled by a power of two so that max |B| sits in [2^14, 2^15)
bmax = float(B.abs().max())
sc = 2.0 ** (14 - int(np.floor(np.log2(bmax))))
delta = torch.empty(I, dtype=torch.float32, device=dev)
for r0 in range(0, I, 4096):
    blk = B[r0:r0 + 4096]
    h = (blk * sc).to(torch.float16).to(torch.float32) / sc
    delta[r0:r0 + 4096] = (h - blk).abs().max(dim=1).values
print(f"{cfg}: max|B| {bmax:.3e}, scale 2^{int(np.log2(sc))}, row rounding error: median {float(delta.median()):.2e} max {float(delta.max()):.2e}")
ipc = ip.cpu().numpy().astype(np.int64)
ixc = ix.cpu().numpy()
rng = np.random.default_rng(1)
users = rng.choice(U, nu, replace=False)
Kp = 100
res = {kc: 0 for kc in (128, 160, 192, 256, 384, 512)}
ratios = []
for u in users:
    h = torch.as_tensor(ixc[ipc[u]:ipc[u + 1]], device=dev, dtype=torch.long)
    if h.numel() == 0:
        continue
    e = B[h].sum(dim=0)
    e[h] = -1e30
    Eu = float(delta[h].sum())
    top = torch.topk(e, 512).values
    for kc in res:
        # conservative: the Kc-th APPROX score is at most the Kc-th exact + E_u; need e_(Kp) > a_(Kc) + E_u
        if float(top[Kp - 1]) > float(top[kc - 1]) + 2.0 * Eu:
            res[kc] += 1
    ratios.append(Eu / max(1e-30, float(top[Kp - 1] - top[255])))
n = len(ratios)
print(f"{n} users, Kp = {Kp}: share verified without fallback by Kc: " + ", ".join(f"{kc}: {v / n:.4f}" for kc, v in res.items()))
r = np.array(ratios)
print(f"E_u / (e_100 - e_256): median {np.median(r):.3f}, 90 % {np.quantile(r, 0.9):.3f}, 99 % {np.quantile(r, 0.99):.3f}, max {r.max():.3f}")
