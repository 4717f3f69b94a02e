"""Round 6 gate: how sparse are the int8 rows K9's first pass streams?  q = rint(b * 127 / rowmax) per row (k_dense_to_i8's rule).
Prints the share of zero entries and of all-zero 16-byte / 64-byte blocks, in the natural column order and with the columns sorted by
item popularity, overall and weighted by how often a row is read (its item's degree).  usage: python tools/k9_sparsity_gate.py [C3|C4]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from recpack_amd import _lib, dist as rdist  # noqa: E402
from recpack_amd.synth import CONFIGS, config_csr  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
B = rdist.model_dense_tensor(m, dev)
deg = torch.bincount(ix.long(), minlength=I).float()
order = torch.argsort(-deg)
Ipad = (I + 63) // 64 * 64
tot = {k: 0.0 for k in ("zero", "z16", "z64", "z16p", "z64p", "abs1")}
wt = {k: 0.0 for k in tot}
wsum = 0.0
rows = 0
for r0 in range(0, I, 2048):
    A = B[r0:r0 + 2048]
    mxr = A.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    q = torch.round(A * (127.0 / mxr)).to(torch.int8)
    w = deg[r0:r0 + 2048]
    nz = (q != 0)
    def blocks(nzm, bs):
        pad = torch.zeros((nzm.shape[0], Ipad - I), dtype=torch.bool, device=dev)
        z = torch.cat([nzm, pad], 1).view(nzm.shape[0], -1, bs).any(dim=2)
        return 1.0 - z.float().mean(dim=1)  # share of all-zero blocks per row
    per = {"zero": 1.0 - nz.float().mean(dim=1), "abs1": (q.abs() <= 1).float().mean(dim=1), "z16": blocks(nz, 16), "z64": blocks(nz, 64),
           "z16p": blocks(nz[:, order], 16), "z64p": blocks(nz[:, order], 64)}
    for k, v in per.items():
        tot[k] += float(v.sum())
        wt[k] += float((v * w).sum())
    wsum += float(w.sum())
    rows += A.shape[0]
    del A, q, nz
print(cfg, "rows", rows)
for k in tot:
    print(f"  {k:5s}: mean over rows {tot[k] / rows:.4f}   weighted by how often the row is read {wt[k] / wsum:.4f}", flush=True)
