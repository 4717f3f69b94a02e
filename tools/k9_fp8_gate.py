"""Gate for an fp8 (e4m3, one scale per row) first pass of K9: the candidate list is chosen by a per-item UPPER BOUND
u_j = a_j + eps * A_j + sum of the rows' subnormal half-steps (a_j the approximate score, A_j the sum of |quantised entries|: one more
accumulator in the first pass); the list proves the user's exact top-Kp when its (Kc+1)-th bound is below the exact Kp-th score.
Prints the share of sampled users a list of Kc candidates proves, next to the same test for int8 rows (round to nearest, bound
sum of half-steps) and the fp16 copy the library uses.
python tools/k9_fp8_gate.py [config [n_users]]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from recpack_amd import _lib
from recpack_amd import dist as rdist
from recpack_amd.synth import CONFIGS, config_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
B = rdist.model_dense_tensor(m, dev)[:, :I]
rowmax = torch.empty(I, dtype=torch.float32, device=dev)
for r0 in range(0, I, 4096):
    rowmax[r0:r0 + 4096] = B[r0:r0 + 4096].abs().max(dim=1).values
rowmax.clamp_(min=1e-30)
ipc = ip.cpu().numpy().astype(np.int64)
ixc = ix.cpu().numpy()
rng = np.random.default_rng(1)
users = rng.choice(U, nu, replace=False)
Kp = 100
kcs = (128, 192, 256, 320, 384, 512, 768)
fmts = ("fp16", "e4m3", "e4m3_p2", "int8")
res = {f: {kc: 0 for kc in kcs} for f in fmts}
need = {f: [] for f in fmts}
n = 0
e4_max, e4_eps, e4_sub = 448.0, 2.0 ** -4 * (1 + 2.0 ** -4), 2.0 ** -10  # largest normal; relative half-ulp of a normal (of |q|); half a subnormal step
g16 = float(2.0 ** np.floor(np.log2(60000.0 / float(rowmax.max()))))  # the library's fp16 copy: one power-of-two scale
for u in users:
    h = torch.as_tensor(ixc[ipc[u]:ipc[u + 1]], device=dev, dtype=torch.long)
    if h.numel() == 0:
        continue
    n += 1
    R = B[h]
    e = R.sum(dim=0)
    e[h] = -1e30
    tau = float(torch.topk(e, Kp).values[-1])
    ub = {}
    # fp16 copy, global scale: error per entry <= 2^-11 |b| (normal) -> the library bounds it per row by 2^-11 rowmax
    q = (R * g16).half().float() / g16
    ub["fp16"] = q.sum(dim=0) + float((rowmax[h] * 2.0 ** -11).sum())
    # e4m3, scale rowmax -> 448
    s = (rowmax[h] / e4_max).unsqueeze(1)
    q = (R / s).to(torch.float8_e4m3fn).float()
    ub["e4m3"] = (q * s).sum(dim=0) + e4_eps * (q.abs() * s).sum(dim=0) + float((s * e4_sub).sum())
    # ... with a power-of-two scale (the decode is then exact in f32)
    s2 = torch.exp2(torch.ceil(torch.log2(rowmax[h] / e4_max))).unsqueeze(1)
    q = (R / s2).to(torch.float8_e4m3fn).float()
    ub["e4m3_p2"] = (q * s2).sum(dim=0) + e4_eps * (q.abs() * s2).sum(dim=0) + float((s2 * e4_sub).sum())
    # int8, round to nearest, step rowmax / 127
    s8 = (rowmax[h] / 127.0).unsqueeze(1)
    q = torch.round(R / s8)
    ub["int8"] = (q * s8).sum(dim=0) + float((s8 * 0.5).sum())
    for f in fmts:
        v = ub[f].clone()
        v[h] = -1e30
        cnt = int((v >= tau).sum())  # every item whose bound reaches the exact Kp-th score must be in the list
        need[f].append(cnt)
        for kc in kcs:
            if cnt <= kc:
                res[f][kc] += 1
for f in fmts:
    a = np.array(need[f])
    print(f"{cfg} {f:8s}: list length needed: median {int(np.median(a))}, 90 % {int(np.quantile(a, 0.9))}, 99 % {int(np.quantile(a, 0.99))}, max {a.max()}; "
          "share of users proved by Kc: " + ", ".join(f"{kc}: {v / n:.3f}" for kc, v in res[f].items()), flush=True)
