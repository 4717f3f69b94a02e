"""Round 6 gate: could K9's first pass stop early?  Columns sorted by an upper bound of |score| - a user's score at column c is at most
the sum of the d_u largest |B[h][c]| of that column, S_c(d_u), and cruder d_u * max_h |B[h][c]| - and a user's pass ends once the bound of
every remaining column is below his Kc-th best approximate score.  Measures, on the fitted model of a config, for sampled users: the
share of the columns (in bound order) a pass would still have to read.  usage: python tools/k9_colbound_gate.py [C3|C4]"""
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
Kc = 192
absB = None
ds = (16, 32, 64, 128, 256)
S = {d: torch.zeros(I, device=dev) for d in ds}
mx = torch.zeros(I, device=dev)
for c0 in range(0, I, 4096):  # column strips: the d largest magnitudes of every column
    A = B[:, c0:c0 + 4096].abs()
    top = torch.topk(A, 256, dim=0).values  # [256, strip]
    cs = torch.cumsum(top, 0)
    for d in ds:
        S[d][c0:c0 + 4096] = cs[d - 1]
    mx[c0:c0 + 4096] = top[0]
    del A, top, cs
g = torch.Generator(device="cpu").manual_seed(3)
users = torch.randint(0, U, (400,), generator=g).tolist()
lens = (ip[1:] - ip[:-1])
res = {"crude": [], "topd": []}
for u in users:
    a, b = int(ip[u]), int(ip[u + 1])
    d_u = b - a
    if d_u == 0:
        continue
    h = ix[a:b].long()
    sc = B[h].sum(0)
    sc[h] = -1e30
    aK = float(torch.topk(sc, Kc).values[-1])
    if aK <= 0:
        continue
    d_b = next((d for d in ds if d >= d_u), None)
    res["crude"].append(float((d_u * mx >= aK).float().mean()))
    if d_b is not None:
        res["topd"].append(float((S[d_b] >= aK).float().mean()))
for k, v in res.items():
    t = torch.tensor(v)
    print(f"{cfg} bound {k}: users {len(v)}, share of columns still to read: mean {t.mean():.3f} median {t.median():.3f} p90 {t.quantile(0.9):.3f} max {t.max():.3f}", flush=True)
# weighted by history length (the pass's bytes are d_u * columns read)
