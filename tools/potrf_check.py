"""k_potrf_trtri (the 128 x 128 diagonal block of the blocked Cholesky: float64 factor + inverse in LDS) against numpy float64:
    python tools/potrf_check.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
rng = np.random.default_rng(5)
ok = True
for trial, cond in enumerate((1e1, 1e3, 1e5)):
    Q, _ = np.linalg.qr(rng.standard_normal((128, 128)))
    ev = np.logspace(0, np.log10(cond), 128)
    S = (Q * ev) @ Q.T
    S = ((S + S.T) / 2).astype(np.float32)
    A = torch.tensor(S, device=dev)
    W = torch.zeros(128, 128, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.potrf_trtri_128(ctx, A, W, info)
    torch.cuda.synchronize()
    L = np.linalg.cholesky(S.astype(np.float64))
    Wr = np.linalg.inv(L)
    eL = np.abs(A.cpu().numpy() - L).max() / np.abs(L).max()
    eW = np.abs(W.cpu().numpy() - Wr).max() / np.abs(Wr).max()
    up = float(np.abs(np.triu(A.cpu().numpy(), 1)).max() + np.abs(np.triu(W.cpu().numpy(), 1)).max())
    print(f"cond {cond:.0e}: |L - L64| {eL:.1e}, |W - W64| {eW:.1e}, upper triangles {up}, info {int(info.item())}")
    ok = ok and eL < 2e-7 and eW < 2e-7 and up == 0.0 and int(info.item()) == 0
# a non-positive pivot is reported
S = np.eye(128, dtype=np.float32)
S[40, 40] = -1.0
A = torch.tensor(S, device=dev)
_lib.potrf_trtri_128(ctx, A, W, info)
torch.cuda.synchronize()
print("info for a negative pivot at 40:", int(info.item()))
ok = ok and int(info.item()) == 41
A = torch.tensor((Q * ev) @ Q.T, device=dev, dtype=torch.float32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    _lib.potrf_trtri_128(ctx, A.clone(), W, info)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 200 * 1e6:.0f} us per call (with a clone and the launch)")
sys.exit(0 if ok else 1)
