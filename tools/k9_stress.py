"""Randomised equality stress of K9: the two-pass predict (fp16 first pass with the threshold pre-filter, narrow or wide tiles,
exact rescoring, proof, fallback) against the one-pass kernel, bit for bit, on random dense models of several value distributions.
    python tools/k9_stress.py [iterations [seed]]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import powerlaw_csr  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _lib.Context(0)
bad = 0
for it in range(iters):
    I = int(rng.choice([600, 2600, 9000, 12000, 20000, 26000]))
    U = int(rng.integers(50, 1500))
    dens = float(rng.choice([0.002, 0.01, 0.03]))
    X = powerlaw_csr(U, I, dens, seed=int(rng.integers(1 << 30))).to_scipy()
    dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
    kind = int(rng.integers(4))
    if kind == 0:
        B = rng.standard_normal((I, I), dtype=np.float32) * 0.01
    elif kind == 1:  # heavy tail
        B = (rng.standard_normal((I, I), dtype=np.float32) ** 3) * 0.003
    elif kind == 2:  # few distinct values: ties at the boundary
        B = rng.choice(np.array([0.0, 0.25, -0.25, 0.5, 1.0], dtype=np.float32), size=(I, I), p=[0.6, 0.15, 0.1, 0.1, 0.05])
    else:  # mostly zero
        B = np.where(rng.random((I, I), dtype=np.float32) < 0.02, rng.standard_normal((I, I), dtype=np.float32), 0).astype(np.float32)
    np.fill_diagonal(B, 0.0)
    m = _lib.Model.from_dense_host(ctx, B)
    del B
    Kp = int(rng.choice([1, 10, 50, 100, 200]))
    if 4 * (((Kp + 24 + 63) // 64) * 64) > I:
        Kp = 10
    excl = bool(rng.integers(2))
    tile = int(rng.choice([0, 11200, 8960, 6400]))
    ctx.set_option("predict_half_tile", tile)
    out = []
    for half in (0, 1):
        ctx.set_option("predict_half", half)
        out.append(_lib.predict_topk(ctx, m, dX, Kp, exclude_history=excl))
    (i0, v0, n0), (i1, v1, n1) = out
    same = np.array_equal(n0, n1) and np.array_equal(i0, i1) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    if not same:
        bad += 1
        print(f"MISMATCH it={it} U={U} I={I} dens={dens} kind={kind} Kp={Kp} excl={excl} tile={tile}", flush=True)
    m.close()
    dX.close()
ctx.set_option("predict_half", 1)
print(f"{iters} random cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
