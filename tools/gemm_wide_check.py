"""The 256 x 128 three-stage plane product kernel (k_gemm_hp16w, option gemm_wide) against the 128 x 128 one and float64: plain,
K-clipped and lower-only products on shapes with an odd number of 128-row tiles; then the 8192^3 rate of both."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from recpack_amd import _lib

if os.environ.get('RP_LIB'):
    _lib.LIB_PATH = os.environ['RP_LIB']  # (diagnostic builds)

ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(7)
ok = True
for (M, N, K) in ((3200, 2944, 2048), (4096, 4096, 4096), (2944, 3072, 1024)):
    A = (torch.randn((M, K), generator=g) * torch.exp(1.5 * torch.randn((M, 1), generator=g))).to(dev)
    B = torch.randn((N, K), generator=g).to(dev)
    C0 = torch.randn((M, N), generator=g).to(dev)
    ref = (0.5 * (A.double() @ B.double().T) - C0.double())
    scale = 0.5 * (A.double().abs() @ B.double().abs().T) + C0.double().abs()
    out = {}
    for wide in (0, 1):
        ctx.set_option("gemm_wide", wide)
        C = C0.clone()
        _lib.gemm_nt(ctx, A, B, C, alpha=0.5, beta=-1.0)
        ctx.sync()
        out[wide] = C
        err = float(((C.double() - ref).abs() / scale).max())
        print(f"plain {M}x{N}x{K} wide={wide}: error relative to sum |a||b| = {err:.2e}", flush=True)
        ok = ok and err < 5e-7
    same = float((out[0] - out[1]).abs().max())
    print(f"   wide vs narrow: max |difference| = {same:.3e}", flush=True)
    # K from the row (upper-triangular A), lower tiles only: the clipped and skipped tiles of both kernels
    if M == N:
        for kw in (dict(k_from_row=True, m_off=0), dict(k_from_row=True, k_from_col=True, m_off=0, lower_only=True), dict(lower_only=True)):
            res = {}
            for wide in (0, 1):
                ctx.set_option("gemm_wide", wide)
                C = C0.clone()
                _lib.gemm_nt_ex(ctx, A, B, C, 0, K, alpha=0.5, beta=-1.0, **kw)
                ctx.sync()
                res[wide] = C
            d = float((res[0] - res[1]).abs().max())
            print(f"   {kw}: wide vs narrow max |difference| = {d:.3e}", flush=True)
            ok = ok and d < 1e-3 * float(res[0].abs().max())
n = 8192
A = torch.randn((n, n), dtype=torch.float32, device=dev)
C = torch.empty((n, n), dtype=torch.float32, device=dev)
for wide in (0, 1):
    ctx.set_option("gemm_wide", wide)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.gemm_nt(ctx, A, A, C)
        ctx.sync()
        dt = time.perf_counter() - t0
    print(f"wide={wide} {n}^3: {dt * 1e3:.1f} ms = {2 * n ** 3 / dt / 1e12:.1f} TFLOP/s (fp32-equivalent)", flush=True)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
