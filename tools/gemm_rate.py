"""n^3 plane product (rphip_gemm_nt_f32 on random fp32 operands), a few repetitions: python tools/gemm_rate.py [n [wide [reps]]]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from recpack_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wide = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = _lib.Context(0)
ctx.set_option("gemm_wide", wide)
dev = torch.device("cuda", 0)
A = torch.randn((n, n), dtype=torch.float32, device=dev)
C = torch.empty((n, n), dtype=torch.float32, device=dev)
for it in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.gemm_nt(ctx, A, A, C)
    ctx.sync()
    dt = time.perf_counter() - t0
print(f"wide={wide} {n}^3: {dt * 1e3:.2f} ms = {2 * n ** 3 / dt / 1e12:.1f} TFLOP/s (fp32-equivalent)", flush=True)
