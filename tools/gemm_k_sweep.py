"""The plane product at M = N = 16384 for a range of K (with beta = 1: C is read and written, as in the solve's updates): whole call
(amax + plane cut + product) and the rate, to see what a short K costs.  python tools/gemm_k_sweep.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from recpack_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = _lib.Context(0)
if os.environ.get("RECPACK_XCD_BLOCK"):
    ctx.set_option("gemm_xcd_block", int(os.environ["RECPACK_XCD_BLOCK"]))
dev = torch.device("cuda", 0)
A = torch.randn((n, 16384), dtype=torch.float32, device=dev)
C = torch.zeros((n, n), dtype=torch.float32, device=dev)
for K in (1024, 2048, 4096, 8192, 16384):
    best = 1e9
    for it in range(3):
        ctx.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.gemm_nt(ctx, A[:, :K], A[:, :K], C, alpha=-1.0, beta=1.0)
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"M = N = {n}, K = {K}: {best * 1e3:.2f} ms = {2.0 * n * n * K / best / 1e12:.1f} TFLOP/s fp32-equivalent", flush=True)
