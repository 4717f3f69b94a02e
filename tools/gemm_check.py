"""C = alpha A B^T + beta C of the library's GEMM (gemm_mode 0: fp32 MFMA, 2: two fp16 planes) against float64 numpy."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from recpack_amd import _lib

ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
for mode, sup in ((0, 0), (2, 0)):
    ctx.set_option("gemm_mode", mode)
    ctx.set_option("gemm_supertile", sup)
    for (M, N, K) in ((256, 256, 512), (384, 256, 4096), (1024, 1024, 16384)):
        g = torch.Generator(device="cpu").manual_seed(5)
        A = torch.randn((M, K), generator=g, dtype=torch.float32) * torch.exp(2.0 * torch.randn((M, 1), generator=g))
        B = torch.randn((N, K), generator=g, dtype=torch.float32)
        C0 = torch.randn((M, N), generator=g, dtype=torch.float32)
        ref = 0.5 * (A.double() @ B.double().T) - 1.0 * C0.double()
        Ad, Bd, Cd = A.to(dev), B.to(dev), C0.to(dev).clone()
        _lib.gemm_nt(ctx, Ad, Bd, Cd, alpha=0.5, beta=-1.0)
        ctx.sync()
        err = ((Cd.cpu().double() - ref).abs() / (0.5 * (A.double().abs() @ B.double().abs().T) + C0.double().abs())).max().item()
        print(f"mode {mode} sup {sup} M={M} N={N} K={K}: max error relative to sum |a||b| = {err:.2e}", flush=True)
    if mode == 2:  # the pre-split path (many tiles), in several K chunks
        ctx.set_option("gemm_workspace_bytes", 64 << 20)
        g = torch.Generator(device="cpu").manual_seed(6)
        M, N, K = 3072, 2944, 16384
        A = torch.randn((M, K), generator=g, dtype=torch.float32) * torch.exp(2.0 * torch.randn((M, 1), generator=g))
        B = torch.randn((N, K), generator=g, dtype=torch.float32)
        C0 = torch.randn((M, N), generator=g, dtype=torch.float32)
        ref = 0.5 * (A.double() @ B.double().T) - 1.0 * C0.double()
        Ad, Bd, Cd = A.to(dev), B.to(dev), C0.to(dev).clone()
        _lib.gemm_nt(ctx, Ad, Bd, Cd, alpha=0.5, beta=-1.0)
        ctx.sync()
        err = ((Cd.cpu().double() - ref).abs() / (0.5 * (A.double().abs() @ B.double().abs().T) + C0.double().abs())).max().item()
        print(f"mode {mode} pre-split {M}x{N}x{K} in chunks: max error relative to sum |a||b| = {err:.2e}", flush=True)
        ctx.set_option("gemm_workspace_bytes", 4 << 30)
        for ps in (0, 1):
            ctx.set_option("gemm_presplit", ps)
            n = 8192
            A = torch.randn((n, n), dtype=torch.float32, device=dev)
            C = torch.empty((n, n), dtype=torch.float32, device=dev)
            for it in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _lib.gemm_nt(ctx, A, A, C)
                ctx.sync()
                dt = time.perf_counter() - t0
            print(f"mode {mode} presplit {ps} {n}^3: {dt * 1e3:.1f} ms = {2 * n ** 3 / dt / 1e12:.1f} TFLOP/s (fp32-equivalent)", flush=True)
        continue
    n = 8192
    A = torch.randn((n, n), dtype=torch.float32, device=dev)
    C = torch.empty((n, n), dtype=torch.float32, device=dev)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.gemm_nt(ctx, A, A, C)
        ctx.sync()
        dt = time.perf_counter() - t0
    print(f"mode {mode} sup {sup} {n}^3: {dt * 1e3:.1f} ms = {2 * n ** 3 / dt / 1e12:.1f} TFLOP/s (fp32-equivalent)", flush=True)
