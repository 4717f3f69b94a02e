"""Time EASE fit (K4-K7) on a synthetic matrix: python tools/ease_bench.py <users> <items> <density> [l2]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import powerlaw_csr  # noqa: E402

U, I, dens = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
l2 = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0
dev = torch.device("cuda", 0)
ip, ix = powerlaw_csr(U, I, dens, seed=20260003, device=dev)
ctx = _lib.Context(0)
if os.environ.get("RECPACK_GEMM_MODE"):
    ctx.set_option("gemm_mode", int(os.environ["RECPACK_GEMM_MODE"]))
if os.environ.get("RECPACK_GEMM_FLUSH_K"):
    ctx.set_option("gemm_flush_k", int(os.environ["RECPACK_GEMM_FLUSH_K"]))
if os.environ.get("RECPACK_GEMM_WIDE"):
    ctx.set_option("gemm_wide", int(os.environ["RECPACK_GEMM_WIDE"]))
if os.environ.get("RECPACK_GEMM_SUPERTILE"):
    ctx.set_option("gemm_supertile", int(os.environ["RECPACK_GEMM_SUPERTILE"]))
for name, opt in (("RECPACK_XCD_BLOCK", "gemm_xcd_block"), ("RECPACK_CHOL_PANEL", "chol_panel"), ("RECPACK_INV_PANEL", "inv_panel"), ("RECPACK_CHOL_SUB", "chol_sub_panel"), ("RECPACK_N22_EARLY", "inv_n22_early")):
    if os.environ.get(name):
        ctx.set_option(opt, int(os.environ[name]))
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
for it in range(int(os.environ.get('RECPACK_BENCH_ITERS', '3'))):
    ctx.reset_timings()
    t0 = time.perf_counter()
    m = _lib.ease_fit(ctx, dX, l2, 0.0)
    ctx.sync()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    n_pad = (I + 127) // 128 * 128
    chol, inv = tm["cholesky"], tm["inverse"]
    print(f"I={I} nnz={ix.numel()} fit {dt*1e3:.1f} ms | gram_dense {tm['gram_dense']['ms']:.1f} ms | cholesky {chol['ms']:.1f} ms "
          f"({chol['work']/chol['ms']/1e9:.1f} TF/s) | inverse {inv['ms']:.1f} ms ({inv['work']/inv['ms']/1e9:.1f} TF/s) | scale {tm['ease_scale']['ms']:.1f} ms | n^3 model {n_pad**3/((chol['ms']+inv['ms'])*1e9):.1f} TF/s")
    del m
