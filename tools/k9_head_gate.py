"""Gate for a head / tail split of K9's first pass: how long does the two-pass predict take when the H most popular items are
taken out of the histories (the tail that would still be streamed), and what does a dense fp16 product of the head part cost
(torch.matmul = hipBLASLt, as a yardstick only).
    python tools/k9_head_gate.py [config [H ...]]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from recpack_amd import _lib
from recpack_amd.synth import CONFIGS, config_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
Hs = [int(a) for a in sys.argv[2:]] or [1024, 2048, 4096]
Kp = 100
dev = torch.device("cuda", 0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
oi = torch.empty((U, Kp), dtype=torch.int32, device=dev)
ov = torch.empty((U, Kp), dtype=torch.float32, device=dev)
on = torch.empty((U,), dtype=torch.int32, device=dev)


def timed(X, label):
    for it in range(2):
        torch.cuda.synchronize()
        ctx.reset_timings()
        t0 = time.perf_counter()
        _lib.predict_topk_dev(ctx, m, X, Kp, oi, ov, on, exclude_history=True)
        ctx.sync()
        dt = time.perf_counter() - t0
    print(f"{cfg} {label}: {dt * 1e3:.1f} ms", flush=True)
    return dt


t_full = timed(dX, f"all {int(ix.numel())} interactions")
cnt = torch.bincount(ix.long(), minlength=I)
order = torch.argsort(cnt, descending=True)
rows = torch.repeat_interleave(torch.arange(U, device=dev), (ip[1:] - ip[:-1]).long())
for H in Hs:
    is_head = torch.zeros(I, dtype=torch.bool, device=dev)
    is_head[order[:H]] = True
    keep = ~is_head[ix.long()]
    share = 1.0 - float(keep.float().mean())
    ix_t = ix[keep].contiguous()
    deg = torch.bincount(rows[keep], minlength=U)
    ip_t = torch.zeros(U + 1, dtype=ip.dtype, device=dev)
    ip_t[1:] = torch.cumsum(deg, 0)
    dT = _lib.DeviceCSR.from_device(ctx, ip_t, ix_t, (U, I))
    t_tail = timed(dT, f"H={H}: head share of the interactions {share:.3f}; tail only")
    # yardstick: dense fp16 product of 131072 users' head part
    Uc = 131072
    A = torch.zeros((Uc, H), dtype=torch.float16, device=dev)
    Bh = torch.randn((H, I), dtype=torch.float16, device=dev)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        C = A @ Bh
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"   torch fp16 matmul {Uc} x {H} x {I}: {dt * 1e3:.1f} ms = {2.0 * Uc * H * I / dt / 1e12:.0f} TFLOP/s -> {dt * U / Uc * 1e3:.0f} ms for all users; "
          f"tail {t_tail * 1e3:.0f} + head = {(t_tail + dt * U / Uc) * 1e3:.0f} ms against {t_full * 1e3:.0f} ms", flush=True)
    del A, Bh, C, dT
