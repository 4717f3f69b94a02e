import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recpack_amd import _lib
from recpack_amd import dist as rdist
from recpack_amd.synth import powerlaw_csr
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
for U, I, dens in ((200000, 50000, 0.001), (1000000, 100000, 0.0005)):
    ip, ix = powerlaw_csr(U, I, dens, seed=20260003, device=dev)
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    for it in range(3):
        ctx.reset_timings(); ctx.sync(); t0 = time.perf_counter()
        m = _lib.ease_fit(ctx, dX, 200.0, 0.0); ctx.sync()
        dt = time.perf_counter() - t0
        tm = ctx.timings()
        if it == 2:
            B = rdist.model_dense_tensor(m, dev)
            v = B.contiguous().view(torch.int32)
            h = int(v.to(torch.int64).sum().item()) ^ int((v.to(torch.int64) * 2654435761 % 1000003).sum().item())
            print(f"I={I} fit {dt*1e3:.1f} ms cholesky {tm['cholesky']['ms']:.1f} inverse {tm['inverse']['ms']:.1f} hash {h}", flush=True)
            del B, v
        m.close()
    dX.close(); ctx.trim()
