"""K9 on a FITTED EASE model (the proof of the two-pass predict needs real scores): time of the fused predict + top-Kp for all users by
route - `predict_user_tasks` 1 (one task per user, round 5) / 0 (one per (user, tile) + merge, round 4) - and bit equality of the results.
    python tools/k9_route_bench.py [C4] [Kp]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import CONFIGS, config_csr  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
Kp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
res = {}
for excl in (False, True):
    for route in (1, 0, 1, 0):
        ctx.set_option("predict_user_tasks", route)
        oi = torch.empty((U, Kp), dtype=torch.int32, device=dev)
        ov = torch.empty((U, Kp), dtype=torch.float32, device=dev)
        on = torch.empty((U,), dtype=torch.int32, device=dev)
        ctx.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.predict_topk_dev(ctx, m, dX, Kp, oi, ov, on, exclude_history=excl)
        ctx.sync()
        dt = time.perf_counter() - t0
        print(f"{cfg} exclude_history={excl} predict_user_tasks={route}: {dt * 1e3:.1f} ms", flush=True)
        key = (excl, route)
        if key in res:
            assert torch.equal(res[key][0], oi) and torch.equal(res[key][1], ov.view(torch.int32)) and torch.equal(res[key][2], on), "run to run"
        res[key] = (oi, ov.view(torch.int32), on)
    a, b = res[(excl, 1)], res[(excl, 0)]
    print("   routes identical:", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]), torch.equal(a[2], b[2]), flush=True)
