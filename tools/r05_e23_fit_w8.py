"""The emulated rank's distributed EASE fit alone (W = 8, C4, rank 0), twice; prints the wall time of the second run."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import CONFIG_SEED_BASE, CONFIGS, powerlaw_csr  # noqa: E402

dev = torch.device("cuda", 0)
U, I, dens = CONFIGS["C4"]
ip, ix = powerlaw_csr(U, I, dens, CONFIG_SEED_BASE + 4, device=dev)
ctx = _lib.Context(0)
W = int(os.environ.get("E23_WORLD", "8"))
r = int(os.environ.get("E23_RANK", "0"))
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
comm = _lib.Comm.loopback(ctx, W, r, 100.0, 10.0)
for it in range(2):
    ctx.sync()
    torch.cuda.synchronize()
    ctx.reset_timings()
    t0 = time.perf_counter()
    m = _lib.ease_fit_dist(ctx, comm, dX, 200.0, 0.0, 0)
    ctx.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tm = ctx.timings()
    print(f"fit {it}: wall {1e3 * (t1 - t0):.1f} ms; " + ", ".join(f"{k} {v['ms']:.1f}" for k, v in tm.items() if v["ms"] > 0.5), flush=True)
    del m
