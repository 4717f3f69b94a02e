"""Time the multi-GPU EASE routes of recpack_amd.dist (one rank per GPU; plain `python` = one rank):
    python tools/ease_sharded_bench.py <users> <items> <density> [l2 [route [panel_tiles]]]
route: dist (the library's distributed solve, default) | sweeps | replicated | single (rphip_ease_fit, for comparison)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd import dist as rdist  # noqa: E402
from recpack_amd.synth import powerlaw_csr  # noqa: E402

U, I, dens = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
l2 = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
backend = os.environ.get("RECPACK_DIST_BACKEND", "nccl")
local = int(os.environ.get("LOCAL_RANK", "0"))
if backend != "nccl":
    local %= max(1, torch.cuda.device_count())
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx = _lib.Context(local, stream.cuda_stream)
ip, ix = powerlaw_csr(U, I, dens, seed=20260003, device=dev)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
route = sys.argv[5] if len(sys.argv) > 5 else "dist"
pt = int(sys.argv[6]) if len(sys.argv) > 6 else 0
comm = rdist.make_comm(ctx, dev) if route == "dist" else None
for it in range(2):
    ctx.reset_timings()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if route == "dist":
        m, lay = rdist.ease_fit_sharded(ctx, dX, l2, 0.0, dev, panel_tiles=pt, comm=comm)
    elif route == "sweeps":
        m, cols = rdist.ease_fit_sweeps(ctx, dX, l2, 0.0, dev, panel_tiles=pt or 16)
    elif route == "single":
        m = _lib.ease_fit(ctx, dX, l2, 0.0)
    else:
        m = rdist.ease_fit_replicated(ctx, dX, l2, 0.0, dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        n_pad = -(-I // 128) * 128
        fl = 2.0 if route == "sweeps" else 1.0
        extra = ""
        if route == "dist":
            st = _lib.ease_fit_dist_stats(ctx)
            tm = ctx.timings()
            extra = (f"; nb {st['panel']}, {st['pivots']} pivots, slab {st['slab_bytes'] / 2**30:.2f} GiB + {st['other_peak_bytes'] / 2**30:.2f} GiB, "
                     f"received {st['bytes_received'] / 2**30:.2f} GiB, rank flop {st['flop'] / 1e12:.1f} T; kernel ms: update {tm['sweep_update']['ms']:.0f} "
                     f"chain {tm['sweep_chain']['ms']:.0f} mirror {tm['mirror']['ms']:.1f} scale {tm['ease_scale']['ms']:.1f} gram {tm['gram_dense']['ms']:.1f}")
        print(f"world {world}: I={I} {route} fit {dt*1e3:.1f} ms = {fl * n_pad**3 / dt / 1e12:.1f} TF/s by the {fl:.0f} n^3 model{extra}", flush=True)
    if route in ("dist", "sweeps") and it == 1:
        t0 = time.perf_counter()
        rdist.predict_topk_sharded(ctx, m, dX, 100, dev, comm=comm)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"world {world}: sharded predict + top-100 of {U} users {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    del m
