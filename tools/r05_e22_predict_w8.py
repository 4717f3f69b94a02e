"""Where the emulated rank's EASE predict (0.164 s at W = 8, C4) goes: the predict of the rank's users alone on a complete model,
the gather alone (int8 phase / f32 strips), both together as the step runs them."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recpack_amd import _lib, dist as rdist  # noqa: E402
from recpack_amd.synth import CONFIG_SEED_BASE, CONFIGS, powerlaw_csr  # noqa: E402

dev = torch.device("cuda", 0)
U, I, dens = CONFIGS["C4"]
ip, ix = powerlaw_csr(U, I, dens, CONFIG_SEED_BASE + 4, device=dev)
ctx = _lib.Context(0)
W, r, Kp = 8, 0, 100
user_parts = rdist.partition_users(ip.cpu().numpy(), W)
ub, ue = user_parts[r]


def sync():
    ctx.sync()
    torch.cuda.synchronize()


dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
em = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ptr, n_it, ld = em.dense_device_ptr()
Breal = torch.as_tensor(rdist._DevPtr(ptr, (n_it, ld), "<f4", em), device=dev)
pm = _lib.Model.from_dense_cols(ctx, I, 0, I, Breal[:, :I])
oi = torch.empty((ue - ub, Kp), dtype=torch.int32, device=dev)
ov = torch.empty((ue - ub, Kp), dtype=torch.float32, device=dev)
on = torch.empty((ue - ub,), dtype=torch.int32, device=dev)
_lib.predict_topk_dev(ctx, pm, dX, Kp, oi, ov, on, user_begin=ub, user_end=ue)
sync()
for it in range(3):
    ctx.reset_timings()
    t0 = time.perf_counter()
    _lib.predict_topk_dev(ctx, pm, dX, Kp, oi, ov, on, user_begin=ub, user_end=ue)
    sync()
    t1 = time.perf_counter()
    tm = ctx.timings()
    print(f"predict alone, {ue - ub} users: {1e3 * (t1 - t0):.1f} ms; kernels: " +
          ", ".join(f"{k} {v['ms']:.1f}" for k, v in tm.items() if v["ms"] > 0.05), flush=True)
comm = _lib.Comm.loopback(ctx, W, r, 100.0, 10.0)
em_r = _lib.ease_fit_dist(ctx, comm, dX, 200.0, 0.0, 0)
sync()
for it in range(3):
    t0 = time.perf_counter()
    efull = rdist.ease_gather_model(ctx, em_r, I, comm)
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"gather alone: call returns after {1e3 * (t1 - t0):.1f} ms (int8 phase queued), everything done after {1e3 * (t2 - t0):.1f} ms", flush=True)
    # wait for the side stream too
    pm.adopt_gather(efull)
    t3 = time.perf_counter()
    _lib.predict_topk_dev(ctx, pm, dX, Kp, oi, ov, on, user_begin=ub, user_end=0 + ub + 1)
    sync()
    t4 = time.perf_counter()
    print(f"   one-user predict behind the adopted gather (waits for the f32 strips): {1e3 * (t4 - t3):.1f} ms", flush=True)
    del efull
for it in range(3):
    sync()
    t0 = time.perf_counter()
    efull = rdist.ease_gather_model(ctx, em_r, I, comm)
    pm.adopt_gather(efull)
    _lib.predict_topk_dev(ctx, pm, dX, Kp, oi, ov, on, user_begin=ub, user_end=ue)
    sync()
    t1 = time.perf_counter()
    print(f"gather + predict as the step runs them: {1e3 * (t1 - t0):.1f} ms", flush=True)
    del efull
