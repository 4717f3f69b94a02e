"""Functional check of the multi-GPU EASE routes of recpack_amd.dist (ease_fit_replicated; ease_fit_sharded = column slabs +
predict_topk_sharded) against the single-GPU solve.

    python tools/dist_ease_check.py [users items density [panel_tiles]]         (one rank)
    RECPACK_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_ease_check.py
(with gloo all ranks may share one GPU; with nccl one rank per GPU).  Exit code 0 = B of every rank equals the reference.
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd import dist as rdist  # noqa: E402
from recpack_amd.synth import powerlaw_csr  # noqa: E402

U, I, dens = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (4000, 700, 0.02)
panel_tiles = int(sys.argv[4]) if len(sys.argv) > 4 else 16  # small values exercise the wide-panel steps at test sizes
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
backend = os.environ.get("RECPACK_DIST_BACKEND", "nccl")
local = int(os.environ.get("LOCAL_RANK", "0"))
if backend != "nccl":
    local %= max(1, torch.cuda.device_count())
torch.cuda.set_device(local)
device = torch.device("cuda", local)
if world > 1 or os.environ.get("RECPACK_FORCE_PG"):  # (RECPACK_FORCE_PG=1: a one-rank group, to exercise the RCCL calls)
    if "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1")
    dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
ctx = _lib.Context(local)
X = powerlaw_csr(U, I, dens, seed=123).to_scipy()
dX = _lib.DeviceCSR.from_host(ctx, X.indptr, X.indices, X.shape)
ok = True
n_pad = -(-I // (128 * world)) * (128 * world)
for l2, alpha in ((50.0, 0.0), (5.0, 0.5)):
    ref = _lib.ease_fit(ctx, dX, l2, alpha)
    B_ref = ref.dense_download()
    nu = min(U, 300)
    b = _lib.predict_topk(ctx, ref, dX, 20, exclude_history=True, user_begin=0, user_end=nu)
    # ---- the replicated route: every rank ends with all of B
    m = rdist.ease_fit_replicated(ctx, dX, l2, alpha, device, panel_tiles=panel_tiles)
    B = m.dense_download()
    err = float(np.abs(B - B_ref).max() / np.abs(B_ref).max())
    same_empty = m.empty_rows() == ref.empty_rows()
    a = _lib.predict_topk(ctx, m, dX, 20, exclude_history=True, user_begin=0, user_end=nu)
    agree = float((a[0] == b[0]).mean())
    print(f"rank {rank}/{world} replicated l2={l2} alpha={alpha}: max|dB|/max|B| = {err:.2e}, empty rows equal: {same_empty}, top-20 agreement {agree:.4f}", flush=True)
    ok = ok and err <= 2e-5 and same_empty and agree > 0.99
    del m, B
    ctx.trim()
    torch.cuda.empty_cache()
    # ---- the column-sharded route: this rank's columns of B, peak memory, and the sharded predict + merge
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    # (128-column panels at these small sizes: the two panel buffers are (2 n_pad + nb) nb floats and must stay below 0.1 n_pad^2)
    ms, (c0, nc) = rdist.ease_fit_sharded(ctx, dX, l2, alpha, device, panel_tiles=1 if n_pad < 8192 else 2)
    peak = torch.cuda.max_memory_allocated() - base
    budget = n_pad * n_pad * 4 * (1.0 / world + 0.1) + (1 << 20)
    err_s = 0.0
    if nc > 0:
        slab = ms._keep[:I, :nc].cpu().numpy()
        err_s = float(np.abs(slab - B_ref[:, c0:c0 + nc]).max() / np.abs(B_ref).max())
    users, pi, pv, pn = rdist.predict_topk_sharded(ctx, ms, dX, 20, device, exclude_history=True, user_chunk=257)
    users = users.cpu().numpy()
    sel = users < nu
    same = pi.cpu().numpy()[sel] == b[0][users[sel]]
    agree_s = float(same.mean()) if sel.any() else 1.0
    # scores where both picked the same item at the same position (a near-tie at the K-th place shifts the rest of a row)
    dv = float((np.abs(pv.cpu().numpy()[sel] - b[1][users[sel]]) * same).max() / max(1e-30, np.abs(b[1]).max())) if sel.any() else 0.0
    mem_ok = world == 1 or n_pad < 2560 or peak <= budget  # (below 20 tiles the 128-column granularity alone exceeds the 10 %)
    print(f"rank {rank}/{world} columns [{c0},{c0 + nc}) l2={l2} alpha={alpha}: max|dB|/max|B| = {err_s:.2e}, sharded top-20 agreement {agree_s:.4f}, "
          f"score diff {dv:.1e}, peak {peak / 2**20:.1f} MiB (budget n_pad^2*4*(1/world+0.1) = {budget / 2**20:.1f} MiB)", flush=True)
    ok = ok and err_s <= 1e-4 and agree_s > 0.99 and dv <= 1e-4 and mem_ok
    del ms
if dist.is_initialized():
    t = torch.tensor([1 if ok else 0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    ok = bool(t.item())
    dist.destroy_process_group()
sys.exit(0 if ok else 1)
