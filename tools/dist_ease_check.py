"""Functional check of recpack_amd.dist.ease_fit_sharded against the single-GPU solve.

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
for l2, alpha in ((50.0, 0.0), (5.0, 0.5)):
    ref = _lib.ease_fit(ctx, dX, l2, alpha)
    B_ref = ref.dense_download()
    m = rdist.ease_fit_sharded(ctx, dX, l2, alpha, device, panel_tiles=panel_tiles)
    B = m.dense_download()
    err = float(np.abs(B - B_ref).max() / np.abs(B_ref).max())
    same_empty = m.empty_rows() == ref.empty_rows()
    # predict through the sharded model
    a = _lib.predict_topk(ctx, m, dX, 20, exclude_history=True, user_begin=0, user_end=min(U, 300))
    b = _lib.predict_topk(ctx, ref, dX, 20, exclude_history=True, user_begin=0, user_end=min(U, 300))
    agree = float((a[0] == b[0]).mean())
    print(f"rank {rank}/{world} l2={l2} alpha={alpha}: max|dB|/max|B| = {err:.2e}, empty rows equal: {same_empty}, top-20 agreement {agree:.4f}", flush=True)
    ok = ok and err <= 2e-5 and same_empty and agree > 0.99
if dist.is_initialized():
    t = torch.tensor([1 if ok else 0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    ok = bool(t.item())
    dist.destroy_process_group()
sys.exit(0 if ok else 1)
