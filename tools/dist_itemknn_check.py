"""Functional check of the multi-GPU ItemKNN path on the HIP kernels (recpack_amd.dist: csr_for_shard = K0 with the degree pass
sharded + one all-reduce, fit_itemknn_sharded = item-row shards + all-gather, user-sharded predict) against the single-GPU fit.

    RECPACK_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_itemknn_check.py [users items density K]
(with gloo all ranks may share one GPU; with nccl one rank per GPU).  Exit code 0 = every rank's model and predictions are
bit-identical to the single-GPU ones."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd import dist as rdist  # noqa: E402
from recpack_amd.synth import powerlaw_csr  # noqa: E402

U, I, dens = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (20000, 30000, 0.0015)
K = int(sys.argv[4]) if len(sys.argv) > 4 else 50
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
backend = os.environ.get("RECPACK_DIST_BACKEND", "nccl")
local = int(os.environ.get("LOCAL_RANK", "0"))
if backend != "nccl":
    local %= max(1, torch.cuda.device_count())
torch.cuda.set_device(local)
device = torch.device("cuda", local)
dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
ctx = _lib.Context(local)
ip, ix = powerlaw_csr(U, I, dens, seed=321, device=device)
# ---- single-GPU reference on this rank's device
dX0 = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
ref = _lib.itemknn_fit(ctx, dX0, K)
r_idx, r_val, r_nnz = ref.topk_download()
Kp = 20
r_pi, r_pv, r_pn = _lib.predict_topk(ctx, ref, dX0, Kp, exclude_history=True)
deg0 = dX0.item_degrees()
# ---- sharded: plan from the degrees, K0 per shard (+ all-reduce), fit of the shard's rows, all-gather, predict of the shard's users
parts = rdist.partition_by_cost(rdist.item_row_cost(deg0, int(ix.numel()), dX0.num_products()), world)
uparts = rdist.partition_users(ip.cpu().numpy(), world)
ok = True
# (1) the collectives inside the library (the communicator: RCCL with the nccl backend, torch.distributed callbacks otherwise),
# (2) the torch-driven collectives of round 2
for comm in (rdist.make_comm(ctx, device), None):
    dX = rdist.csr_for_shard(ctx, ip, ix, (U, I), parts[rank], device, comm=comm)
    deg_ok = bool((dX.item_degrees() == deg0).all())
    m = rdist.fit_itemknn_sharded(ctx, dX, K, parts, device, comm=comm)
    idx, val, nnz = m.topk_download()
    same_model = bool((idx == r_idx).all() and (nnz == r_nnz).all() and (val == r_val).all())
    ub, ue = uparts[rank]
    pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=True, user_begin=ub, user_end=ue)
    same_pred = bool((pi == r_pi[ub:ue]).all() and (pn == r_pn[ub:ue]).all() and (pv == r_pv[ub:ue]).all())
    print(f"rank {rank}/{world} ({'library communicator' if comm is not None else 'torch.distributed'}): rows {parts[rank]}, users [{ub},{ue}): "
          f"degrees after the all-reduce equal: {deg_ok}, model bit-identical: {same_model}, predictions bit-identical: {same_pred}", flush=True)
    ok = ok and deg_ok and same_model and same_pred
    if comm is not None:
        comm.close()
t = torch.tensor([1 if ok else 0], device=device)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
ok = bool(t.item())
dist.destroy_process_group()
sys.exit(0 if ok else 1)
