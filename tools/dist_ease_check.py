"""Functional check of the multi-GPU EASE routes of recpack_amd.dist against the single-GPU solve: ease_fit_sharded (the library's
distributed solve on block-cyclic panels, through the communicator: RCCL with the nccl backend, torch.distributed callbacks
otherwise) + predict_topk_sharded.

    python tools/dist_ease_check.py [users items density [panel_tiles [dist_tiles]]]         (one rank)
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
dist_tiles = int(sys.argv[5]) if len(sys.argv) > 5 else 1     # panel width (128-column tiles) of the distributed solve's first pass
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
    # ---- the library's distributed solve: this rank's columns of B (block-cyclic panels), memory, the sharded predict + merge
    comm = rdist.make_comm(ctx, device, transport=os.environ.get("RECPACK_DIST_TRANSPORT", "auto"))
    if l2 == 50.0:  # the transport's four operations on a known pattern, every word checked (what bench.py --gpus N runs before its warm-up)
        st_ = comm.selftest(words=50_000, watchdog_s=60.0)
        print(f"rank {rank}/{world} transport self-test: {'ok' if st_['ok'] else 'FAILED'} {st_['ops']}", flush=True)
        ok = ok and st_["ok"]
    for pt in (dist_tiles, 0):
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        ms, lay = rdist.ease_fit_sharded(ctx, dX, l2, alpha, device, panel_tiles=pt, comm=comm)
        st = _lib.ease_fit_dist_stats(ctx)
        torch_peak = torch.cuda.max_memory_allocated() - base
        err_s, nc = 0.0, 0
        if ms is not None:
            cols = ms.local_columns()
            nc = len(cols)
            ptr, _, ld = ms.dense_device_ptr()
            slab = torch.as_tensor(rdist._DevPtr(ptr, (I, ld), "<f4", ms), device=device)[:, :nc].cpu().numpy()
            err_s = float(np.abs(slab - B_ref[:, cols]).max() / np.abs(B_ref).max())
        users, pi, pv, pn = rdist.predict_topk_sharded(ctx, ms, dX, 20, device, exclude_history=True, comm=comm)
        users = users.cpu().numpy()
        sel = users < nu
        same = pi.cpu().numpy()[sel] == b[0][users[sel]]
        agree_s = float(same.mean()) if sel.any() else 1.0
        # scores where both picked the same item at the same position (a near-tie at the K-th place shifts the rest of a row)
        dv = float((np.abs(pv.cpu().numpy()[sel] - b[1][users[sel]]) * same).max() / max(1e-30, np.abs(b[1]).max())) if sel.any() else 0.0
        # memory: the panels are n_pad^2 * 4 / world bytes (+ one panel of imbalance); everything else the fit holds (two pivot columns,
        # T, staging, the plane workspaces of the products) is O(n nb), not O(n^2): at most 10 n_pad nb floats + the fixed minimum sizes
        # of the workspaces - 15 % of the matrix at 100 000 items with nb = 2048 - and nothing comes from torch
        mem_ok = (torch_peak == 0 and st["slab_bytes"] <= (st["n_pad"] // world + st["panel"]) * st["n_pad"] * 4
                  and st["other_peak_bytes"] <= 10 * st["n_pad"] * st["panel"] * 4 + (1024 << 20))  # (workspaces: powers of two from 64 MiB)
        print(f"rank {rank}/{world} dist panel_tiles={pt} (nb {st['panel']}, {st['pivots']} pivots) l2={l2} alpha={alpha}: {nc} columns, "
              f"max|dB|/max|B| = {err_s:.2e}, sharded top-20 agreement {agree_s:.4f}, score diff {dv:.1e}, slab {st['slab_bytes'] / 2**20:.1f} MiB + "
              f"{st['other_peak_bytes'] / 2**20:.1f} MiB, torch allocations {torch_peak}, received {st['bytes_received'] / 2**20:.1f} MiB "
              f"(n_pad^2*4 = {st['n_pad'] ** 2 * 4 / 2**20:.1f} MiB), rccl {comm.is_rccl}", flush=True)
        ok = ok and err_s <= 1e-4 and agree_s > 0.99 and dv <= 1e-4 and mem_ok
        # the panels gathered into one full model on every rank: all of B, and the single-GPU predict on this rank's share of the users
        full = rdist.ease_gather_model(ctx, ms, I, comm)
        Bg = full.dense_download()
        err_g = float(np.abs(Bg - B_ref).max() / np.abs(B_ref).max())
        ub_, ue_ = (nu * rank) // world, (nu * (rank + 1)) // world
        g = _lib.predict_topk(ctx, full, dX, 20, exclude_history=True, user_begin=ub_, user_end=ue_)
        agree_g = float((g[0] == b[0][ub_:ue_]).mean()) if ue_ > ub_ else 1.0
        print(f"rank {rank}/{world} gathered model: max|dB|/max|B| = {err_g:.2e}, empty rows equal: {full.empty_rows() == ref.empty_rows()}, "
              f"users [{ub_},{ue_}) agreement {agree_g:.4f}", flush=True)
        ok = ok and err_g <= 1e-4 and full.empty_rows() == ref.empty_rows() and agree_g > 0.99
        # the owner route (round 5): first pass on the gathered int8 rows, the candidates' exact sums from every rank's own panels, one
        # exchange back - the predict of the gathered model on this rank's users, bit for bit (columns, f32 scores, counts), with and
        # without history removal; refused alike by every rank where the first pass does not apply (fewer than 4 Kc items)
        if world > 1:
            uparts = rdist.partition_users(X.indptr, world)
            ub2, ue2 = uparts[rank]
            try:
                rows8 = rdist.ease_gather_rows_i8(ctx, ms, I, comm)
                for excl in (True, False):
                    oi, ov, on = rdist.predict_topk_owner(ctx, ms, rows8, dX, 20, uparts, device, comm, exclude_history=excl)
                    g2 = _lib.predict_topk(ctx, full, dX, 20, exclude_history=excl, user_begin=ub2, user_end=ue2)
                    same_o = (np.array_equal(oi.cpu().numpy(), g2[0]) and np.array_equal(ov.cpu().numpy().view(np.uint32), g2[1].view(np.uint32))
                              and np.array_equal(on.cpu().numpy(), g2[2]))
                    so = _lib.predict_owner_stats(ctx)
                    print(f"rank {rank}/{world} owner route (exclude_history={excl}): users [{ub2},{ue2}) identical to the gathered model's predict: {same_o}"
                          f" (users without a proof: {so['unproven_here']} here, {so['unproven_all_ranks']} on all ranks)", flush=True)
                    ok = ok and same_o
                del rows8
            except _lib.RecpackHipError as e:
                refused = e.code == _lib.RPHIP_ELIMIT and I < 4 * 128
                print(f"rank {rank}/{world} owner route refused ({'expected: too few items for the first pass' if refused else e})", flush=True)
                ok = ok and refused
        del ms, full
    comm.close()
if dist.is_initialized():
    t = torch.tensor([1 if ok else 0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    ok = bool(t.item())
    dist.destroy_process_group()
sys.exit(0 if ok else 1)
