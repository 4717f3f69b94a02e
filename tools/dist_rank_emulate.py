"""ONE rank of a `world`-rank run of the metric's step, measured alone on one GPU (the 8-GPU claim made
checkable without an 8-GPU node).

    python tools/dist_rank_emulate.py [--world 8] [--ranks 0,7] [--config C4] [--gbps 100] [--latency-us 10] [--out profiles/r04_rank_emulation.json]

For every rank asked for, the process runs that rank's EXACT kernel sequence of the distributed step on the loop-back communicator
(rphip_comm_loopback_create): the collectives are stream-ordered and asynchronous like the RCCL ones, move nothing between ranks
- what the rank would receive is filled with its own data or noise of the right size - and hold their stream for
latency + bytes / rate, rate = ONE link of the fully connected node (default 100 GB/s: two thirds of an xGMI link's 153 GB/s)
and bytes = what the busiest link carries: a broadcast its whole message (the root feeds 7 links at once), an all-gather one rank's
part (7 parts over 7 links), an exchange the largest piece to or from one peer.  Measured per rank:

  ItemKNN half   K0 of the shard (rphip_csr_from_device_shard: the degree pass over the rank's item rows) + the all-reduce of the
                 degree vector (modelled) + the fit of the rank's rows (cost-balanced ranges, as bench.py cuts them) + the
                 all-gather of the neighbour lists (modelled; the rank's own lists stand in for the others') + finalisation of the
                 full model + K8 for the rank's users
  EASE half      rphip_ease_fit_dist (Gram of the rank's panels, the symmetric block sweeps with look-ahead: update stream, chain,
                 pivot stream, communication stream) + the predict route bench.py takes at this size: rphip_ease_model_allgather_i8 (the
                 int8 rows of the panels on every rank) + rphip_predict_topk_owner for the rank's USERS (first pass, candidates
                 all-gathered, exact sums of the candidates in the rank's columns for the users of all ranks, one exchange, ranking and
                 proof; --predict-route gather: the full model gathered + the single-GPU predict).  That predict runs on the REAL B of the single-GPU fit of the same process (a fresh model object per
                 run: the fp16 copy of the first pass is rebuilt as it would be): the emulated fit works on noise, and on noise the
                 fp16 first pass of K9 cannot prove its lists (every user would take the exact second pass)

The results of such a run are meaningless (the data of the other ranks never existed); the launches, their sizes and their
overlap are rank r's.  `speedup_emulated` = the single-GPU step of the same build (measured in the same process) divided by the
slowest emulated rank.  What the emulation cannot show: contention on the xGMI links beyond the modelled rate, RCCL's own kernels
taking CUs from the products, skew between the ranks' arrival at a collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default="0,7", help="comma separated ranks to emulate ('' = none: the single-GPU step alone)")
    ap.add_argument("--config", default="C4")
    ap.add_argument("--gbps", type=float, default=100.0)
    ap.add_argument("--latency-us", type=float, default=10.0)
    ap.add_argument("--K", type=int, default=200)
    ap.add_argument("--Kp", type=int, default=100)
    ap.add_argument("--l2", type=float, default=200.0)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--panel-tiles", type=int, default=0, help="panel width of the distributed solve in 128-column tiles (0: the library's choice)")
    ap.add_argument("--skip-single", action="store_true", help="do not measure the single-GPU step (no speed-up is computed then)")
    ap.add_argument("--predict-route", choices=("owner", "gather"), default="owner",
                    help="EASE predict of a rank: owner = int8 rows gathered, exact sums by the columns' owners (rphip_predict_topk_owner); "
                         "gather = the full model gathered in two phases (round 5's first route)")
    ap.add_argument("--owner-on-emulated", action="store_true",
                    help="with --skip-single (sizes whose single-GPU B does not fit beside a rank, e.g. C5's 160 GB): run the owner route on the "
                         "EMULATED fit's own panels and int8 rows - noise, on which no list can be proven, so EVERY user takes the exact pass: "
                         "the route's worst case in memory and time (an upper bound of the rank's predict)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch

    from recpack_amd import _lib, dist as rdist
    from recpack_amd.synth import CONFIG_SEED_BASE, CONFIGS, powerlaw_csr

    dev = torch.device("cuda", 0)
    U, I, dens = CONFIGS[args.config]
    ip, ix = powerlaw_csr(U, I, dens, CONFIG_SEED_BASE + int(args.config[1:]), device=dev)
    nnz = int(ix.numel())
    ctx = _lib.Context(0)
    W = args.world
    Kp = min(args.Kp, I)

    peak = {"used": 0}

    def sync():
        ctx.sync()
        torch.cuda.synchronize()
        free_b, total_b = torch.cuda.mem_get_info(dev)  # (device-wide: the library's own allocations count)
        peak["used"] = max(peak["used"], total_b - free_b)
        peak["total"] = total_b

    # the static plan of bench.py: cost-balanced item-row ranges, user ranges
    dX0 = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    P_total = dX0.num_products()
    deg_full = torch.as_tensor(dX0.item_degrees(), device=dev).to(torch.int32)
    row_parts = rdist.partition_by_cost(rdist.item_row_cost(dX0.item_degrees(), nnz, P_total), W)
    user_parts = rdist.partition_users(ip.cpu().numpy(), W)
    dX0.close()

    out = {"config": args.config, "world": W, "link_model": {"gbytes_per_s": args.gbps, "latency_us": args.latency_us},
           "matrix": {"users": U, "items": I, "nnz": nnz}, "ranks": {}}

    # ---- the single-GPU step of the same build
    if not args.skip_single:
        oi = torch.empty((U, Kp), dtype=torch.int32, device=dev)
        ov = torch.empty((U, Kp), dtype=torch.float32, device=dev)
        on = torch.empty((U,), dtype=torch.int32, device=dev)
        best = None
        for _ in range(args.repeats + 1):
            sync()
            t0 = time.perf_counter()
            dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
            m = _lib.itemknn_fit(ctx, dX, args.K)
            _lib.predict_topk_dev(ctx, m, dX, Kp, oi, ov, on)
            sync()
            t1 = time.perf_counter()
            em = _lib.ease_fit(ctx, dX, args.l2, 0.0)
            sync()
            t2 = time.perf_counter()
            _lib.predict_topk_dev(ctx, em, dX, Kp, oi, ov, on)
            sync()
            t3 = time.perf_counter()
            cur = {"itemknn_s": t1 - t0, "ease_fit_s": t2 - t1, "ease_predict_s": t3 - t2, "step_s": t3 - t0}
            if best is None or cur["step_s"] < best["step_s"]:
                best = cur
            del m
            if _ < args.repeats:
                del em
            dX.close()
        out["single_gpu"] = {k: round(v, 5) for k, v in best.items()}
        out["single_gpu"]["hbm_used_peak_gb"] = round(peak["used"] / 1e9, 2)
        del oi, ov, on
        ptr, n_it, ld = em.dense_device_ptr()
        Breal = torch.as_tensor(rdist._DevPtr(ptr, (n_it, ld), "<f4", em), device=dev)  # (the single-GPU model stays alive: its B is the emulated ranks' "gathered" B)
        # the model the emulated ranks predict with: the REAL matrix, its int8 rows built here, outside every timed region (a rank's
        # own int8 rows come out of the gather's first phase, which the emulation does run and time)
        pm = _lib.Model.from_dense_cols(ctx, I, 0, I, Breal[:, :I])
        wi = torch.empty((1, Kp), dtype=torch.int32, device=dev)
        wv = torch.empty((1, Kp), dtype=torch.float32, device=dev)
        wn = torch.empty((1,), dtype=torch.int32, device=dev)
        _lib.predict_topk_dev(ctx, pm, _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I)), Kp, wi, wv, wn, user_begin=0, user_end=1)
        sync()

    # ---- one rank at a time
    for r in [int(x) for x in args.ranks.split(",") if x.strip() != ""]:
        comm = _lib.Comm.loopback(ctx, W, r, args.gbps, args.latency_us)
        rb, re_ = row_parts[r]
        ub, ue = user_parts[r]
        oi = torch.empty((ue - ub, Kp), dtype=torch.int32, device=dev)
        ov = torch.empty((ue - ub, Kp), dtype=torch.float32, device=dev)
        on = torch.empty((ue - ub,), dtype=torch.int32, device=dev)
        best = None
        for _ in range(args.repeats + 1):
            sync()
            t0 = time.perf_counter()
            # ItemKNN half.  The degree all-reduce: the loop-back leaves the partial degrees; the full vector (what the sum over the
            # ranks would give) is copied in so that the fit selects among the true keys - 4 * I bytes, inside the timed region
            dX = _lib.DeviceCSR.from_device_shard(ctx, ip, ix, (U, I), (rb, re_))
            comm.all_reduce_sum_i32(dX.degrees_device_ptr(), I)
            dst = torch.as_tensor(rdist._DevPtr(dX.degrees_device_ptr(), (I,), "<i4", dX), device=dev)
            ctx.sync()
            dst.copy_(deg_full)
            torch.cuda.synchronize()
            dX.finish_degrees()
            sync()
            t_k0 = time.perf_counter()
            print(f"  rank {r}: K0 shard done", flush=True)
            local = _lib.itemknn_fit(ctx, dX, args.K, rb, re_)
            sync()
            t_fit = time.perf_counter()
            full = _lib.itemknn_gather(ctx, comm, local, [p[0] for p in row_parts] + [row_parts[-1][1]])
            sync()
            t_gather = time.perf_counter()
            print(f"  rank {r}: fit + gather done", flush=True)
            _lib.predict_topk_dev(ctx, full, dX, Kp, oi, ov, on, user_begin=ub, user_end=ue)
            sync()
            t1 = time.perf_counter()
            print(f"  rank {r}: ItemKNN half done", flush=True)
            # EASE half
            dXe = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
            ctx.reset_timings()
            em_r = _lib.ease_fit_dist(ctx, comm, dXe, args.l2, 0.0, args.panel_tiles)
            sync()
            t2 = time.perf_counter()
            print(f"  rank {r}: EASE fit done", flush=True)
            tm = ctx.timings()
            st = _lib.ease_fit_dist_stats(ctx)
            # the gather: its int8 phase is complete (in stream order) when the call returns, the f32 rows follow on a stream of their
            # own (its contents are the emulated fit's noise: only its time counts).  The predict of the rank's users runs on the REAL
            # matrix and ADOPTS the pending gather: its first pass runs beside the f32 strips, its rescoring waits for them - what the
            # rank's own predict on the gathered model does.
            standin_s = 0.0
            if args.predict_route == "owner" and args.skip_single and args.owner_on_emulated:
                efull = rdist.ease_gather_rows_i8(ctx, em_r, I, comm)
                sync()
                t_g = time.perf_counter()
                bounds = [p[0] for p in user_parts] + [user_parts[-1][1]]
                _lib.predict_topk_owner(ctx, comm, em_r, efull, dXe, bounds, Kp, oi, ov, on)
                sync()
                so = _lib.predict_owner_stats(ctx)
                standin_s = 0.0
            elif args.predict_route == "owner" and not args.skip_single:
                # the owner route: the int8 rows of the emulated fit's panels are gathered (timed: quantisation, all-gathers, unpacking),
                # then the route runs on the REAL matrix - its int8 rows (pm) for the first pass, the rank's share of the real columns
                # (a contiguous eighth: the same number of columns as its block-cyclic panels) for the exact sums; the library's stand-in
                # for the other ranks' parts is timed inside and subtracted
                efull = rdist.ease_gather_rows_i8(ctx, em_r, I, comm)
                t_g = time.perf_counter()
                cb, ce = (I * r) // W, (I * (r + 1)) // W
                local_real = _lib.Model.from_dense_cols(ctx, I, cb, ce - cb, Breal[:, cb:ce])
                bounds = [p[0] for p in user_parts] + [user_parts[-1][1]]
                _lib.predict_topk_owner(ctx, comm, local_real, pm, dXe, bounds, Kp, oi, ov, on)
                sync()
                so = _lib.predict_owner_stats(ctx)
                standin_s = so["loopback_standin_us"] / 1e6
                del local_real
            else:
                efull = rdist.ease_gather_model(ctx, em_r, I, comm)
                t_g = time.perf_counter()
                if not args.skip_single:
                    pm.adopt_gather(efull)
                    _lib.predict_topk_dev(ctx, pm, dXe, Kp, oi, ov, on, user_begin=ub, user_end=ue)
                so = None
            sync()
            del efull, em_r
            t3 = time.perf_counter() - standin_s
            cur = {"step_s": t3 - t0, "itemknn_s": t1 - t0, "itemknn_k0_shard_s": t_k0 - t0, "itemknn_fit_rows_s": t_fit - t_k0,
                   "itemknn_gather_finalize_s": t_gather - t_fit, "itemknn_predict_users_s": t1 - t_gather,
                   "ease_fit_s": t2 - t1, "ease_predict_s": t3 - t2, "ease_gather_call_s": t_g - t2,
                   "ease_update_stream_kernel_s": tm["sweep_update"]["ms"] / 1e3, "ease_chain_kernel_s": tm["sweep_chain"]["ms"] / 1e3,
                   "ease_gram_s": tm["gram_dense"]["ms"] / 1e3, "ease_mirror_s": tm["mirror"]["ms"] / 1e3,
                   "ease_bytes_received": st["bytes_received"], "ease_bytes_sent": st["bytes_sent"], "ease_pivots": st["pivots"],
                   "ease_panel": st["panel"], "rows": [int(rb), int(re_)], "users": [int(ub), int(ue)], "ease_predict_route": args.predict_route}
            if so is not None:
                cur["ease_predict_unproven_users"] = so["unproven_here"]
                cur["ease_predict_standin_subtracted_s"] = standin_s
            cur["hbm_used_peak_gb"] = round(peak["used"] / 1e9, 2)
            cur["hbm_total_gb"] = round(peak.get("total", 0) / 1e9, 2)
            if best is None or cur["step_s"] < best["step_s"]:
                best = cur
            del local, full
            dX.close()
            dXe.close()
        out["ranks"][str(r)] = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in best.items()}
        print(f"rank {r} of {W}:", out["ranks"][str(r)], flush=True)
        comm.close()
        del oi, ov, on

    if out["ranks"] and "single_gpu" in out:
        slow = {k: max(v[k] for v in out["ranks"].values()) for k in ("step_s", "itemknn_s", "ease_fit_s", "ease_predict_s")}
        s = out["single_gpu"]
        out["speedup_emulated"] = {"step": round(s["step_s"] / slow["step_s"], 2), "itemknn": round(s["itemknn_s"] / slow["itemknn_s"], 2),
                                   "ease_fit": round(s["ease_fit_s"] / slow["ease_fit_s"], 2),
                                   "ease_predict": round(s["ease_predict_s"] / slow["ease_predict_s"], 2),
                                   "note": f"single-GPU time / the slowest of the emulated ranks {sorted(out['ranks'])}"}
    print(json.dumps(out), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
