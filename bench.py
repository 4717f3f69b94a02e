#!/usr/bin/env python3
"""Headline benchmark: fit + predict wall-seconds (ItemKNN, EASE) on the 1M x 100K synthetic power-law matrix (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C4]

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (a child
`python -m torch.distributed.run`, before this process touches the GPU) and forwards rank 0's JSON line and the
children's exit code; under torchrun (the driver's multi-GPU command) it is one of the ranks.
`value` = seconds per step; a step is ONE pass of the whole metric - ItemKNN fit + predict AND EASE fit + predict - over one
synthetic interaction matrix that is already resident in HBM:
  ItemKNN: K0 (X -> X^T, degrees) + K1-K3 (Gram + cosine + top-K, K=200) [+ all-gather of the neighbour lists when N > 1]
           + K8 (predict with fused top-100 for every user)                                  -> `value_itemknn`
  EASE:    K4 (dense Gram) + K5/K6 (the solve) + K7 + K9 (predict with fused top-100)        -> `value_ease`
Default workload: C4 = 1M users x 100K items, 50M interactions - the shape BASELINE.json's metric is quoted on; it fits one
MI355X.  With N > 1 the same matrix is sharded (ItemKNN: item rows for fit, users for predict; EASE: the library's distributed
solve on block-cyclic column panels, rphip_ease_fit_dist, and the column-sharded predict + merge): strong scaling.

Prints one JSON line (rank 0).  `roofline` is the dominant kernel's: the EASE solve against the dense 16-bit MFMA peak (three fp16
plane products per fp32-equivalent product; `frac_n3_model` states the plain n^3 model beside it); `roofline_hbm` prices the HBM-bound
ItemKNN kernel with the longest launch (K8) with its algorithmic byte model (4*P-model for the SpGEMM instances,
8*K*nnz-model for predict) against 8 TB/s - its `legs` hold all three ItemKNN kernels, `spgemm_all_kernels` the whole K1-K3 group;
`k9_predict` prices K9 by the bytes it streams; `parity` = the parity gates of SURVEY.md §8(d), checked AFTER the timed steps on the models the
last step left (384 random rows of the ItemKNN model bit-identical to the oracle's; the whole-matrix probe of the EASE model, oracle/ease_probe.py:
two random float64 probes through (X^T X + l2 I)(I - B) must give the same 1 / P_jj for every item, bar 1e-4) - a failed gate changes `status`;
`itemknn_c2` / `ease_c3` = BASELINE.json's configs[1] / [2] outside the timed step; `roofline_hbm.spgemm_all_kernels.{traffic,
measured_fabric_GBps}` = the counter-side bytes of the K1-K3 kernels (committed PMC pass) over this run's time;
`cold_first_step_s` is the first step of the fresh context; `cpu_baseline` = the
reference-shaped route (sklearn cosine + per-row argpartition loop, float64 np.linalg.inv) timed on a bounded sample on the host's
cores and scaled, with the oracle's own vectorised route under `port`.  `status` is "ok" or names what failed; a failed or
timed-out run exits non-zero.  N > 1: every way out leaves a status line on rank 0 - a rank that fails writes a note beside the rendezvous
that rank 0's watcher thread turns into the line (phase, rank, reason, the last collective entered); watchdogs cover the communicator set-up, the
transport self-test and the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
LDS_ATOMIC_PEAK = 2.7e12  # measured: tools/lds_atomic_bench.hip, non-returning u32 LDS adds, all 256 CUs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C4", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--K", type=int, default=200)
    ap.add_argument("--Kp", type=int, default=100)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0, help="0 disables the CPU leg")
    ap.add_argument("--ease-config", default="C3",
                    help="extra EASE legs outside the timed step (N=1 only), comma separated configs -> `ease_<cfg>`; 'none' disables them")
    ap.add_argument("--itemknn-config", default="C2",
                    help="extra ItemKNN legs outside the timed step (N=1 only), comma separated configs -> `itemknn_<cfg>` (C2 = BASELINE.json "
                         "configs[1]); 'none' disables them")
    ap.add_argument("--no-parity-check", dest="parity_check", action="store_false", default=True,
                    help="skip the parity verdict (oracle rows of the ItemKNN model, whole-matrix probe of the EASE model) printed under `parity`")
    ap.add_argument("--no-ease-in-step", dest="ease_in_step", action="store_false", default=True,
                    help="time the ItemKNN half alone (the step then leaves EASE out and `status` says so)")
    ap.add_argument("--l2", type=float, default=200.0)
    ap.add_argument("--tars-config", default="C4",
                    help="config of the time-aware leg (N=1 only): TARSItemKNN-style valued cosine fit + decayed predict on the "
                         "config's matrix with seeded timestamps; 'none' skips it")
    ap.add_argument("--ease-transport", default="auto", choices=["auto", "rccl", "torch"],
                    help="N > 1: collectives of the distributed EASE solve - the library's own RCCL communicator (auto with the nccl "
                         "backend) or callbacks through torch.distributed")
    ap.add_argument("--ease-predict-route", choices=("auto", "owner", "gather"), default="auto",
                    help="N > 1: auto = the owner route (int8 rows gathered, exact sums by the columns' owners), then the gathered model, then "
                         "columns sharded; owner = that route or an error; gather = round 5's first route (the full model gathered)")
    ap.add_argument("--selftest", action="store_true", help="N = 1: run the transport self-test on a one-rank RCCL communicator too (N > 1: always)")
    ap.add_argument("--inject-failure-rank", type=int, default=-1,
                    help="test hook (tests/test_gpu_ease.py): this rank raises in its first warm-up step - the line rank 0 then prints is what is tested")
    ap.add_argument("--inject-failure-hard", action="store_true", help="test hook: the injected failure is a silent os._exit(9), as a crash would be")
    ap.add_argument("--timeout", type=float, default=1500.0,
                    help="seconds the timed region (warm-up + steps) may take; past it every rank prints what it has with status "
                         "'timeout' and exits with code 3")
    return ap.parse_args()


def cpu_baseline(indptr, indices, shape, K, Kp, S_csr, budget_s, P_total):
    """Time the oracle (port of the reference's scipy path) on a bounded sample; scale to the full workload."""
    import scipy.sparse as sp

    from oracle import oracle_np as onp

    U, I = shape
    nnz = indices.shape[0]
    X = sp.csr_matrix((np.ones(nnz, dtype=np.int32), indices, indptr), shape=shape)
    rng = np.random.default_rng(1)
    d = np.diff(indptr).astype(np.int64)
    w = np.bincount(indices, weights=np.repeat(d, d), minlength=I)  # w_i = sum_{u in i} d_u
    # --- fit: binarise + transpose at full size, Gram rows + top-K on a row sample
    t0 = time.perf_counter()
    Xb = onp.binarize(X)
    _ = Xb.T.tocsr()
    t_prep = time.perf_counter() - t0
    # ~3e7 scalar products per second for scipy's csr_matmat on one core
    target_p = max(1.0, 0.5 * budget_s * 2.5e7)
    order = rng.permutation(I)
    cw = np.cumsum(w[order])
    n_rows = int(min(I, max(8, np.searchsorted(cw, target_p) + 1)))
    rows = np.sort(order[:n_rows])
    t0 = time.perf_counter()
    onp.itemknn_fit(X, K, rows=rows)
    t_rows = max(1e-9, time.perf_counter() - t0 - t_prep)
    p_sample = float(w[rows].sum())
    fit_full = t_prep + t_rows * (P_total / max(1.0, p_sample))
    # --- predict: X[users] @ S + per-user top-Kp on a user sample
    n_users = int(min(U, max(64, 0.5 * budget_s * 4.0e4 * 200.0 / max(1, K) / max(1.0, nnz / U))))
    users = np.sort(rng.choice(U, size=n_users, replace=False))
    Xs = X[users]
    t0 = time.perf_counter()
    scores = onp.predict_full(Xs, S_csr)
    onp.topk_rows(scores, Kp)
    t_pred = time.perf_counter() - t0
    pred_full = t_pred * (nnz / max(1, Xs.nnz))
    out = {
        "value": round(fit_full + pred_full, 3),
        "unit": "s",
        "cores": 1,
        "kind": "port",
        "sample": (f"oracle/oracle_np.py (scipy csr_matmat + vectorised top-K, 1 thread): binarise+transpose at full size "
                   f"{t_prep:.2f}s; Gram+top-K of {n_rows} random item rows ({p_sample / P_total:.3%} of P) {t_rows:.2f}s; "
                   f"predict+top-{Kp} of {n_users} random users ({Xs.nnz / nnz:.3%} of nnz) {t_pred:.2f}s; "
                   f"scaled by P and nnz to fit {fit_full:.1f}s + predict {pred_full:.1f}s"),
        "host_cpus": os.cpu_count(),
    }
    # --- the same sample through the reference-SHAPED restatement: sklearn's cosine_similarity + setdiag(0) + the per-row
    # argpartition loop (nearest_neighbour.py:80, util.py:63-73), then X @ S + the same loop on the predictions
    t0 = time.perf_counter()
    onp.itemknn_fit_reference_shaped(X, K, rows=rows)
    t_rows_r = time.perf_counter() - t0
    t0 = time.perf_counter()
    onp.predict_topk_reference_shaped(Xs, S_csr, Kp)
    t_pred_r = time.perf_counter() - t0
    fit_r = t_rows_r * (P_total / max(1.0, p_sample))
    pred_r = t_pred_r * (nnz / max(1, Xs.nnz))
    out["reference_shaped"] = {
        "value": round(fit_r + pred_r, 3), "unit": "s", "cores": 1,
        "sample": (f"sklearn cosine_similarity(X.T[rows], X.T, dense_output=False) + zeroed diagonal + per-row argpartition loop on the "
                   f"same {n_rows} item rows {t_rows_r:.2f}s; X @ S + per-row argpartition loop on the same {n_users} users "
                   f"{t_pred_r:.2f}s; scaled by P and nnz to fit {fit_r:.1f}s + predict {pred_r:.1f}s"),
    }
    return out


def cpu_baseline_ease(indptr, indices, shape, l2, Kp, sizes=(8192, 16384), budget_s=150.0):
    """The reference's EASE._fit (ease.py:78-86: dense Gram, float64 LAPACK inverse, column scaling) through the oracle, on the
    matrix restricted to a random sample of I' items for I' in `sizes`, on all host cores (BLAS threads reported), and
    X @ B for a sample of users; extrapolated to the full item count by I^3 (inverse) - labelled as such."""
    import scipy.sparse as sp

    from oracle import oracle_np as onp

    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:  # noqa: BLE001
        blas_threads = None
    U, I = shape
    nnz = indices.shape[0]
    X = sp.csr_matrix((np.ones(nnz, dtype=np.float64), indices, indptr), shape=shape)
    rng = np.random.default_rng(2)
    points = []
    B = None
    cols = None
    for n in sizes:
        if n > I:
            continue
        if points and points[-1]["fit_s"] * (n / points[-1]["items"]) ** 3 > budget_s:
            break
        cols = np.sort(rng.choice(I, size=n, replace=False))
        Xs = X[:, cols].tocsr()
        t0 = time.perf_counter()
        B = onp.ease_fit(Xs, l2)
        points.append({"items": int(n), "nnz": int(Xs.nnz), "fit_s": round(time.perf_counter() - t0, 3)})
    if not points:
        return None
    last = points[-1]
    fit_full = last["fit_s"] * (I / last["items"]) ** 3
    # predict: X @ B (base.py:248) on a user sample of the last restricted matrix, scaled by nnz * I
    n_users = 2000
    users = np.sort(rng.choice(U, size=n_users, replace=False))
    Xp = X[users][:, cols].tocsr()
    t0 = time.perf_counter()
    sc = Xp @ B
    onp.topk_per_row_loop(sp.csr_matrix(sc), Kp)  # the metrics' per-row loop (util.py:63-73)
    t_pred = time.perf_counter() - t0
    pred_full = t_pred * (nnz / max(1, Xp.nnz)) * (I / last["items"])
    return {"value": round(fit_full + pred_full, 1), "unit": "s", "cores": blas_threads, "kind": "port",
            "measured_points": points,
            "sample": (f"oracle ease_fit (scipy Gram .toarray(), np.linalg.inv float64, {blas_threads} BLAS threads) on random item subsets "
                       f"{[p['items'] for p in points]}: {[p['fit_s'] for p in points]} s; EXTRAPOLATED by (I/I')^3 from the largest point to "
                       f"{I} items: fit {fit_full:.0f} s; X @ B + top-{Kp} of {n_users} users on the {last['items']}-item subset {t_pred:.2f}s, "
                       f"scaled by nnz x I (single-threaded scipy): predict {pred_full:.0f} s"),
            "host_cpus": os.cpu_count()}


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 matrix peak (v_mfma_f32_32x32x16_bf16 / _f16); never the sparse figure


def ease_leg(args, ctx, device, Kp, config, repeats):
    """EASE fit + predict on one config (C4 = the metric's 1M x 100K matrix, 40 GB of B; C3 = BASELINE.json configs[2],
    200K x 50K; lambda = 200); the solve is priced with the n^3 flop model of SURVEY.md §8(d) against the dense fp32
    MFMA peak."""
    import torch

    from recpack_amd import _lib
    from recpack_amd.synth import CONFIG_SEED_BASE, CONFIGS, powerlaw_csr

    U, I, dens = CONFIGS[config]
    ip, ix = powerlaw_csr(U, I, dens, CONFIG_SEED_BASE + int(config[1:]), device=device)
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    out_idx = torch.empty((U, Kp), dtype=torch.int32, device=device)
    out_val = torch.empty((U, Kp), dtype=torch.float32, device=device)
    out_nnz = torch.empty((U,), dtype=torch.int32, device=device)
    best = None
    walls = []
    for it in range(repeats + 1):  # run 0 is the warm-up (first-touch of the 4 n^2-byte allocation on a fresh lease): not reported as the leg
        ctx.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model = _lib.ease_fit(ctx, dX, args.l2, 0.0)
        torch.cuda.synchronize()
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        _lib.predict_topk_dev(ctx, model, dX, Kp, out_idx, out_val, out_nnz)
        torch.cuda.synchronize()
        t_pred = time.perf_counter() - t0
        tm = ctx.timings()
        walls.append(round(t_fit + t_pred, 4))
        if it > 0 and (best is None or t_fit + t_pred < best[0] + best[1]):
            best = (t_fit, t_pred, tm)
        del model
    t_fit, t_pred, tm = best
    n_pad = (I + 127) // 128 * 128
    solve_ms = tm["cholesky"]["ms"] + tm["inverse"]["ms"]
    tf = n_pad ** 3 / (solve_ms * 1e-3) / 1e12
    return {
        "workload": f"{config}: EASE l2={args.l2}, {U} users x {I} items, nnz {int(ix.numel())}, predict + top-{Kp} for all users",
        "fit_s": round(t_fit, 4),
        "predict_s": round(t_pred, 4),
        "fit_plus_predict_s": round(t_fit + t_pred, 4),
        "wall_s_all_runs": walls,  # [warm-up, timed ...]: the warm-up pays the first allocation of B
        "kernel_s_in_fit": round(sum(v["ms"] for k, v in tm.items() if not k.startswith("predict")) / 1e3, 4),  # HIP-event time inside the fit wall time
        "phases_ms": {k: round(v["ms"], 3) for k, v in tm.items() if v["launches"]},
        # the solve's products run on the 16-bit matrix pipe with fp32 operands cut into two fp16 planes: THREE fp16 MFMA products per
        # fp32-equivalent product, so the matrix pipe executes 3 n^3 fp16 flop for the n^3 of the model - priced against the dense
        # fp16 / bf16 peak; `fp32_equivalent` is the same time against the n^3 model (what an fp32 MFMA kernel would have to sustain)
        "roofline": {
            "kernel": "k_gemm_hp16w + k_split_planes (large products) / k_gemm_nt_h2 (skinny ones) (K5 Cholesky + K6 inverse: v_mfma_f32_16x16x32_f16 on "
                      "fp32 operands cut into two fp16 planes, fp32 accumulate, compensated master sums)",
            "bound": "mfma",
            "achieved": round(3.0 * tf, 2),
            "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(3.0 * tf / MFMA_BF16_PEAK_TFLOPS, 4),
            "flops_model": "3 fp16 plane products x n^3 (Cholesky n^3/3 + inverse 2n^3/3), n padded to 128",
            "fp32_equivalent": {"achieved": round(tf, 2), "peak_fp32_mfma": MFMA_F32_PEAK_TFLOPS, "frac_of_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                                "flops_model": "n^3"},
            "solve_ms": round(solve_ms, 2),
        },
        "dtype": "f32 storage; products as 2 x fp16 planes on the fp16 MFMA (22 bits + rounding to nearest = fp32's 2^-24), f32 accumulate + compensated sums, float64 diagonal blocks",
    }


def tars_leg(args, ctx, device, Kp, config):
    """The time-aware family's device work on one config: every interaction gets a seeded timestamp
    (hours over 90 days), the host computes the exponential-decay weights and unit-length columns as TARSItemKNN does
    (time_aware_item_knn/base.py:181-195, sklearn's normalize), the GPU runs the valued cosine fit (rphip_itemknn_fit_values,
    K neighbours) and the decayed predict + top-Kp for all users (the wave kernel on a valued matrix)."""
    import torch

    from recpack_amd import _lib
    from recpack_amd.synth import CONFIG_SEED_BASE, CONFIGS, powerlaw_csr

    U, I, dens = CONFIGS[config]
    ip, ix = powerlaw_csr(U, I, dens, CONFIG_SEED_BASE + int(config[1:]), device=device)
    indptr, indices = ip.cpu().numpy(), ix.cpu().numpy()
    rng = np.random.default_rng(CONFIG_SEED_BASE + 100 + int(config[1:]))
    t0 = time.perf_counter()
    ts = rng.integers(1, 90 * 24, size=indices.shape[0]).astype(np.float64)
    w = np.exp(-0.002 * (ts.max() + 1 - ts))
    n2 = np.bincount(indices, weights=w * w, minlength=I)
    wn = w / np.sqrt(np.where(n2 > 0, n2, 1.0))[indices]
    host_s = time.perf_counter() - t0
    dX = _lib.DeviceCSR.from_host(ctx, indptr, indices, (U, I))
    out_idx = torch.empty((U, Kp), dtype=torch.int32, device=device)
    out_val = torch.empty((U, Kp), dtype=torch.float32, device=device)
    out_nnz = torch.empty((U,), dtype=torch.int32, device=device)
    best = None
    for it in range(2):
        dX.set_values(wn)
        ctx.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model = _lib.itemknn_fit_values(ctx, dX, args.K)
        torch.cuda.synchronize()
        t_fit = time.perf_counter() - t0
        dX.set_values(w)
        t0 = time.perf_counter()
        _lib.predict_topk_dev(ctx, model, dX, Kp, out_idx, out_val, out_nnz)
        torch.cuda.synchronize()
        t_pred = time.perf_counter() - t0
        tm = ctx.timings()
        if best is None or t_fit + t_pred < best[0] + best[1]:
            best = (t_fit, t_pred, tm)
        del model
    dX.close()
    t_fit, t_pred, tm = best
    return {
        "workload": f"{config}: valued cosine ItemKNN K={args.K} on exponential-decay weights (TARSItemKNN), {U} users x {I} items, "
                    f"nnz {int(indices.shape[0])}, decayed predict + top-{Kp} for all users",
        "fit_s": round(t_fit, 4),
        "predict_s": round(t_pred, 4),
        "fit_plus_predict_s": round(t_fit + t_pred, 4),
        "host_weights_s": round(host_s, 3),
        "phases_ms": {k: round(v["ms"], 3) for k, v in tm.items() if v["launches"]},
        "dtype": "f64 Gram (LDS f64 atomics), f64 weight x f32 similarity in exact 64-bit fixed-point sums",
    }


def itemknn_leg(args, ctx, device, Kp, config, repeats):
    """ItemKNN cosine fit + predict on one config outside the timed step (C2 = BASELINE.json configs[1]: 100K users x 20K items):
    K0 + K1-K3 + K8 with the CSR resident in HBM, best of `repeats` after one warm-up."""
    import torch

    from recpack_amd import _lib
    from recpack_amd.synth import CONFIG_SEED_BASE, CONFIGS, powerlaw_csr

    U, I, dens = CONFIGS[config]
    ip, ix = powerlaw_csr(U, I, dens, CONFIG_SEED_BASE + int(config[1:]), device=device)
    out_idx = torch.empty((U, Kp), dtype=torch.int32, device=device)
    out_val = torch.empty((U, Kp), dtype=torch.float32, device=device)
    out_nnz = torch.empty((U,), dtype=torch.int32, device=device)
    best = None
    walls = []
    P_cfg = None
    for it in range(repeats + 1):
        ctx.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
        model = _lib.itemknn_fit(ctx, dX, args.K)
        torch.cuda.synchronize()
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        _lib.predict_topk_dev(ctx, model, dX, Kp, out_idx, out_val, out_nnz)
        torch.cuda.synchronize()
        t_pred = time.perf_counter() - t0
        tm = ctx.timings()
        P_cfg = dX.num_products()
        walls.append(round(t_fit + t_pred, 5))
        if it > 0 and (best is None or t_fit + t_pred < best[0] + best[1]):
            best = (t_fit, t_pred, tm)
        del model, dX
    t_fit, t_pred, tm = best
    nnz = int(ix.numel())
    sp_ms = sum(tm[k]["ms"] for k in ("gram_topk", "gram_topk_wave", "gram_finalize", "topk_merge"))
    sp_bytes = 4.0 * P_cfg + 8.0 * nnz + 4.0 * (U + I + 2) + 8.0 * I * args.K
    k8_bytes = 8.0 * args.K * nnz + 4.0 * (nnz + U) + 8.0 * U * Kp
    k8_ms = tm["predict"]["ms"]
    return {
        "workload": f"{config}: ItemKNN cosine K={args.K}, {U} users x {I} items, nnz {nnz} (P={P_cfg}), predict + top-{Kp} for all users",
        "fit_s": round(t_fit, 5), "predict_s": round(t_pred, 5), "fit_plus_predict_s": round(t_fit + t_pred, 5),
        "wall_s_all_runs": walls,
        "phases_ms": {k: round(v["ms"], 4) for k, v in tm.items() if v["launches"]},
        "spgemm_all_kernels": {"ms": round(sp_ms, 4), "algorithmic_bytes": sp_bytes, "GBps": round(sp_bytes / (sp_ms * 1e-3) / 1e9, 1) if sp_ms > 0 else None,
                               "frac": round(sp_bytes / (sp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sp_ms > 0 else None},
        "k8_predict": {"ms": round(k8_ms, 4), "algorithmic_bytes": k8_bytes, "GBps": round(k8_bytes / (k8_ms * 1e-3) / 1e9, 1) if k8_ms > 0 else None,
                       "frac": round(k8_bytes / (k8_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k8_ms > 0 else None},
        "dtype": "int32 counts, f64 similarity, i32 fixed-point score sums",
    }


def parity_verdict(args, ctx, device, indptr_t, indices_t, shape, knn_model, ease_model, n_rows=384):
    """SURVEY.md 8(d) "parity gates per run", AFTER the timed steps, on the models the last step left (N = 1): the oracle as the
    checker, never inside the timed region.
      ItemKNN: `n_rows` random item rows of the fitted model against oracle/oracle_np.py::itemknn_fit on the same matrix - neighbour
               sets and order bit-identical, values to 1e-12 (the full-size test checks 10 000 rows);
      EASE:    the whole-matrix probe (oracle/ease_probe.py): A (I - B) is diagonal, so two random float64 probes must give the
               same 1 / P_jj for every item - every entry of B takes part; bar 1e-4 relative per item (north_star)."""
    import scipy.sparse as sp
    import torch

    from oracle import oracle_np as onp
    from oracle.ease_probe import ease_probe
    from recpack_amd import dist as rdist

    U, I = shape
    out = {"checked_after_timed_steps": True}
    t0 = time.perf_counter()
    idx, val, nnz_rows = knn_model.topk_download()
    ip, ix = indptr_t.cpu().numpy(), indices_t.cpu().numpy()
    X = sp.csr_matrix((np.ones(ix.shape[0], dtype=np.int32), ix, ip), shape=shape)
    rows = np.sort(np.random.default_rng(11).choice(I, size=min(I, n_rows), replace=False))
    ref = onp.itemknn_fit(X, args.K, rows=rows)
    key = np.where(ref["idx"] < 0, np.iinfo(np.int32).max, ref["idx"])
    perm = np.argsort(key, axis=1, kind="stable")  # (the oracle's rows are in rank order, the model's in column order)
    ridx = np.take_along_axis(ref["idx"], perm, axis=1)
    rval = np.take_along_axis(ref["val"], perm, axis=1)
    same = (idx[rows] == ridx).all(axis=1) & (nnz_rows[rows] == ref["nnz"])
    live = ridx >= 0
    vrel = float(np.max(np.abs(val[rows][live] - rval[live]) / rval[live])) if live.any() and same.all() else None
    out.update({"itemknn_rows_checked": int(len(rows)), "itemknn_rows_identical": int(same.sum()), "itemknn_value_max_rel": vrel,
                "itemknn_ok": bool(same.all() and vrel is not None and vrel <= 1e-12), "itemknn_check_s": round(time.perf_counter() - t0, 2)})
    del X, ref
    if ease_model is not None:
        t0 = time.perf_counter()
        B = rdist.model_dense_tensor(ease_model, device)
        pr = ease_probe(indptr_t, indices_t, shape, B, args.l2, 0.0)
        torch.cuda.synchronize()
        del B
        out.update({"ease_probe_rel": pr["probe_rel"], "ease_probe_rel_median": pr["probe_rel_median"], "ease_items_checked": pr["items_checked"],
                    "ease_bounds_violation": pr["bounds_violation"], "ease_diag_max": pr["diag_max"],
                    "ease_ok": bool(pr["probe_rel"] <= 1e-4 and pr["diag_max"] == 0.0 and pr["bounds_violation"] <= 1e-6),
                    "ease_check_s": round(time.perf_counter() - t0, 2),
                    "ease_probe_what": "max over items of |r_1 - r_0| / |r_0|, r_p = (X^T X + l2 I)(z_p - B z_p) / z_p = 1 / P_jj for two random float64 probes"})
    return out


def roofline_ease(tm, I, world, steps, dist_stats, nnz, U, Kp, k9_stats=None):
    """The EASE solve against the MFMA peak.  One rank: K5 (Cholesky) + K6 (inverse) of rphip_ease_fit, n^3 flop by SURVEY.md
    section 8(d) (n padded to 128).  N > 1: the symmetric block sweeps of rphip_ease_fit_dist, the rank's own products as counted
    by the library (n^3 / world + the T products).  The products run on the 16-bit matrix pipe with each fp32 operand cut into two
    fp16 planes: THREE fp16 MFMA products per fp32-equivalent product - `achieved` counts those against the dense fp16 peak,
    `frac_n3_model` is the plain flop model against the same peak, `fp32_equivalent` the same time against the fp32 MFMA peak."""
    if world == 1:
        n_pad = (I + 127) // 128 * 128
        solve_ms = (tm["cholesky"]["ms"] + tm["inverse"]["ms"]) / steps
        flop = float(n_pad) ** 3
        what = "k_gemm_hp16w + k_split_planes (large products) / k_gemm_nt_h2 (skinny ones): K5 Cholesky + K6 inverse"
    else:
        solve_ms = tm["sweep_update"]["ms"] / steps  # the update stream's products (the chain runs beside them on other streams)
        flop = float(dist_stats["flop"])
        what = (f"k_gemm_hp16w + k_split_planes: the rank's symmetric rank-{dist_stats['panel']} updates of rphip_ease_fit_dist "
                f"({dist_stats['pivots']} pivots; kernel time of the update stream, the pivot chain overlaps it)")
    tf = flop / (solve_ms * 1e-3) / 1e12 if solve_ms > 0 else 0.0
    k9 = tm["predict_dense"]
    out = {
        "kernel": what + " (v_mfma_f32_16x16x32_f16 on fp32 operands cut into two fp16 planes, fp32 accumulate, compensated master sums); "
                  f"`frac` = {3.0 * tf / MFMA_BF16_PEAK_TFLOPS:.3f} of the matrix pipe by the THREE executed plane products per fp32 product, "
                  f"{tf / MFMA_BF16_PEAK_TFLOPS:.3f} by SURVEY 8(d)'s n^3 model (`frac_n3_model`)",
        "bound": "mfma", "achieved": round(3.0 * tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(3.0 * tf / MFMA_BF16_PEAK_TFLOPS, 4),
        "flops_model": "3 fp16 plane products x the fp32-equivalent flop (n^3 per solve: Cholesky n^3/3 + inverse 2n^3/3, n padded to 128)",
        "frac_n3_model": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
        "fp32_equivalent": {"achieved": round(tf, 2), "peak_fp32_mfma": MFMA_F32_PEAK_TFLOPS, "frac_of_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4)},
        "solve_ms_per_step": round(solve_ms, 2), "flop_per_solve": flop,
    }
    if k9["launches"]:
        # K9 beside it.  Since round 4 the first pass streams the INT8 copy of B's rows (1 byte per column per interaction) for a
        # candidate list of Kc = 192 (Kp = 100) and the second pass gathers the candidates' f32 entries for the exact rescoring
        # (4 bytes per candidate per interaction, a 64-byte line each at worst); the users whose list proves nothing (1 %) read the
        # int8 rows once more for a list of 256 (round 6), and who is still unproven (0.3 %) takes the one-pass kernel on the f32
        # rows (4*I bytes per interaction).  Both are priced at their MEASURED share of the users (rphip_predict_stats of the last
        # timed step; their histories are longer than the average, so this understates their bytes).  `bytes_streamed` / time
        # against the HBM peak is the roofline of what the kernels read; at half the bytes of the fp16 pass the first pass is no
        # longer bound by them alone.  The 4*I*nnz no-reuse model of SURVEY.md 8(d) prices f32 rows that the
        # kernel no longer reads - kept as `model_bytes_f32_rows` for reference only.
        ms = k9["ms"] / steps
        kc = ((Kp + 92 + 63) // 64) * 64
        two_pass = bool(k9_stats) and k9_stats.get("users", -1) > 0
        fail_share = round(k9_stats["exact_pass_users"] / k9_stats["users"], 5) if two_pass else 0.0
        second_share = round(k9_stats["second_list_users"] / k9_stats["users"], 5) if two_pass else 0.0
        streamed = (1.0 + second_share) * I * nnz / world + 4.0 * kc * nnz / world + fail_share * 4.0 * I * nnz / world + 8.0 * U * Kp
        out["k9_predict"] = {"ms_per_step": round(ms, 2), "bytes_streamed": streamed,
                             "bytes_streamed_what": f"1*I*nnz (int8 rows of B, first pass) + 4*Kc*nnz (candidates' f32 entries, Kc = {kc}) + "
                                                    f"{second_share}*1*I*nnz (users without a proof: a second list of 256) + "
                                                    f"{fail_share}*4*I*nnz (still unproven: the one-pass kernel) + 8*U*Kp (output); "
                                                    "the two shares are the measured shares of the USERS (not priced when unknown: N > 1)",
                             "users": k9_stats if two_pass else None,
                             "GBps": round(streamed / (ms * 1e-3) / 1e9, 1),
                             "frac_of_hbm_peak": round(streamed / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "model_bytes_f32_rows": 4.0 * I * nnz / world + 8.0 * U * Kp,
                             "TFLOPs": round(2.0 * I * nnz / world / (ms * 1e-3) / 1e12, 2)}
    return out


def pmc_traffic(config, kernel, world):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/pmc_traffic.sh: one rocprofv3 --pmc run per
    counter on this same command).  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B read
    requests as 64 B (MI355X_MICROARCH.md, HBM section), so it is doubled.  None when no pass exists for the config."""
    import glob

    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r*_pmc_traffic_{config.lower()}.json")))
    if world != 1 or not found:
        return None, "no PMC pass committed for this configuration"
    path = found[-1]  # the latest round's pass (a kernel that was renamed since is simply not found: traffic = null)
    prof = json.load(open(path))
    # stale = the sources of the kernels differ from those the pass was taken on (hashes recorded by tools/pmc_traffic.sh)
    import hashlib

    src_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "recpack_amd", "csrc")
    stale = False
    for f, h in prof.get("_sources_sha256", {"rp_predict.hip": None}).items():
        fp = os.path.join(src_dir, f)
        if h is None or not os.path.exists(fp) or hashlib.sha256(open(fp, "rb").read()).hexdigest() != h:
            stale = True
    want = kernel.replace(" ", "")
    for name, c in prof.items():
        if name.startswith("_"):
            continue
        if want in name.replace(" ", "") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = c["FETCH_SIZE"]["avg_per_dispatch"], c["WRITE_SIZE"]["avg_per_dispatch"]
            return (2.0 * f + w) * 1024.0, (f"profiles/{os.path.basename(path)}: (2 x FETCH_SIZE {f:.0f} KiB + WRITE_SIZE {w:.0f} KiB) "
                                            "per launch; the x2 is the guide's gfx950 correction for wide reads and is an upper "
                                            "bound for this kernel's 16-B gathers (uncalibrated width)" +
                                            ("; NOTE: the kernels' sources have changed since the pass - re-run tools/pmc_traffic.sh" if stale else ""))
    return None, "kernel not in the PMC pass"


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process tree (this process has not
    touched the GPU - no HIP call, no torch.cuda query - and never execs), stream rank 0's JSON line through, return
    the children's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # under an external launcher (the driver's `python -m torch.distributed.run ... bench.py`): this host driver supports dmabuf
        # IPC only, RCCL's peer mappings fail with hipIpcGetMemHandle otherwise.  Set before torch (and with it HIP) is imported: the
        # process has not touched the GPU yet, nothing is re-executed.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import threading

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    # (rank notes of an earlier run that used the same rendezvous port must be gone BEFORE the process group comes up - its set-up is
    # the barrier behind which rank 0 starts to watch for them)
    note_base = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"recpack_bench_{os.environ.get('MASTER_PORT', '0')}")
    if world > 1:
        import glob

        for fn in ((glob.glob(note_base + ".rank*") + [note_base + ".ack"]) if rank == 0 else [f"{note_base}.rank{rank}"]):
            try:
                os.remove(fn)
            except OSError:
                pass
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # RECPACK_DIST_BACKEND=gloo lets N ranks share the GPUs that are there (functional test of the sharded path on a
    # one-GPU box); the default is nccl = RCCL over xGMI, one rank per GPU
    backend = os.environ.get("RECPACK_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from recpack_amd import _lib, dist as rdist
    from recpack_amd.synth import CONFIG_SEED_BASE, CONFIGS, powerlaw_csr

    U, I, dens = CONFIGS[args.config]
    seed = CONFIG_SEED_BASE + int(args.config[1:])
    t_gen = time.perf_counter()
    indptr_t, indices_t = powerlaw_csr(U, I, dens, seed, device=device)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    nnz = int(indices_t.numel())

    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    ctx = _lib.Context(local_rank, stream.cuda_stream)
    info = ctx.device_info()

    # static plan (data dependent only through the degrees): item-row ranges and user ranges per rank
    dX0 = _lib.DeviceCSR.from_device(ctx, indptr_t, indices_t, (U, I))
    P_total = dX0.num_products()
    deg = dX0.item_degrees()
    row_parts = rdist.partition_by_cost(rdist.item_row_cost(deg, nnz, P_total), world)
    user_parts = rdist.partition_users(indptr_t.cpu().numpy(), world)
    dX0.close()
    ub, ue = user_parts[rank]
    Kp = min(args.Kp, I)
    out_idx = torch.empty((ue - ub, Kp), dtype=torch.int32, device=device)
    out_val = torch.empty((ue - ub, Kp), dtype=torch.float32, device=device)
    out_nnz = torch.empty((ue - ub,), dtype=torch.int32, device=device)
    ease_on = bool(args.ease_in_step)
    e_idx = e_val = e_nnz = None
    if ease_on and world == 1:
        e_idx = torch.empty((U, Kp), dtype=torch.int32, device=device)
        e_val = torch.empty((U, Kp), dtype=torch.float32, device=device)
        e_nnz = torch.empty((U,), dtype=torch.int32, device=device)
    comm = None
    comm_note = None
    transport = None
    METRIC = "fit+predict wall-sec (ItemKNN, EASE) on 1M×100K synth; achieved HBM GB/s"
    state = {"phase": "communicator set-up", "timing": False}
    # ---- first-contact hardening (N > 1): EVERY way out of this process leaves a status line on rank 0.
    # A rank != 0 that fails writes a one-line note beside the rendezvous (a file named after MASTER_PORT) before it leaves; rank 0
    # polls for such notes from a thread and, when one appears while it is itself stuck in a collective the dead rank will never
    # join, prints the line - status, phase, the rank and its reason - and exits non-zero.  A watchdog covers the set-up and the
    # self-test too (the timed region has its own).
    def read_notes():
        import glob

        others = []
        for fn in sorted(glob.glob(note_base + ".rank*")):
            try:
                msg = open(fn).read().strip()
            except OSError:
                continue
            if msg:
                others.append(f"rank {fn.rsplit('rank', 1)[1]}: {msg}")
        return others

    printed = {"done": False}
    die_lock = threading.Lock()  # (rank 0: the watcher, the SIGTERM thread, a watchdog and the main thread may all arrive; the first one prints)

    def die(code, status, extra=None):
        if os.environ.get("RECPACK_BENCH_DEBUG"):
            sys.stderr.write(f"[debug] rank {rank} die({code}): {status[:120]}\n")
        if rank == 0:
            die_lock.acquire()  # (never released: whoever comes second waits here until the first has printed and left)
            if printed["done"]:
                os._exit(code)
            printed["done"] = True
            if world > 1 and "failed on rank" not in status:
                # rank 0 failed itself - often BECAUSE another rank left (a transport that notices the dead peer): that rank's own words
                # belong in the line (a moment for its note to land)
                for _ in range(10):
                    others = read_notes()
                    if others:
                        break
                    time.sleep(0.1)
                if others:
                    status += " | other ranks: " + " | ".join(others)
            line = {"metric": METRIC, "n_gpus": world, "status": status, "phase": state["phase"], "transport": transport}
            line.update(extra or {})
            print(json.dumps(line), flush=True)
            try:
                open(note_base + ".ack", "w").write("printed")
            except OSError:
                pass
        else:
            sys.stderr.write(f"rank {rank}: {status}\n")
            try:
                with open(f"{note_base}.rank{rank}", "w") as f:
                    f.write(status)
            except OSError:
                pass
            # a launcher (torch.distributed.run) terminates the other ranks the moment one exits non-zero: rank 0 must have had the
            # time to print its line first - wait for its acknowledgement (at most 20 s)
            for _ in range(200):
                if os.path.exists(note_base + ".ack"):
                    break
                time.sleep(0.1)
        os._exit(code)

    if world > 1 and rank == 0:
        def watch_notes():
            while True:
                time.sleep(0.5)
                others = read_notes()
                if others:
                    die(4, f"failed on {others[0].split(':', 1)[0]} during the {state['phase']}: " + " | ".join(o.split(": ", 1)[1] for o in others))

        threading.Thread(target=watch_notes, daemon=True).start()
        # rank 0 must get its line out even when the launcher ends the job (SIGTERM after another rank crashed without a word) while
        # the main thread sits inside a collective of the library, where no Python signal handler runs: CPython's C-level handler
        # writes the signal's number to a wake-up socket at once, whichever thread the kernel picks, and a thread of its own reads it
        import signal
        import socket

        term_r, term_w = socket.socketpair()
        term_w.setblocking(False)
        signal.set_wakeup_fd(term_w.fileno(), warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda signum, frame: None)

        def wait_term():
            signum = term_r.recv(1)[0]
            if os.environ.get("RECPACK_BENCH_DEBUG"):
                sys.stderr.write(f"[debug] rank 0 woke up for signal {signum}\n")
            if signum != signal.SIGTERM:
                return
            die(7, f"terminated by the launcher (signal {signum}) during the {state['phase']}" +
                (f"; last collective entered: {state.get('collective')}" if state.get("collective") else ""))

        threading.Thread(target=wait_term, daemon=True).start()
    setup_dog = None
    if world > 1:
        setup_dog = threading.Timer(min(args.timeout, 600.0), lambda: die(3, f"timeout: the {state['phase']} did not finish within {min(args.timeout, 600.0):.0f} s"))
        setup_dog.daemon = True
        setup_dog.start()
    if world > 1:  # the library's communicator: every collective of the step goes through it
        # The choice of transport is COLLECTIVE: a rank whose RCCL communicator fails to come up must not fall back alone (the
        # others would wait in ncclBroadcast for a rank that talks through torch.distributed): every rank reports, and if any
        # failed they all switch - or, with an explicit --ease-transport rccl, the run stops with a status that says so.
        want = "rccl" if (args.ease_transport == "rccl" or (args.ease_transport == "auto" and backend == "nccl")) else "torch"
        err = None
        if want == "rccl":
            try:
                comm = rdist.make_comm(ctx, device, transport="rccl")
            except Exception as e:  # noqa: BLE001 - e.g. librccl refuses to initialise
                err = f"{type(e).__name__}: {e}"
                sys.stderr.write(f"rank {rank}: library RCCL communicator failed ({err})\n")
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm.close()
                    comm = None
                comm_note = f"library RCCL communicator failed on some rank ({err or 'another rank'})"
                if args.ease_transport == "rccl":
                    die(5, "failed: --ease-transport rccl was asked for and " + comm_note, {"transport": "none"})
                want = "torch"
        if comm is None:
            comm = rdist.make_comm(ctx, device, transport="torch")
        transport = "rccl (library communicator)" if comm.is_rccl else "torch.distributed callbacks" + (f" (FALLBACK: {comm_note})" if comm_note else "")
    # Transport self-test BEFORE the warm-up (N > 1; N = 1 with --selftest): each of the four operations once on a known pattern,
    # every received word checked on the device, 30 s of watchdog per operation - a hang or a wrong byte is reported by NAME with a
    # non-zero exit code instead of a solve that waits for ever in its first pivot (the RCCL transport had never run with more than
    # one rank when this was written: the pool leases single-GPU boxes).
    selftest = None
    if world > 1 or args.selftest:
        st_comm = comm if comm is not None else rdist.make_comm(ctx, device, transport="rccl")

        state["phase"] = "transport self-test"

        def hang(name):
            die(6, f"failed: transport self-test: {name} did not complete within 30 s", {"transport_selftest": {"ok": False, "hung": name}})

        def run_selftest():
            res = st_comm.selftest(words=1 << 18, watchdog_s=30.0, on_hang=hang)
            if world > 1:  # (everybody must agree before anybody goes on: per operation, the smallest flag over the ranks)
                names = list(res["ops"])
                okt = torch.tensor([1 if res["ops"][k]["mismatches"] == 0 else 0 for k in names], dtype=torch.int32, device=device)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                res["failed_on_some_rank"] = [k for k, f in zip(names, okt.tolist()) if not f]
                res["ok_all_ranks"] = not res["failed_on_some_rank"]
            return res

        selftest = run_selftest()
        if selftest.get("failed_on_some_rank") == ["broadcast_large"]:
            # only the large broadcast (scatter + all-gather by grouped point-to-point sends) delivered wrong words: every rank
            # switches to the transport's own broadcast for all sizes and the test runs again - slower pivots, not a failed run
            ctx.set_option("comm_broadcast_split", 0)
            first = selftest
            selftest = run_selftest()
            selftest["broadcast_split"] = "off: scatter + all-gather delivered wrong words on some rank (" + \
                f"{first['ops']['broadcast_large']['mismatches']} here); the transport's own broadcast is used for every size"
        if comm is None:
            st_comm.close()
        if not selftest.get("ok_all_ranks", selftest["ok"]):
            bad = ", ".join(f"{k}: {v['mismatches']} wrong words" for k, v in selftest["ops"].items() if v["mismatches"]) or \
                "wrong words on another rank in " + ", ".join(selftest.get("failed_on_some_rank", []))
            die(6, "failed: transport self-test: " + bad, {"transport_selftest": selftest})
    if setup_dog is not None:
        setup_dog.cancel()

    keep = {}
    parts = {"itemknn": 0.0, "ease_fit": 0.0, "ease_predict": 0.0}
    state["phase"] = "warm-up"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        keep.clear()  # the step before's models and matrices go back to the context's block cache BEFORE the new ones are asked for
        if rank == args.inject_failure_rank:
            if args.inject_failure_hard:
                os._exit(9)  # (no note, no word: a crash)
            raise RuntimeError("injected failure (test hook --inject-failure-rank)")
        t0 = time.perf_counter()
        if world > 1:  # K0 with the degree pass sharded by item rows + one all-reduce of the degree vector
            state["collective"] = "all-reduce of the item degrees (K0, rphip_comm_all_reduce_sum_i32)"
            dX = rdist.csr_for_shard(ctx, indptr_t, indices_t, (U, I), row_parts[rank], device, comm=comm)
        else:
            dX = _lib.DeviceCSR.from_device(ctx, indptr_t, indices_t, (U, I))  # K0
        if world == 1:
            model = _lib.itemknn_fit(ctx, dX, args.K)  # K1-K3
        else:
            state["collective"] = "all-gather of the neighbour lists (rphip_itemknn_gather)"
            model = rdist.fit_itemknn_sharded(ctx, dX, args.K, row_parts, device, comm=comm)  # K1-K3 + all-gather
        _lib.predict_topk_dev(ctx, model, dX, Kp, out_idx, out_val, out_nnz, user_begin=ub, user_end=ue)  # K8
        keep["model"], keep["dX"] = model, dX
        if not ease_on:
            return
        torch.cuda.synchronize()  # (the two halves are timed apart; the ranks are NOT held together here)
        t1 = time.perf_counter()
        if world == 1:
            em = _lib.ease_fit(ctx, dX, args.l2, 0.0)  # K4-K7 (X^T and the degrees are the ItemKNN half's)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            _lib.predict_topk_dev(ctx, em, dX, Kp, e_idx, e_val, e_nnz)  # K9
            keep["k9_stats"] = _lib.predict_stats(ctx)  # (host fields of the context: which users took which pass, for `k9_predict`)
        else:
            dXe = _lib.DeviceCSR.from_device(ctx, indptr_t, indices_t, (U, I))  # (the ItemKNN shard holds its own rows of X^T only)
            state["collective"] = "distributed EASE solve (rphip_ease_fit_dist: pivot broadcasts, panel exchange)"
            em, _ = rdist.ease_fit_sharded(ctx, dXe, args.l2, 0.0, device, comm=comm)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            # predict.  First choice (round 5): the owner route - only the int8 rows are gathered, every rank runs the first pass for
            # ITS users, the candidates' exact sums come from every rank's own panels (rphip_predict_topk_owner); refused by every
            # rank alike (RPHIP_ELIMIT / RPHIP_ENOMEM) when it does not apply.  Then: while the whole of B fits every GPU, gather the
            # panels and score this rank's users with the single-GPU kernels; otherwise every rank scores all users against its
            # own columns + merge
            full = None
            owner_done = False
            #  the route's static limits are checked ONCE, here, not by gathering n^2 bytes of int8 rows in every step for
            # a call that then refuses: Kc = ceil64(Kp + 92) <= 256, 4 Kc <= items (rows of X are sorted: powerlaw_csr); a refusal the
            # ranks agreed on at run time (memory, the exact pass's limits) is remembered for the steps that follow
            owner_ok = args.ease_predict_route in ("auto", "owner") and not state.get("owner_refused")
            if owner_ok and args.ease_predict_route == "auto":
                kc = (Kp + 92 + 63) // 64 * 64
                owner_ok = kc <= 256 and 4 * kc <= I
            if owner_ok:
                rows8 = None
                try:
                    state["collective"] = "all-gather of the int8 rows (rphip_ease_model_allgather_i8)"
                    rows8 = rdist.ease_gather_rows_i8(ctx, em, I, comm)
                    state["collective"] = "owner-route predict (rphip_predict_topk_owner: candidate all-gather, part exchange)"
                    _lib.predict_topk_owner(ctx, comm, em, rows8, dXe, [p[0] for p in user_parts] + [user_parts[-1][1]], Kp, out_idx, out_val, out_nnz)
                    keep["ease_rows8"] = rows8
                    keep["ease_predict_route"] = "int8 rows all-gathered, users sharded, exact sums by the columns' owners"
                    owner_done = True
                except _lib.RecpackHipError as e:  # (decided before the first collective or agreed on by the ranks: everybody lands here or nobody)
                    if rows8 is not None:
                        rows8.close()  # (n^2 bytes back to the block cache now, not at the next collection)
                    if e.code not in (_lib.RPHIP_ENOMEM, _lib.RPHIP_ELIMIT, _lib.RPHIP_ESTATE) or args.ease_predict_route == "owner":
                        raise
                    state["owner_refused"] = f"{type(e).__name__}: {e}"
            if owner_done:
                pass
            elif rdist.ease_gather_fits(I, _lib.ease_fit_dist_stats(ctx)["panel"], info["hbm_bytes"], world):
                try:
                    state["collective"] = "all-gather of the model's panels (rphip_ease_model_allgather)"
                    full = rdist.ease_gather_model(ctx, em, I, comm)
                except _lib.RecpackHipError as e:  # (the ranks agree on their allocations: everybody lands here or nobody)
                    if e.code != _lib.RPHIP_ENOMEM:
                        raise
            if owner_done:
                pass
            elif full is not None:
                _lib.predict_topk_dev(ctx, full, dXe, Kp, out_idx, out_val, out_nnz, user_begin=ub, user_end=ue)
                keep["ease_full"] = full
                keep["ease_predict_route"] = "panels all-gathered, users sharded"
            else:
                state["collective"] = "column-sharded predict (rphip_predict_topk_dist: list exchange)"
                rdist.predict_topk_sharded(ctx, em, dXe, Kp, device, exclude_history=False, comm=comm)
                keep["ease_predict_route"] = "columns sharded, lists exchanged and merged"
            keep["dXe"] = dXe
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if state["timing"]:
            parts["itemknn"] += t1 - t0
            parts["ease_fit"] += t2 - t1
            parts["ease_predict"] += t3 - t2
        keep["ease_model"] = em

    result = {"metric": METRIC, "status": "starting", "n_gpus": world}

    def give_up():  # every rank has the same timer: a hung collective or an overrun costs the run, and says so
        die(3, f"timeout: the {state['phase']} did not finish within {args.timeout:.0f} s (--timeout)" +
            (f"; last collective entered: {state.get('collective')}" if state.get("collective") else ""))

    dog = threading.Timer(args.timeout, give_up)
    dog.daemon = True
    dog.start()
    try:
        cold_s = None
        for w in range(args.warmup):
            if w == 0:  # the first step of a fresh context: every allocation of the step is a hipMalloc (later steps take cached blocks)
                torch.cuda.synchronize()
                tc = time.perf_counter()
            step()
            if w == 0:
                torch.cuda.synchronize()
                cold_s = time.perf_counter() - tc
        barrier()
        ctx.reset_timings()
        state["phase"], state["timing"] = "timed steps", True
        # clock and power of this rank's GPU during the timed steps (a sysfs sampler thread: recpack_amd/devmon.py) - the boxes of a
        # pool differ by several per cent on the MFMA-bound solve, and the line should say which kind of box it was measured on
        from recpack_amd.devmon import DeviceMonitor

        mon = DeviceMonitor(local_rank).start() if rank == 0 else None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        if mon is not None:
            mon.stop()
        state["timing"] = False
    except Exception as e:  # noqa: BLE001 - the other ranks may be stuck in a collective this rank left: report and leave, no barrier
        dog.cancel()
        die(4, f"failed in the {state['phase']}" + (f" (last collective entered: {state.get('collective')})" if state.get("collective") else "") +
            f": {type(e).__name__}: {e}")
    dog.cancel()
    if world > 1:
        t = torch.tensor([elapsed, parts["itemknn"], parts["ease_fit"], parts["ease_predict"]], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, parts["itemknn"], parts["ease_fit"], parts["ease_predict"] = (float(v) for v in t.tolist())
    tm = ctx.timings()
    if rank == 0:
        sec_per_step = elapsed / args.steps
        # The SpGEMM (K1-K3) runs as two instances of one kernel: k_itemknn_fit<false,1024,16> (rows with more than 255
        # users: u16 counter pairs, no candidate list, two 1024-thread workgroups per CU) and k_itemknn_fit<false,256,8>
        # (lighter rows: byte counters, first-touch lists, 256-thread workgroups, three per CU).  Each is priced with its own share of the 4*P byte model
        # (4*P + 8*nnz + 4*(U+I+2) + 8*I*K); `roofline` is the one that takes longer.
        cnt = ctx.counters()
        p_rank = P_total / world
        p_light = float(min(cnt[0], p_rank))
        p_heavy = p_rank - p_light
        rows0 = row_parts[0][1] - row_parts[0][0]
        other = 8.0 * nnz + 4.0 * (U + I + 2) + 8.0 * rows0 * args.K
        legs = {}
        for name, slot, p_share in (("k_itemknn_fit<false,1024,16>", "gram_topk", p_heavy), ("k_itemknn_fit<false,256,8>", "gram_topk_wave", p_light)):
            t = tm[slot]
            if t["launches"]:
                avg = t["ms"] / t["launches"]
                by = 4.0 * p_share + other * (p_share / p_rank if p_rank else 0.0)
                legs[name] = (avg, by)
        # K8 priced with its own byte model: 8*K*nnz + 4*(nnz+U) + 8*U*Kp per rank share (K9 has its own timing slot)
        u_rank = user_parts[0][1] - user_parts[0][0]
        share = u_rank / U
        p_bytes = 8.0 * args.K * nnz * share + 4.0 * (nnz * share + u_rank) + 8.0 * u_rank * Kp
        k8 = tm["predict"]
        if k8["launches"]:
            legs["k_predict_wave<unsigned int, 4096, false>"] = (k8["ms"] / k8["launches"], p_bytes)
        leg_what = {"k_itemknn_fit<false,1024,16>": "K1-K3 heavy rows: Gram + similarity key + top-K",
                    "k_itemknn_fit<false,256,8>": "K1-K3 light rows: Gram + similarity key + top-K",
                    "k_predict_wave<unsigned int, 4096, false>": "K8: scores X.S + fused top-Kp, 32-bit per-user-scaled fixed-point sums, 4096-column "
                                                                 "tiles (bound by instruction issue and LDS round trips per (user, tile) in "
                                                                 "practice: the 400 MB model stays in L2/MALL)"}
        # `roofline` = the HBM-bound kernel with the longest launch; all of them are under `legs`
        dom = max(legs, key=lambda k: legs[k][0])
        avg_ms, algo_bytes = legs[dom]
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        spgemm_ms = sum(tm[k]["ms"] for k in ("gram_topk", "gram_topk_wave", "gram_finalize", "topk_merge")) / args.steps
        spgemm_bytes = 4.0 * p_rank + other
        phases = {k: round(v["ms"] / args.steps, 4) for k, v in tm.items() if v["launches"]}
        traffic, traffic_note = pmc_traffic(args.config, dom, world)
        # (ii) of SURVEY.md 8(d)'s three SpGEMM numbers: the bytes the PMC pass saw for every kernel of the K1-K3 group (per launch),
        # over the group's time in THIS run
        sp_traffic = {}
        for kname in ("k_itemknn_fit<false,1024,16>", "k_itemknn_fit<false,256,8>", "k_itemknn_fit<true,1024,32>", "k_itemknn_merge_wave", "k_itemknn_merge("):
            tb, _ = pmc_traffic(args.config, kname, world)
            if tb:
                sp_traffic[kname.rstrip("(")] = tb
        sp_traffic_sum = sum(sp_traffic.values()) if sp_traffic else None
        result.update({
            "value": round(sec_per_step, 6),
            "unit": "s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 3),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("ItemKNN: int32 counts, f64 similarity, i32 fixed-point score sums (per-user power-of-two scale) of f32 similarities; "
                      "EASE: f32 storage, products as 2 x fp16 planes on the fp16 MFMA with f32 accumulate + compensated sums, f64 diagonal blocks"),
            "data": "synthetic",
            "status": "ok" if ease_on else "ok (ItemKNN half only: --no-ease-in-step)",
            # the two halves of the metric, seconds per step (host clock around each half; max over the ranks)
            "value_itemknn": round(parts["itemknn"] / args.steps, 6) if ease_on else round(sec_per_step, 6),
            "value_ease": round((parts["ease_fit"] + parts["ease_predict"]) / args.steps, 6) if ease_on else None,
            "value_ease_fit": round(parts["ease_fit"] / args.steps, 6) if ease_on else None,
            "value_ease_predict": round(parts["ease_predict"] / args.steps, 6) if ease_on else None,
            "config": {
                "workload": f"{args.config}: {U} users x {I} items, nnz {nnz} (P={P_total} scalar products): ItemKNN cosine K={args.K} fit + predict "
                            f"top-{Kp} for all users" + (f", then EASE l2={args.l2} fit + predict top-{Kp} for all users" if ease_on else ""),
                "parallelism": (f"ItemKNN: item-row shards x{world} (fit) + user shards x{world} (predict); EASE: block-cyclic column panels x{world} "
                                f"(library solve, {'own RCCL communicator' if comm is not None and comm.is_rccl else 'torch.distributed callbacks'}) + "
                                f"predict: {keep.get('ease_predict_route', '-')}") if world > 1 else "1 GPU",
                "generate_s": round(t_gen, 2),
                "device": dict(info, under_load=mon.summary() if mon is not None else None,
                               sclk_mhz_under_load=(mon.summary().get("sclk_mhz") if mon is not None else None),
                               power_w_under_load=(mon.summary().get("power_w") if mon is not None else None)),
                "transport_note": comm_note,
            },
            "transport": transport,  # N > 1: what the collectives of the step ran on (null at N = 1: there are none)
            "transport_selftest": selftest,  # N > 1 (or --selftest): the four operations checked before the warm-up
            "phases_ms_per_step": phases,
            "throughput": {
                "interactions_per_s": round(nnz / sec_per_step, 1),
                "gram_products_per_s": round(p_rank / (spgemm_ms * 1e-3), 1) if spgemm_ms > 0 else None,
            },
            "roofline": {
                "kernel": dom + " (" + leg_what[dom] + ")",
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_ms": round(avg_ms, 4),
                "legs": {k: {"what": leg_what[k], "avg_launch_ms": round(v[0], 4), "algorithmic_bytes": v[1],
                             "GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None,
                             "frac": round(v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[0] > 0 else None,
                             "traffic": sp_traffic.get(k) if k in sp_traffic else (traffic if k == dom else None),
                             "traffic_over_algorithmic": round(sp_traffic[k] / v[1], 3) if k in sp_traffic and v[1] > 0 else None} for k, v in legs.items()},
                # the three numbers of the SpGEMM - (i) the 4*P model above, (ii) bytes the PMC
                # counters saw / time, (iii) accumulator updates per second against the LDS-atomic peak measured with
                # tools/lds_atomic_bench.hip on MI355X (u32 adds, no return: 4.8 per clock per CU, 2.7e12/s chip-wide)
                "measured_fabric_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                "accumulator_updates": {"per_s": round(p_rank / (spgemm_ms * 1e-3), 1) if spgemm_ms > 0 else None,
                                        "lds_atomic_peak_per_s": LDS_ATOMIC_PEAK,
                                        "frac": round(p_rank / (spgemm_ms * 1e-3) / LDS_ATOMIC_PEAK, 4) if spgemm_ms > 0 else None},
                "spgemm_all_kernels": {"ms": round(spgemm_ms, 4), "algorithmic_bytes": spgemm_bytes,
                                       "GBps": round(spgemm_bytes / (spgemm_ms * 1e-3) / 1e9, 1) if spgemm_ms > 0 else None,
                                       "frac": round(spgemm_bytes / (spgemm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if spgemm_ms > 0 else None,
                                       # (ii): fabric bytes of the committed PMC pass (2 x FETCH_SIZE + WRITE_SIZE per launch, every kernel
                                       # of the group) over this run's time; (iii) is `accumulator_updates` above
                                       "traffic": sp_traffic_sum, "traffic_per_kernel": sp_traffic or None,
                                       "measured_fabric_GBps": round(sp_traffic_sum / (spgemm_ms * 1e-3) / 1e9, 1) if sp_traffic_sum and spgemm_ms > 0 else None,
                                       "traffic_over_algorithmic": round(sp_traffic_sum / spgemm_bytes, 3) if sp_traffic_sum else None},
            },
        })
        result["cold_first_step_s"] = round(cold_s, 4) if cold_s is not None else None  # (needs --warmup >= 1; `value` is the warm rate)
        if ease_on:
            # the headline roofline is the DOMINANT kernel's: the EASE solve (two thirds of the step) against the dense 16-bit MFMA peak;
            # the HBM-bound ItemKNN kernels (K8, the SpGEMM instances) follow under `roofline_hbm`
            re_ = roofline_ease(tm, I, world, args.steps, _lib.ease_fit_dist_stats(ctx) if world > 1 else None, nnz, U, Kp, keep.get("k9_stats"))
            hbm = result.pop("roofline")
            k9 = re_.pop("k9_predict", None)
            re_["share_of_step"] = round(re_["solve_ms_per_step"] / (sec_per_step * 1e3), 4)
            re_["traffic"] = None  # (an MFMA-bound kernel: the operand traffic of the solve is in profiles/, not a bound)
            result["roofline"] = re_
            result["roofline_hbm"] = hbm
            if k9:
                result["k9_predict"] = k9
        if world == 1 and args.cpu_baseline_seconds > 0:
            idx, val, nnz_rows = keep["model"].topk_download()
            from recpack_amd.algorithms import topk_arrays_to_csr

            S = topk_arrays_to_csr(idx, val, nnz_rows, I)
            host_csr = (indptr_t.cpu().numpy(), indices_t.cpu().numpy())
            port = cpu_baseline(host_csr[0], host_csr[1], (U, I), args.K, Kp, S, args.cpu_baseline_seconds, P_total)
            ease_cpu = cpu_baseline_ease(host_csr[0], host_csr[1], (U, I), args.l2, Kp) if ease_on else None
            ref = port.pop("reference_shaped")
            # top level: the route recpack itself runs (sklearn cosine + per-row argpartition loop; float64 np.linalg.inv), both halves
            result["cpu_baseline"] = {
                "value": round(ref["value"] + (ease_cpu["value"] if ease_cpu else 0.0), 1), "unit": "s",
                "cores": ease_cpu["cores"] if ease_cpu else 1, "kind": "port",
                "sample": ("ItemKNN half, reference-shaped, 1 thread (scipy's SpGEMM and the per-row loops are single-threaded): " + ref["sample"] +
                           (" | EASE half, all BLAS threads: " + ease_cpu["sample"] if ease_cpu else "")),
                "itemknn_reference_shaped_s": ref["value"], "ease_s": ease_cpu["value"] if ease_cpu else None,
                "ease_measured_points": ease_cpu["measured_points"] if ease_cpu else None,
                "port": port,  # the oracle's own vectorised route (what the parity tests run)
                "host_cpus": os.cpu_count(),
            }
        if world == 1 and args.parity_check:
            try:
                result["parity"] = parity_verdict(args, ctx, device, indptr_t, indices_t, (U, I), keep["model"], keep.get("ease_model") if ease_on else None)
            except Exception as e:  # noqa: BLE001 - the checker must not cost the measurement: the line says what happened
                result["parity"] = {"error": f"{type(e).__name__}: {e}"}
            if not (result["parity"].get("itemknn_ok", False) and result["parity"].get("ease_ok", True)):
                result["status"] = "parity check failed" if "error" not in result["parity"] else "parity check did not run: " + result["parity"]["error"]
        keep.clear()
        if world == 1 and args.itemknn_config.lower() != "none":
            torch.cuda.empty_cache()
            for cfg in (c.strip().upper() for c in args.itemknn_config.split(",")):
                result[f"itemknn_{cfg.lower()}"] = itemknn_leg(args, ctx, device, Kp, cfg, 3)
        if world == 1 and args.ease_config.lower() != "none":
            del out_idx, out_val, out_nnz, e_idx, e_val, e_nnz
            torch.cuda.empty_cache()  # (no ctx.trim(): the legs' smaller matrices are cut from the blocks the step left in the cache)
            for cfg in (c.strip().upper() for c in args.ease_config.split(",")):
                result[f"ease_{cfg.lower()}"] = ease_leg(args, ctx, device, Kp, cfg, 1 if CONFIGS[cfg][1] > 60000 else 2)
                torch.cuda.empty_cache()
        if world == 1 and args.tars_config.lower() != "none":
            torch.cuda.empty_cache()
            result["time_aware"] = tars_leg(args, ctx, device, Kp, args.tars_config.strip().upper())
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
