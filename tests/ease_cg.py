"""Float64 reference columns of the EASE closed form at sizes no host can invert (test infrastructure, not product).

The reference computes B = -P / diag(P) with P = (X^T X + lambda I)^-1 by a float64 LAPACK inverse
(/root/reference/recpack/algorithms/ease.py:81-86).  Column j of P solves the SPD system (X^T X + lambda I) p = e_j, so
for sampled columns the float64 answer is available without the inverse:

1. `solve_columns`: Jacobi-preconditioned conjugate gradients in float64, all sampled columns at once, with
   G v = X^T (X v) as two sparse products.  It runs wherever its tensors live (the GPU box's device for C4 / C5 - plain
   torch gathers and index_add, no code of the library under test).
2. `certify_on_host`: WHERE the iterate came from does not matter - the host recomputes the residual
   r_j = e_j - (X^T (X p_j) + lambda p_j) in float64 with scipy, and because lambda_min(X^T X + lambda I) >= lambda the
   true solution is within ||r_j||_2 / lambda of p_j.  The returned bound on |B_ref - B_exact| is asserted to be
   orders of magnitude below the 1e-4 parity bar, so the comparison is against a host-certified float64 reference.
"""
import numpy as np


from oracle.ease_probe import GramOps as _Ops  # noqa: E402  (X v and X^T w in float64: shared with the whole-matrix probe)


def solve_columns(indptr_t, indices_t, shape, l2, cols, tol=1e-13, max_iter=3000, log=None):
    """P[:, cols] of P = (X^T X + l2 I)^-1 by preconditioned CG in float64.  Returns (P_cols [I, k] as numpy, iterations, the
    solver's own last relative residual)."""
    import torch

    ops = _Ops(indptr_t, indices_t, shape)
    dev = indices_t.device
    I = shape[1]
    k = len(cols)
    lam = float(np.float32(l2))  # the reference adds l2 * identity(float32): lambda is rounded to float32 (ease.py:82)
    E = torch.zeros((I, k), dtype=torch.float64, device=dev)
    E[torch.as_tensor(np.asarray(cols), device=dev), torch.arange(k, device=dev)] = 1.0
    Minv = (1.0 / (ops.deg + lam))[:, None]
    Xk = torch.zeros_like(E)
    R = E.clone()
    Z = Minv * R
    Pd = Z.clone()
    rz = (R * Z).sum(0)
    it = 0
    rel = float("inf")
    for it in range(1, max_iter + 1):
        AP = ops.gram_apply(Pd) + lam * Pd
        alpha = rz / (Pd * AP).sum(0)
        Xk += alpha * Pd
        R -= alpha * AP
        rel = float(R.norm(dim=0).max())  # ||e_j|| = 1
        if rel <= tol:
            break
        Z = Minv * R
        rz_new = (R * Z).sum(0)
        Pd = Z + (rz_new / rz) * Pd
        rz = rz_new
        if log and it % 50 == 0:
            log(f"  cg iteration {it}: max residual {rel:.2e}")
    return Xk.cpu().numpy(), it, rel


def certify_on_host(X_csr, l2, cols, P_cols, alpha=0.0):
    """Host float64 residual of the CG iterate -> reference columns of B with a proven error bound.

    Returns (B_ref [I, k], bound [k]): |B_ref[:, j] - B_exact[:, j]|_max <= bound[j]."""
    lam = float(np.float32(l2))
    cols = np.asarray(cols)
    k = len(cols)
    Xf = X_csr.astype(np.float64)
    R = -(Xf.T @ (Xf @ P_cols) + lam * P_cols)
    R[cols, np.arange(k)] += 1.0
    dp = np.linalg.norm(R, axis=0) / lam  # ||p_j - p_exact||_2 <= ||r_j||_2 / lambda_min, lambda_min >= lambda
    pjj = P_cols[cols, np.arange(k)]
    assert (pjj > dp).all()
    B_ref = -P_cols / pjj[None, :]
    B_ref[cols, np.arange(k)] = 0.0
    # b = -p / p_jj: |db| <= dp / (p_jj - dp) * (1 + max|b|)
    bound = dp / (pjj - dp) * (1.0 + np.abs(B_ref).max(axis=0))
    if alpha != 0:  # ease.py:88-90: column j times c_j^-alpha (a sampled column always has c_j > 0 or is all zero)
        c = np.bincount(X_csr.indices, minlength=X_csr.shape[1])[cols].astype(np.float64)
        w = np.where(c > 0, np.maximum(c, 1.0) ** -float(alpha), 0.0)
        B_ref = B_ref * w[None, :]
        bound = bound * w
    return B_ref, bound


def sample_columns(item_deg, n_popular, n_median, n_random, seed=3):
    I = len(item_deg)
    rng = np.random.default_rng(seed)
    order = np.argsort(-item_deg, kind="stable")
    return np.unique(np.concatenate([order[:n_popular], order[I // 2 - n_median // 2: I // 2 + (n_median + 1) // 2],
                                     rng.choice(I, n_random, replace=False)])).astype(np.int64)


def run_parity(_lib, ctx, cfg, l2=200.0, n_cols=(16, 8, 8), n_users=48, Kp=100, log=print, model=None, keep_model=False, alpha=0.0):
    """EASE fit on config `cfg` through the C ABI, then: sampled columns of B against the host-certified float64
    columns, and scores of a block of users at those columns (the kernel behind predict, K9) against float64 sums of
    the reference columns.  Returns the measured errors; the caller asserts."""
    import time

    import scipy.sparse as sp
    import torch

    from recpack_amd import dist as rdist
    from recpack_amd.synth import CONFIGS, config_csr

    dev = torch.device("cuda", 0)
    U, I, _ = CONFIGS[cfg]
    ip, ix = config_csr(cfg, device=dev)
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    t0 = time.perf_counter()
    m = model if model is not None else _lib.ease_fit(ctx, dX, l2, alpha)
    ctx.sync()
    t_fit = time.perf_counter() - t0
    B = rdist.model_dense_tensor(m, dev)
    assert float(torch.diagonal(B).abs().max()) == 0.0
    deg = torch.bincount(ix.to(torch.int64), minlength=I).cpu().numpy()
    cols = sample_columns(deg, *n_cols)
    t0 = time.perf_counter()
    P_cols, iters, rel = solve_columns(ip, ix, (U, I), l2, cols, log=log)
    torch.cuda.synchronize()
    t_cg = time.perf_counter() - t0
    indptr, indices = ip.cpu().numpy().astype(np.int64), ix.cpu().numpy()
    X = sp.csr_matrix((np.ones(len(indices), dtype=np.float64), indices, indptr), shape=(U, I))
    t0 = time.perf_counter()
    B_ref, bound = certify_on_host(X, l2, cols, P_cols, alpha)
    t_host = time.perf_counter() - t0
    B_gpu = B[:, torch.as_tensor(cols, device=dev)].double().cpu().numpy()
    col_max = np.abs(B_ref).max(axis=0)
    err_col = np.abs(B_gpu - B_ref).max(axis=0) / np.where(col_max > 0, col_max, 1.0)
    # scores of a contiguous block of users (synthetic users are i.i.d.) at the sampled columns, through predict's kernel
    u0 = U // 3
    u1 = u0 + n_users
    pip, pidx, pdat = _lib.predict_full(ctx, m, dX, u0, u1)
    S_gpu = sp.csr_matrix((pdat, pidx, pip), shape=(n_users, I))
    S_ref = X[u0:u1] @ B_ref  # [n_users, k] float64
    S_got = S_gpu[:, cols].toarray()
    row_max = np.abs(S_gpu).max(axis=1).toarray().ravel()
    live = row_max > 0
    err_score = (np.abs(S_got - S_ref).max(axis=1)[live] / row_max[live])
    # how many of the checked (user, item) pairs are in that user's top-Kp list
    pi, pv, pn = _lib.predict_topk(ctx, m, dX, Kp, exclude_history=True, user_begin=u0, user_end=u1)
    in_top = int(sum(np.isin(cols, pi[r, :pn[r]]).sum() for r in range(n_users)))
    top_err = 0.0
    for r in range(n_users):
        got = pi[r, :pn[r]]
        hit = np.nonzero(np.isin(got, cols))[0]
        if len(hit) and row_max[r] > 0:
            ref = S_ref[r, np.searchsorted(cols, got[hit])]
            top_err = max(top_err, float(np.abs(pv[r, hit] - ref).max() / row_max[r]))
    live_c = col_max > 0  # (an item nobody interacted with has an all-zero column)
    err_col = err_col[live_c] if live_c.any() else np.zeros(1)
    out = {"config": cfg, "n_items": I, "l2": l2, "alpha": alpha, "columns": len(cols), "cg_iterations": iters, "cg_residual": rel,
           "certified_bound_rel": float((bound / np.where(col_max > 0, col_max, 1.0)).max()), "err_col_max": float(err_col.max()), "err_col_median": float(np.median(err_col)),
           "err_global": float(np.abs(B_gpu - B_ref).max() / col_max.max()), "err_score_max": float(err_score.max()),
           "topk_pairs_checked": in_top, "err_topk_score_max": top_err, "c_max": int(deg.max()),
           "fit_s": round(t_fit, 2), "cg_s": round(t_cg, 2), "host_certify_s": round(t_host, 2)}
    log("EASE parity " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in out.items()))
    if keep_model:
        out["model"], out["dX"], out["X"] = m, dX, X
    return out
