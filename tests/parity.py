"""Parity rules shared by the CPU and GPU tests."""
import numpy as np
import scipy.sparse as sp


def rows_of(M):
    M = sp.csr_matrix(M)
    return [(M.indices[a:b], M.data[a:b]) for a, b in zip(M.indptr[:-1], M.indptr[1:])]


def tie_aware_topk(cand: sp.csr_matrix, ref_sel: sp.csr_matrix, test_idx, test_nnz, eps=1e-9, positive_only=True, abs_tol=None):
    """cand: all candidate scores per row (reference values); ref_sel: the reference's selection;
    test_idx[r,:test_nnz[r]]: columns chosen by the implementation under test.

    With t = the reference's smallest selected value in the row:
      MUST = {v > t + eps|t|}, MAY = {|v-t| <= eps|t|}; require MUST <= Sel <= MUST u MAY, |Sel| = K'.
    abs_tol[r] (optional) widens row r's band by an absolute amount (an implementation whose values carry a stated
    absolute error bound: two values closer than twice the bound may legitimately swap).
    Returns (bad_rows list, exact_equal_count).
    """
    cand = sp.csr_matrix(cand)
    ref_sel = sp.csr_matrix(ref_sel)
    bad, exact = [], 0
    for r in range(cand.shape[0]):
        a, b = cand.indptr[r], cand.indptr[r + 1]
        cols, vals = cand.indices[a:b], cand.data[a:b]
        if positive_only:
            keep = vals > 0
            cols, vals = cols[keep], vals[keep]
        ra, rb = ref_sel.indptr[r], ref_sel.indptr[r + 1]
        rcols, rvals = ref_sel.indices[ra:rb], ref_sel.data[ra:rb]
        sel = np.asarray(test_idx[r, : int(test_nnz[r])])
        kp = rcols.shape[0]
        if sel.shape[0] != kp or np.unique(sel).shape[0] != kp:
            bad.append(r)
            continue
        if kp == 0:
            exact += 1
            continue
        t = rvals.min()
        band = eps * abs(t) + (float(abs_tol[r]) if abs_tol is not None else 0.0)
        must = set(cols[vals > t + band].tolist())
        may = set(cols[np.abs(vals - t) <= band].tolist())
        s = set(sel.tolist())
        if not (must <= s <= (must | may)):
            bad.append(r)
        if s == set(rcols.tolist()):
            exact += 1
    return bad, exact


def topk_vals_close(ref_sel: sp.csr_matrix, test_idx, test_val, test_nnz, rtol):
    """Every selected (row, col) that the reference also selected must carry the same value."""
    ref = sp.csr_matrix(ref_sel).todok() if ref_sel.shape[0] * ref_sel.shape[1] < 5_000_000 else None
    ref_csr = sp.csr_matrix(ref_sel)
    worst = 0.0
    for r in range(ref_csr.shape[0]):
        a, b = ref_csr.indptr[r], ref_csr.indptr[r + 1]
        lut = dict(zip(ref_csr.indices[a:b].tolist(), ref_csr.data[a:b].tolist()))
        for c, v in zip(test_idx[r, : int(test_nnz[r])].tolist(), test_val[r, : int(test_nnz[r])].tolist()):
            if c in lut:
                worst = max(worst, abs(v - lut[c]) / abs(lut[c]))
    assert worst <= rtol, worst
    return worst


def fx32_bound(hist: sp.csr_matrix, S: sp.csr_matrix) -> np.ndarray:
    """Per-user bound on |score - exact sum| of the default 32-bit predict sums (include/recpack_hip.h, option
    "predict_exact64"): terms are rint(s_ij * 2^(30-e)) with M_u (1 + 2^-12) < 2^e, M_u = the sum over the user's history of
    the lists' largest |similarity| (f32); every term is off by <= half a unit, so a score by <= d_u * 2^(e-31).  (The
    kernel sums M_u in f32; one more 2^-12 of margin here keeps this bound on the safe side of a power-of-two boundary.)"""
    S32 = abs(sp.csr_matrix(S).astype(np.float32)).astype(np.float64)
    rowmax = np.asarray(S32.max(axis=1).todense()).ravel() if S32.nnz else np.zeros(S32.shape[0])
    H = sp.csr_matrix(hist)
    Hb = sp.csr_matrix((np.ones(H.nnz), H.indices, H.indptr), shape=H.shape)
    M = Hb @ rowmax
    d = np.diff(H.indptr).astype(np.float64)
    e = np.frexp(M * (1.0 + 2.0 ** -11))[1]  # M (1 + 2^-11) < 2^e
    return np.where(M > 0, d * np.ldexp(1.0, e - 31), 0.0)
