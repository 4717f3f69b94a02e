"""Parity rules shared by the CPU and GPU tests."""
import numpy as np
import scipy.sparse as sp


def rows_of(M):
    M = sp.csr_matrix(M)
    return [(M.indices[a:b], M.data[a:b]) for a, b in zip(M.indptr[:-1], M.indptr[1:])]


def tie_aware_topk(cand: sp.csr_matrix, ref_sel: sp.csr_matrix, test_idx, test_nnz, eps=1e-9, positive_only=True):
    """cand: all candidate scores per row (reference values); ref_sel: the reference's selection;
    test_idx[r,:test_nnz[r]]: columns chosen by the implementation under test.

    With t = the reference's smallest selected value in the row:
      MUST = {v > t + eps|t|}, MAY = {|v-t| <= eps|t|}; require MUST <= Sel <= MUST u MAY, |Sel| = K'.
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
        must = set(cols[vals > t + eps * abs(t)].tolist())
        may = set(cols[np.abs(vals - t) <= eps * abs(t)].tolist())
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
