"""Round 6 gate for the SpGEMM's "dense head" (VERDICT r5 item 1): what the fit kernels would take at C4 if the H most popular
items' entries left the users' segments (their products coming from a dense H x H block on the matrix cores and, by the Gram's
symmetry, from the head rows' own tiles).  Items are relabelled by popularity here (a gate, not the product path), then:
  full          the fit as it is (relabelled matrix: the same kernels, the same products)
  head rows     fit_rows = [0, H): what the head rows cost today
  tail rows     fit_rows = [H, I)
  tail only     the matrix without the head columns: tail rows on tail-only segments - the light / heavy kernels' time with
                0.38 (H = 1024) of the products per segment
Prints the heavy / light / merge / finalize kernel times of each."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import CONFIGS, config_csr  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
lens = (ip[1:] - ip[:-1]).long()
rows = torch.repeat_interleave(torch.arange(U, device=dev), lens)
deg = torch.bincount(ix.long(), minlength=I)
order = torch.argsort(-deg, stable=True)
P = float((lens * lens).sum())


def relabel(H):
    """H = 0: ids = popularity rank (every tile sorted by popularity: tile 0 holds 85 % of the entries - the heavy kernel's atomics then
    meet in one tile and it slows down by 40 %, a confound).  H > 0: only the H most popular items move (to ids [0, H), by popularity);
    the others keep their original relative order behind them, so the tiles stay the random mix they are in the product path."""
    rank = torch.empty_like(order)
    if H == 0:
        rank[order] = torch.arange(I, device=dev)
    else:
        is_head = torch.zeros(I, dtype=torch.bool, device=dev)
        is_head[order[:H]] = True
        rank[order[:H]] = torch.arange(H, device=dev)
        tail_ids = torch.nonzero(~is_head).flatten()  # ascending original ids
        rank[tail_ids] = H + torch.arange(I - H, device=dev)
    ixr_ = rank[ix.long()]
    key = rows * I + ixr_
    key, _ = torch.sort(key)
    return (key % I).to(torch.int32)



def run(tag, ip_, ix_, fit_rows=None):
    best = None
    for it in range(3):
        ctx.reset_timings()
        dX = _lib.DeviceCSR.from_device(ctx, ip_, ix_, (U, I), fit_rows=fit_rows)
        m = _lib.itemknn_fit(ctx, dX, 200) if fit_rows is None else _lib.itemknn_fit(ctx, dX, 200, fit_rows[0], fit_rows[1])
        ctx.sync()
        tm = ctx.timings()
        cur = [tm[k]["ms"] for k in ("gram_topk", "gram_topk_wave", "topk_merge", "gram_finalize")]
        if best is None or sum(cur) < sum(best):
            best = cur
        del m, dX
    l2 = (ip_[1:] - ip_[:-1]).long()
    print(f"{tag:34s} nnz {ix_.numel():9d}  P {float((l2 * l2).sum()) / P:5.3f} of C4's  heavy {best[0]:6.2f}  light {best[1]:6.2f}  merge {best[2]:5.2f}  "
          f"finalize {best[3]:5.2f}  sum {sum(best):6.2f} ms", flush=True)


run("full (original labels)", ip, ix)
run("full (popularity labels)", ip, relabel(0))
for H in (256, 1024, 2048, 4096, 8192):
    ixr = relabel(H)
    run(f"  full (head of {H} moved to the front)", ip, ixr)
    hl = torch.zeros(U, dtype=torch.int64, device=dev).scatter_add_(0, rows, (ixr < H).long())
    print(f"H = {H}: head entries {float(hl.sum()) / ix.numel():.3f} of nnz; head x head products {float((hl * hl).sum()) / P:.3f} P, head x tail "
          f"{float((hl * (lens - hl)).sum()) / P:.3f} P (each way), tail x tail {float(((lens - hl) ** 2).sum()) / P:.3f} P", flush=True)
    run(f"  head rows [0, {H})", ip, ixr, (0, H))
    run(f"  tail rows [{H}, I)", ip, ixr, (H, I))
    keep = ixr >= H
    ip_t = torch.zeros(U + 1, dtype=torch.int32, device=dev)
    ip_t[1:] = torch.cumsum((lens - hl), 0).to(torch.int32)
    run("  tail-only matrix, all rows", ip_t, ixr[keep].contiguous())
