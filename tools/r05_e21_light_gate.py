"""VERDICT r4 item 2(a), the gate: the light-row SpGEMM instance on a ONE-tile matrix whose light rows look like C4's
(users of lognormal length 50, items Zipf, rows of <= 255 users carrying ~16 K products each) - the time per product there is
what a whole-row accumulator could approach at C4 (it is optimistic: a byte tile of 25.6 KB holds the whole row, four
workgroups per CU; a hash table for a C4 row's ~10 K distinct columns is larger).  Prints ps per product for both."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import CONFIGS, config_csr, powerlaw_csr  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0)


def run(tag, ip, ix, U, I):
    best = None
    for it in range(4):
        ctx.reset_timings()
        dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
        m = _lib.itemknn_fit(ctx, dX, 200)
        ctx.sync()
        tm = ctx.timings()
        cur = (tm["gram_topk_wave"]["ms"], tm["gram_topk_wave"]["work"], tm["gram_topk"]["ms"], tm["topk_merge"]["ms"], tm["gram_finalize"]["ms"])
        if best is None or cur[0] < best[0]:
            best = cur
        del m
    deg = torch.bincount(ix.long(), minlength=I)
    light = (deg <= 255) & (deg > 0)
    n_light = int(light.sum())
    lens = (ip[1:] - ip[:-1]).long()
    ent_len = torch.repeat_interleave(lens, lens)          # every entry carries its user's history length
    row_cost = torch.zeros(I, dtype=torch.int64, device=ix.device).scatter_add_(0, ix.long(), ent_len)
    lp = float(row_cost[light].sum())                      # products of the light rows (all tiles)
    P = float((lens * lens).sum())
    print(f"   P = {P:.3e}, light share {lp / P:.3f}, median light degree {int(deg[light].median())}", flush=True)
    print(f"{tag}: {U} x {I}, nnz {ix.numel()}, light rows {n_light}; light kernel {best[0]:.2f} ms for {lp:.3e} products = "
          f"{best[0] * 1e9 / max(lp, 1):.2f} ps/product, {lp / max(n_light, 1):.0f} products/row; heavy {best[2]:.2f} merge {best[3]:.2f} finalize {best[4]:.2f} ms",
          flush=True)
    return best[0] * 1e9 / max(lp, 1)


U, I = 60000, 25600
for U in (60000, 180000):
    ip, ix = powerlaw_csr(U, I, 50.0 / I, seed=5, device=dev)
    run(f"one tile U={U}", ip, ix, U, I)
ip, ix = config_csr("C4", device=dev)
U4, I4, _ = CONFIGS["C4"]
run("C4", ip, ix, U4, I4)
