"""Certified column error of the EASE fit for a flush period of the compensated sums: python tools/ease_flush_check.py <kc> [config l2]..."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ease_cg

from recpack_amd import _lib

kc = int(sys.argv[1])
ctx = _lib.Context(0)
ctx.set_option("gemm_flush_k", kc)
args = sys.argv[2:] or ["C3", "200"]
for cfg, l2 in zip(args[0::2], args[1::2]):
    r = ease_cg.run_parity(_lib, ctx, cfg, l2=float(l2), n_cols=(8, 4, 4), n_users=32, log=lambda *_: None)
    print(f"flush_k {kc} {cfg} l2={l2}: err_col_max {r['err_col_max']:.2e} median {r['err_col_median']:.2e} scores {r['err_score_max']:.2e} "
          f"bound {r['certified_bound_rel']:.1e} fit {r['fit_s']} s", flush=True)
    ctx.trim()
