"""A context option of the EASE solve across its values (default: the inverse's plane cache, gemm_plane_cache): B of a fit with
each value against B with the first, and the fit time.
    python tools/plane_cache_check.py [users items density [value ...]]      (gemm_plane_cache: 0 off, 1 auto, > 1 a byte limit)
    RP_OPT=chol_sub_panel python tools/plane_cache_check.py 100000 20000 0.002 0 8 4
With `certify <config> <l2>` as the arguments: the certified column error (tests/ease_cg.py) with the cache in its default mode."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import powerlaw_csr  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "certify":
    sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
    import ease_cg
    ctx = _lib.Context(0)
    for cfg, l2 in zip(sys.argv[2::2], sys.argv[3::2]):
        r = ease_cg.run_parity(_lib, ctx, cfg, l2=float(l2), n_cols=(8, 4, 4), n_users=32, log=lambda *_: None)
        print(f"plane cache on, {cfg} l2={l2}: err_col_max {r['err_col_max']:.2e} median {r['err_col_median']:.2e} scores {r['err_score_max']:.2e} "
              f"bound {r['certified_bound_rel']:.1e} fit {r['fit_s']} s", flush=True)
        ctx.trim()
    sys.exit(0)

U, I, dens = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (40000, 9000, 0.004)
modes = [int(a) for a in sys.argv[4:]] or [0, 1, 4 * 4096 * 4096 + (64 << 20)]
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
ip, ix = powerlaw_csr(U, I, dens, seed=7, device=dev)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
ref = None
ok = True
OPT = os.environ.get("RP_OPT", "gemm_plane_cache")
for mode in modes:
    ctx.set_option(OPT, mode)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = _lib.ease_fit(ctx, dX, 50.0, 0.0)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        tm = ctx.timings()
        ctx.reset_timings()
    ptr, _, ld = m.dense_device_ptr()
    from recpack_amd import dist as rdist
    B = torch.as_tensor(rdist._DevPtr(ptr, (I, ld), "<f4", m), device=dev)[:, :I].clone()
    if ref is None:
        ref = B
        err = 0.0
    else:
        err = float((B - ref).abs().max() / ref.abs().max())
        ok = ok and err <= 2e-5 and bool(torch.isfinite(B).all())
    print(f"I={I} {OPT}={mode}: fit {best * 1e3:.1f} ms (cholesky {tm['cholesky']['ms']:.1f} inverse {tm['inverse']['ms']:.1f}), "
          f"max|dB|/max|B| against the first mode {err:.2e}", flush=True)
    del m
sys.exit(0 if ok else 1)
