"""Valued (time-aware) fit and weighted predict at the bench configurations: C2 and C4 matrices with seeded timestamps.
usage: python tools/tars_bench.py [C2 C4]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from recpack_amd import _lib

if os.environ.get("RECPACK_STAMPS"):  # diagnostic build (make stamps): per-phase cycle counters on stderr
    _lib.LIB_PATH = os.path.join(os.getcwd(), "recpack_amd/csrc/librecpack_hip_stamps.so")
from recpack_amd.synth import CONFIGS, config_csr

for cfg in (sys.argv[1:] or ["C2"]):
    dev = torch.device("cuda", 0)
    ip, ix = config_csr(cfg, device=dev)
    U, I, _ = CONFIGS[cfg]
    ctx = _lib.Context(0)
    indptr, indices = ip.cpu().numpy(), ix.cpu().numpy()
    nnz = indices.shape[0]
    rng = np.random.default_rng(7)
    ts = rng.integers(1, 90 * 24, size=nnz).astype(np.float64)  # hours over 90 days
    w = np.exp(-0.002 * (ts.max() + 1 - ts))
    n2 = np.bincount(indices, weights=w * w, minlength=I)
    wn = w / np.sqrt(np.where(n2 > 0, n2, 1.0))[indices]
    dX = _lib.DeviceCSR.from_host(ctx, indptr, indices, (U, I))

    def timed(label, fn):
        for it in range(2):
            ctx.reset_timings()
            t0 = time.perf_counter()
            out = fn()
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3
        print(cfg, label, round(ms, 1), "ms", {k: round(v["ms"], 2) for k, v in ctx.timings().items() if v["launches"]}, flush=True)
        return out

    dX.set_values(wn)
    m = timed("valued cosine fit K=200", lambda: _lib.itemknn_fit_values(ctx, dX, 200))
    dX.set_values(w)
    timed("weighted predict top-100", lambda: _lib.predict_topk_dev_alloc(ctx, m, dX, 100) if hasattr(_lib, "predict_topk_dev_alloc") else _lib.predict_topk(ctx, m, dX, 100, False, user_begin=0, user_end=min(U, 200000)))
    dX.set_values(ts)
    timed("time-distance fit (exponential) K=200", lambda: _lib.itemknn_fit_timedist(ctx, dX, 200, "exponential", 0.002, float(ts.max() - ts.min())))
    timed("hermann fit K=200", lambda: _lib.itemknn_fit_hermann(ctx, dX, 200, float(ts.max() + 1)))
    dX.set_values(None)
    timed("binary predict top-100 (same users)", lambda: _lib.predict_topk(ctx, m, dX, 100, False, user_begin=0, user_end=min(U, 200000)))
