import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from recpack_amd import _lib
from recpack_amd.synth import config_csr, CONFIGS
for cfg in ("C2", "C4"):
    dev = torch.device("cuda", 0)
    ip, ix = config_csr(cfg, device=dev)
    U, I, _ = CONFIGS[cfg]
    ctx = _lib.Context(0)
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    for it in range(2):
        ctx.reset_timings(); t0 = time.perf_counter()
        m = _lib.itemknn_fit(ctx, dX, 200, normalize_X=True); ctx.sync()
        print(cfg, "normalize_X fit", round((time.perf_counter() - t0) * 1e3, 1), "ms", {k: round(v["ms"], 2) for k, v in ctx.timings().items() if v["launches"]})
