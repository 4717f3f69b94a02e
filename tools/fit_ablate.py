"""Time the ItemKNN fit of one library build: python tools/fit_ablate.py <lib.so> [config]
(make -C recpack_amd/csrc fitablate builds librecpack_hip_fabl{1,2,4,3,6}.so: the fit kernels without their counter atomics /
selection / index loads and combinations - a subtractive attribution of their time)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from recpack_amd.synth import CONFIGS, config_csr  # noqa: E402

cfg = sys.argv[2] if len(sys.argv) > 2 else "C4"
dev = torch.device("cuda", 0)
ip, ix = config_csr(cfg, device=dev)
U, I, _ = CONFIGS[cfg]
ctx = _lib.Context(0)
best = None
for it in range(3):
    ctx.reset_timings()
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    m = _lib.itemknn_fit(ctx, dX, 200)
    ctx.sync()
    tm = ctx.timings()
    cur = {k: round(tm[k]["ms"], 2) for k in ("gram_topk", "gram_topk_wave", "topk_merge", "gram_finalize", "csr_transpose", "fit_plan")}
    if best is None or sum(cur.values()) < sum(best.values()):
        best = cur
    del m
print(f"{os.path.basename(sys.argv[1])}: heavy {best['gram_topk']} ms, light {best['gram_topk_wave']} ms, merge {best['topk_merge']} ms, finalize {best['gram_finalize']} ms"
      f" | K0 {best['csr_transpose']} ms, plan {best['fit_plan']} ms | K1-K3 sum {best['gram_topk'] + best['gram_topk_wave'] + best['topk_merge'] + best['gram_finalize']:.2f} ms", flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "gram":  # K4: the dense Gram of EASE (the same accumulate loop, u32 counters)
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    for it in range(2):
        ctx.reset_timings()
        g = _lib.ease_gram_create(ctx, dX)
        ctx.sync()
        print(f"  dense Gram (K4): {ctx.timings()['gram_dense']['ms']:.2f} ms", flush=True)
        del g
