"""K9 on a fitted EASE model: the one-pass kernel against the two-pass routes (options predict_half, predict_int8), times and equality.
python tools/k9_two_pass_bench.py [config [Kp]]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from recpack_amd import _lib
from recpack_amd.synth import CONFIGS, config_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
Kp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
U, I, _ = CONFIGS[cfg]
ip, ix = config_csr(cfg, device=dev)
ctx = _lib.Context(0)
if os.environ.get("RP_HALF_TILE"):
    ctx.set_option("predict_half_tile", int(os.environ["RP_HALF_TILE"]))
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
ctx.sync()
outs = {}
for name, half, i8 in (("one pass", 0, 0), ("fp16 rows", 1, 0), ("int8 rows", 1, 1), ("int8 rows", 1, 1)):
    ctx.set_option("predict_half", half)
    ctx.set_option("predict_int8", i8)
    oi = torch.empty((U, Kp), dtype=torch.int32, device=dev)
    ov = torch.empty((U, Kp), dtype=torch.float32, device=dev)
    on = torch.empty((U,), dtype=torch.int32, device=dev)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.predict_topk_dev(ctx, m, dX, Kp, oi, ov, on, exclude_history=True)
        ctx.sync()
        dt = time.perf_counter() - t0
        print(f"{cfg} Kp={Kp} {name} run {it}: {dt * 1e3:.1f} ms = {4.0 * I * int(ix.numel()) / dt / 1e12:.2f} TB/s by the 4*I*nnz model", flush=True)
    outs[name] = (oi, ov, on)
a = outs["one pass"]
for name in ("fp16 rows", "int8 rows"):
    b = outs[name]
    print(name, "against one pass: identical idx:", bool((a[0] == b[0]).all()), "identical values:", bool((a[1].view(torch.int32) == b[1].view(torch.int32)).all()),
          "identical nnz:", bool((a[2] == b[2]).all()))
