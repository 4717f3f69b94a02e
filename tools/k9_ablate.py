"""Time K9 (EASE predict, two-pass) with one library build: python tools/k9_ablate.py <lib.so> [config]
(make -C recpack_amd/csrc ablate ABL=1024 builds librecpack_hip_abl1024.so: the tile kernel without its selection - the
results are wrong by construction, the time left is streaming + clearing)."""
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
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.ease_fit(ctx, dX, 200.0, 0.0)
oi = torch.empty((U, 100), dtype=torch.int32, device=dev)
ov = torch.empty((U, 100), dtype=torch.float32, device=dev)
on = torch.empty((U,), dtype=torch.int32, device=dev)
best = 1e9
for it in range(3):
    ctx.reset_timings()
    _lib.predict_topk_dev(ctx, m, dX, 100, oi, ov, on, exclude_history=True)
    ctx.sync()
    best = min(best, ctx.timings()["predict_dense"]["ms"])
print(f"{os.path.basename(sys.argv[1])} {cfg}: EASE predict {best:.1f} ms", flush=True)
