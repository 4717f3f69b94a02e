"""Time the wave predict kernel of one library build: python tools/k8_ablate.py <lib.so> [config] [users]
(make -C recpack_amd/csrc ablate builds librecpack_hip_abl{1,2,4,8,3}.so: the kernel without its sum atomics / walk / dirty-map
atomics / entry loads / sum atomics and walk - a subtractive attribution of its time)."""
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
n = int(sys.argv[3]) if len(sys.argv) > 3 else U
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.itemknn_fit(ctx, dX, 200)
oi = torch.empty((n, 100), dtype=torch.int32, device=dev)
ov = torch.empty((n, 100), dtype=torch.float32, device=dev)
on = torch.empty((n,), dtype=torch.int32, device=dev)
best = 1e9
for it in range(3):
    ctx.reset_timings()
    _lib.predict_topk_dev(ctx, m, dX, 100, oi, ov, on, user_begin=0, user_end=n, exclude_history=True)
    ctx.sync()
    best = min(best, ctx.timings()["predict"]["ms"])
print(f"{os.path.basename(sys.argv[1])}: predict of {n} users {best:.2f} ms", flush=True)
