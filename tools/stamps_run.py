import sys, os
sys.path.insert(0, os.getcwd())
import torch
from recpack_amd import _lib
_lib.LIB_PATH = os.path.join(os.getcwd(), "recpack_amd/csrc/librecpack_hip_stamps.so")
from recpack_amd.synth import config_csr, CONFIGS
cfg = sys.argv[1]
dev = torch.device("cuda", 0)
ip, ix = config_csr(cfg, device=dev)
U, I, _ = CONFIGS[cfg]
ctx = _lib.Context(0)
for it in range(2):
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    m = _lib.itemknn_fit(ctx, dX, 200)
    ctx.sync()
print(ctx.timings()["gram_topk"])
if len(sys.argv) > 2:
    n = int(sys.argv[2])
    oi = torch.empty((n, 100), dtype=torch.int32, device=dev); ov = torch.empty((n, 100), dtype=torch.float32, device=dev); on = torch.empty((n,), dtype=torch.int32, device=dev)
    ctx.reset_timings()
    _lib.predict_topk_dev(ctx, m, dX, 100, oi, ov, on, user_begin=0, user_end=n)
    ctx.sync()
    print("predict", ctx.timings()["predict"])
