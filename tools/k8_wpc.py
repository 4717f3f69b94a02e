"""K8 at fewer waves per CU (diagnostic build: make -C recpack_amd/csrc diag; RECPACK_HIP_LIB=.../librecpack_hip_diag.so RP_PW_WPC=4 python tools/k8_wpc.py)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from recpack_amd import _lib
from recpack_amd.synth import config_csr, CONFIGS
dev = torch.device("cuda", 0)
ip, ix = config_csr("C4", device=dev)
U, I, _ = CONFIGS["C4"]
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
m = _lib.itemknn_fit(ctx, dX, 200)
oi = torch.empty((U, 100), dtype=torch.int32, device=dev); ov = torch.empty((U, 100), dtype=torch.float32, device=dev); on = torch.empty((U,), dtype=torch.int32, device=dev)
for it in range(3):
    ctx.reset_timings()
    _lib.predict_topk_dev(ctx, m, dX, 100, oi, ov, on)
    ctx.sync()
    t = ctx.timings()["predict"]["ms"]
print("RP_PW_WPC", os.environ.get("RP_PW_WPC"), "K8 ms", round(t, 2), flush=True)
