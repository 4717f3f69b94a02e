"""K8's time by history length: the same model, prediction matrices that keep only the users with at most 64 items / only the others."""
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
lens = (ip[1:] - ip[:-1]).long()
oi = torch.empty((U, 100), dtype=torch.int32, device=dev); ov = torch.empty((U, 100), dtype=torch.float32, device=dev); on = torch.empty((U,), dtype=torch.int32, device=dev)
def sub(mask):
    keep = torch.repeat_interleave(mask, lens)
    l2 = torch.where(mask, lens, torch.zeros_like(lens))
    p2 = torch.zeros(U + 1, dtype=torch.int64, device=dev); p2[1:] = torch.cumsum(l2, 0)
    return p2.to(torch.int32), ix[keep].contiguous()
for name, mask in (("all", lens >= 0), ("<= 64 items", lens <= 64), ("> 64 items", lens > 64), ("<= 32 items", lens <= 32)):
    p2, i2 = sub(mask)
    d2 = _lib.DeviceCSR.from_device(ctx, p2, i2, (U, I))
    best = 1e9
    for it in range(3):
        ctx.reset_timings()
        _lib.predict_topk_dev(ctx, m, d2, 100, oi, ov, on)
        ctx.sync()
        best = min(best, ctx.timings()["predict"]["ms"])
    print(f"{name}: {int(mask.sum())} users, {int(i2.numel())} interactions: K8 {best:.2f} ms", flush=True)
