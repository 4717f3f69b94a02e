"""Time the dense predict (K9) alone on a synthetic EASE-shaped model: python tools/k9_bench.py <users> <items> <density> [Kp]
B is random (the kernel's cost does not depend on its values); rows are padded to 128 floats as rphip_ease_fit's are."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from recpack_amd import _lib  # noqa: E402
from recpack_amd.synth import powerlaw_csr  # noqa: E402

U, I, dens = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
Kp = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda", 0)
ip, ix = powerlaw_csr(U, I, dens, seed=20260003, device=dev)
import recpack_amd._lib as _l
if len(sys.argv) > 5:
    _l.LIB_PATH = os.path.abspath(sys.argv[5])
ctx = _lib.Context(0)
dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
ld = (I + 127) // 128 * 128
torch.manual_seed(7)
B = torch.empty((I, ld), dtype=torch.float32, device=dev)
for r0 in range(0, I, 8192):
    B[r0:r0 + 8192].normal_()
m = _lib.Model.from_dense_cols(ctx, I, 0, I, B)
out_idx = torch.empty((U, Kp), dtype=torch.int32, device=dev)
out_val = torch.empty((U, Kp), dtype=torch.float32, device=dev)
out_nnz = torch.empty((U,), dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.predict_topk_dev(ctx, m, dX, Kp, out_idx, out_val, out_nnz)
    ctx.sync()
    dt = time.perf_counter() - t0
    nnz = int(ix.numel())
    print(f"U={U} I={I} nnz={nnz} Kp={Kp}: predict {dt * 1e3:.1f} ms = {4.0 * I * nnz / dt / 1e12:.2f} TB/s by the no-reuse model", flush=True)
print("checksum", int(out_idx.long().sum().item()), float(out_val.double().sum().item()))
