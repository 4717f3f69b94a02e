"""Per-step timeline of one workgroup of k_gemm_hp16w (diagnostic build: make -C recpack_amd/csrc gstamps):
stamps per wave and K32 step: 0 loop top, 1 after the counted vmcnt wait, 2 after the barrier, 3 after the early requests (waves 0-3),
4 after the MFMAs.  python tools/gemm_stamps.py [n]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from recpack_amd import _lib

_lib.LIB_PATH = os.path.join(os.getcwd(), "recpack_amd/csrc/librecpack_hip_gstamps.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
A = torch.randn((n, n), dtype=torch.float32, device=dev)
Cm = torch.empty((n, n), dtype=torch.float32, device=dev)
for it in range(2):
    _lib.gemm_nt(ctx, A, A, Cm)
    ctx.sync()
out = np.zeros((8, 48, 5), dtype=np.uint32)
ctx.lib.rphip_debug_gemm_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert ctx.lib.rphip_debug_gemm_stamps(ctx.handle, out.ctypes.data_as(C.c_void_p)) == 0
t = out.astype(np.int64)
base = t[:, 8, 0].min()
print("steps 8..15, cycles since the earliest wave's step-8 top; per wave: top | +vmcnt wait | +barrier | +early requests | +MFMAs")
for w in range(8):
    for sidx in range(8, 14):
        r = t[w, sidx]
        print(f"wave {w} step {sidx}: top {r[0] - base:7d} | wait {r[1] - r[0]:5d} | barrier {r[2] - r[1]:5d} | req {r[3] - r[2]:5d} | mfma {r[4] - r[3]:5d} | to next top {t[w, sidx + 1, 0] - r[4]:5d}")
    print()
d = t[:, 9:40, 0] - t[:, 8:39, 0]
print("mean cycles per step per wave:", d.mean(axis=1).round(0))
for name, a, b in (("vmcnt wait", 0, 1), ("barrier", 1, 2), ("early requests", 2, 3), ("mfma block", 3, 4)):
    print(f"mean {name}: ", (t[:, 8:40, b] - t[:, 8:40, a]).mean(axis=1).round(0))
print("mean tail (late requests, flush, loop):", (t[:, 9:40, 0] - t[:, 8:39, 4]).mean(axis=1).round(0))
