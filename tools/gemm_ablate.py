"""Attribution by removal for k_gemm_hp16w (make -C recpack_amd/csrc gemmablate): the 8192^3 product with parts of the kernel removed.
python tools/gemm_ablate.py"""
import glob
import os
import subprocess
import sys

root = os.getcwd()
libs = [("as shipped", os.path.join(root, "recpack_amd/csrc/librecpack_hip.so"))]
names = {1: "no LDS-DMA requests in the loop", 2: "no fragment reads in the loop", 3: "neither requests nor fragment reads", 4: "no barrier, no counted wait",
         8: "no flush", 15: "MFMAs only"}
for f in sorted(glob.glob(os.path.join(root, "recpack_amd/csrc/librecpack_hip_gabl*.so")), key=lambda p: int(p.split("gabl")[1].split(".")[0])):
    a = int(f.split("gabl")[1].split(".")[0])
    libs.append((names.get(a, str(a)), f))
code = """
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from recpack_amd import _lib
_lib.LIB_PATH = sys.argv[1]
ctx = _lib.Context(0)
dev = torch.device('cuda', 0)
n = 8192
A = torch.randn((n, n), dtype=torch.float32, device=dev)
C = torch.empty((n, n), dtype=torch.float32, device=dev)
for it in range(4):
    ctx.reset_timings()
    _lib.gemm_nt(ctx, A, A, C)
    ctx.sync()
tm = ctx.timings()['cholesky']
print(f"{tm['ms']:.3f}")
"""
for name, lib in libs:
    r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
    out = [l for l in r.stdout.splitlines() if l.strip()]
    print(f"{name:40s} {out[-1] if out else r.stderr[-200:]} ms (whole product incl. amax + split)", flush=True)
