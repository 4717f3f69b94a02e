"""Build librecpack_hip.so (hipcc, gfx950). Used by __graft_entry__.build()."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_hip(jobs: int = 8) -> str:
    csrc = os.path.join(ROOT, "recpack_amd", "csrc")
    subprocess.run(["make", "-C", csrc, f"-j{jobs}"], check=True)
    lib = os.path.join(csrc, "librecpack_hip.so")
    if not os.path.exists(lib):
        raise RuntimeError("librecpack_hip.so was not produced")
    return lib

