"""Build librecpack_hip.so (hipcc, gfx950). Used by __graft_entry__.build() and, on a fresh checkout, by the loader."""
import fcntl
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recpack_amd", "csrc")
LIB = os.path.join(CSRC, "librecpack_hip.so")


def build_hip(jobs: int = 8, quiet: bool = False, only_if_missing: bool = False) -> str:
    """One build at a time per tree: every caller takes an exclusive flock on csrc/.build.lock, so concurrent ranks
    cannot clobber each other's objects; whoever comes second finds the library up to date (make) or present."""
    if not shutil.which("make") or not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        if only_if_missing:
            return LIB
        raise RuntimeError("make / hipcc not found")
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if only_if_missing and os.path.exists(LIB):
                return LIB
            r = subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], stdout=subprocess.PIPE if quiet else None,
                               stderr=subprocess.STDOUT if quiet else None, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"make -C {CSRC} failed (rc {r.returncode})" + (":\n" + r.stdout[-4000:] if quiet and r.stdout else ""))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if not os.path.exists(LIB):
        raise RuntimeError("librecpack_hip.so was not produced")
    return LIB
