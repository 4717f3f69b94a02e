"""The reference's own ``Pipeline`` drives the drop-in classes (build container only: skipped where /root/reference is absent).
See tests/pipeline_dropin_check.py - it needs a fresh interpreter (recpack on sys.path before recpack_amd is imported)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/recpack"), reason="the reference is only present in the build container")
def test_reference_pipeline_runs_the_dropin_classes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pipeline_dropin_check.py")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "PIPELINE DROP-IN OK" in r.stdout
    assert r.stdout.count("ok: Pipeline.run()") == 2
