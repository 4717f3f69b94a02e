#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r05_gpu_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gpu_suite.log
grep -v amdgpu.ids gpurun_out/r05_gpu_suite.log | tail -30
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r05_smoke.log; tail -3 gpurun_out/r05_smoke.log
