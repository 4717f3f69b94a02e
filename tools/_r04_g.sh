#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RP_INV_TRACE=1 python tools/ease_bench.py 1000000 100000 0.0005 > gpurun_out/r04g_trace.log 2>&1
grep -v amdgpu.ids gpurun_out/r04g_trace.log | tail -70
