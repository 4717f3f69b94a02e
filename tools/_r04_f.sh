#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for t in 7168 8192 9216 10240 11200; do echo "== tile $t"; RP_HALF_TILE=$t python3 tools/k9_two_pass_bench.py C4 | grep "int8 rows run 1\|int8 rows against" | tail -2; done) > gpurun_out/r04f_k9.log 2>&1
grep -v amdgpu.ids gpurun_out/r04f_k9.log | tail -20
