#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(RP_K9_TILE_MAJOR=1 python3 tools/k9_two_pass_bench.py C4 | tail -3; python3 tools/k9_ablate.py recpack_amd/csrc/librecpack_hip.so C4;  RP_K9_TILE_MAJOR=1 python3 tools/k9_ablate.py recpack_amd/csrc/librecpack_hip.so C3; python3 tools/k9_ablate.py recpack_amd/csrc/librecpack_hip.so C3) > gpurun_out/r04f_k9.log 2>&1
grep -v amdgpu.ids gpurun_out/r04f_k9.log | tail -10
