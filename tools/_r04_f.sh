#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for kc in 192 256 320; do echo "== Kc $kc"; RP_K9_I8_KC=$kc python3 tools/k9_ablate.py recpack_amd/csrc/librecpack_hip.so C4; done; RP_K9_I8_KC=256 python3 tools/k9_ablate.py recpack_amd/csrc/librecpack_hip.so C3; python3 tools/k9_ablate.py recpack_amd/csrc/librecpack_hip.so C3) > gpurun_out/r04f_k9.log 2>&1
grep -v amdgpu.ids gpurun_out/r04f_k9.log | tail -10
