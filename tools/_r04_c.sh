#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/dist_rank_emulate.py --world 8 --ranks 0,7 --config C4 --out gpurun_out/r04_rank_emulation.json > gpurun_out/r04d_emul.log 2>&1; echo "rc=$?" >> gpurun_out/r04d_emul.log
grep -v amdgpu.ids gpurun_out/r04d_emul.log | grep -v "^  rank" | tail -4 | cut -c1-2600
