#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ease.py -x -q -m gpu -k "sharded" > gpurun_out/r04c_ease.log 2>&1; echo "ease rc=$?" >> gpurun_out/r04c_ease.log
tail -4 gpurun_out/r04c_ease.log
timeout 900 python tools/dist_rank_emulate.py --world 8 --ranks 0,3,7 --config C4 --out gpurun_out/r04_rank_emulation.json > gpurun_out/r04d_emul.log 2>&1; echo "rc=$?" >> gpurun_out/r04d_emul.log
grep -v amdgpu.ids gpurun_out/r04d_emul.log | grep -v "^  rank" | tail -5 | cut -c1-3600
