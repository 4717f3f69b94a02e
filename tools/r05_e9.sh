#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ease.py -x -q -m gpu -k "sharded_multi_rank" > gpurun_out/r05_e9.log 2>&1
grep -v amdgpu.ids gpurun_out/r05_e9.log | tail -60 | cut -c1-300
