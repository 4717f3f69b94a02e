#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_itemknn.py -x -q -m gpu -k "any_k or synthetic or golden or fuzz" 2>&1 | grep -v amdgpu.ids | tail -15
