#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ease.py -x -q -m gpu -k "corner_cases or two_threads" 2>&1 | grep -v amdgpu.ids | tail -25
