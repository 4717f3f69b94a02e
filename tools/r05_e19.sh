#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ease.py tests/test_gpu_itemknn.py -x -q -m gpu -k "sharded" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
