#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ease.py -x -q -m gpu -s -k "two_pass or float64_inverse" > gpurun_out/r04f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04f_tests.log; grep -i "EASE vs\|passed\|failed\|Error\|rc=" gpurun_out/r04f_tests.log | cut -c1-250 | tail -12
