#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ease.py -x -q -m gpu -k "two_pass" > gpurun_out/r04f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04f_tests.log; tail -5 gpurun_out/r04f_tests.log | cut -c1-250
