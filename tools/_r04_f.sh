#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python3 tools/k9_two_pass_bench.py C4 | tail -5; python3 tools/k9_two_pass_bench.py C3 | tail -4) > gpurun_out/r04f_k9.log 2>&1
grep -v amdgpu.ids gpurun_out/r04f_k9.log | tail -12
python -m pytest tests/test_gpu_ease.py -x -q -m gpu -k "predict or topk or two or slab or filter" > gpurun_out/r04f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04f_tests.log; tail -3 gpurun_out/r04f_tests.log
