#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(echo "== C4"; python tools/ease_bench.py 1000000 100000 0.0005; echo "== C3"; python tools/ease_bench.py 200000 50000 0.001) > gpurun_out/r04f_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r04f_bench.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "ease_c3_full_size and not grid" -s > gpurun_out/r04f_full.log 2>&1; echo "full rc=$?" >> gpurun_out/r04f_full.log; grep -i "parity\|passed\|failed\|rc=" gpurun_out/r04f_full.log | cut -c1-260 | tail -4
python -m pytest tests/test_gpu_ease.py -x -q -m gpu > gpurun_out/r04f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04f_tests.log; tail -3 gpurun_out/r04f_tests.log
python tools/potrf_check.py > gpurun_out/r04f_potrf.log 2>&1; tail -3 gpurun_out/r04f_potrf.log
