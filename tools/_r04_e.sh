#!/bin/bash
# round-4 iteration script: EASE parity + solve timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ease.py -x -q -m gpu -k "not sharded" > gpurun_out/r04e_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04e_tests.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "ease_c3_full_size and not grid" > gpurun_out/r04e_full.log 2>&1; echo "full rc=$?" >> gpurun_out/r04e_full.log
(echo "== default"; python tools/ease_bench.py 1000000 100000 0.0005; echo "== measured scales (round 3)"; RP_CHOL_MEASURE_SCALES=1 python tools/ease_bench.py 1000000 100000 0.0005; echo "== flush 2048"; RECPACK_GEMM_FLUSH_K=2048 python tools/ease_bench.py 1000000 100000 0.0005; echo "== C3"; python tools/ease_bench.py 200000 50000 0.001) > gpurun_out/r04e_bench.log 2>&1
tail -3 gpurun_out/r04e_tests.log; tail -3 gpurun_out/r04e_full.log; grep -v amdgpu.ids gpurun_out/r04e_bench.log
