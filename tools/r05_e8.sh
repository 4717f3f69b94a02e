#!/bin/bash
# round 5, experiment 8: N22 chains during the factorisation + diagonal tiles of N N^T inside the wide product
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e8.log
(RECPACK_BENCH_ITERS=3 python tools/ease_bench.py 1000000 100000 0.0005 2>&1 | grep fit
 RECPACK_BENCH_ITERS=3 python tools/ease_bench.py 200000 50000 0.001 2>&1 | grep fit
 python -m pytest tests/test_gpu_ease.py -x -q -m gpu 2>&1 | tail -3
 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "ease_c3_full_size and not grid" 2>&1 | tail -3
) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-250
