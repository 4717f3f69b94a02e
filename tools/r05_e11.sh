#!/bin/bash
# round 5, experiment 11: K9 first pass with one task per user - parity and time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e11.log
(python -m pytest tests/test_gpu_ease.py -x -q -m gpu -k "two_pass or prune or predict or pipeline or sharded_multi_rank" 2>&1 | tail -4
 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "ease_c3_full_size and not grid" 2>&1 | tail -4
 python tools/k9_route_bench.py C4 2>&1 | tail -12
) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-300
