#!/bin/bash
# round 5, experiment 3: XCD-aware tile order of k_gemm_hp16w (gemm_xcd_block) on the C4 solve and the square product, one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e3.log
(for v in 0 4 2 8 16 0 4; do
   echo "== xcd_block $v"
   RECPACK_XCD_BLOCK=$v RECPACK_GEMM_FLUSH_K=4096 RECPACK_BENCH_ITERS=3 python tools/ease_bench.py 1000000 100000 0.0005 2>&1 | grep fit | tail -2
   RECPACK_XCD_BLOCK=$v python tools/gemm_k_sweep.py 2>&1 | grep TFLOP | tail -3
 done) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-250 | tail -60
