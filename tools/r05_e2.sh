#!/bin/bash
# round 5, experiment 2b: s_setprio variants of k_gemm_hp16w on the C4 and C3 solves, interleaved on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e2d.log
(for v in "" _x32 _x64 _x160 _x288 _x544 _x36 _x32; do
   export RECPACK_HIP_LIB=$GRAFT_REPO_ROOT/recpack_amd/csrc/librecpack_hip$v.so
   echo "== variant '$v'"
   RECPACK_GEMM_FLUSH_K=4096 RECPACK_BENCH_ITERS=3 python tools/ease_bench.py 1000000 100000 0.0005 2>&1 | grep fit | tail -2
   python tools/gemm_k_sweep.py 2>&1 | grep TFLOP | tail -3
 done) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-250 | tail -60
