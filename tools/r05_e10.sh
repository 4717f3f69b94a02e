#!/bin/bash
# round 5, experiment 10: when the N22 chains are issued (inv_chain_policy 0 / 1 / 2), interleaved on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e10.log
(for v in 0 1 2 0 1 2; do echo "== policy $v"
 RECPACK_N22_POLICY=$v RECPACK_BENCH_ITERS=3 python tools/ease_bench.py 1000000 100000 0.0005 2>&1 | grep fit | tail -2
 RECPACK_N22_POLICY=$v RECPACK_BENCH_ITERS=3 python tools/ease_bench.py 200000 50000 0.001 2>&1 | grep fit | tail -1
 done) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-250
