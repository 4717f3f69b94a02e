#!/bin/bash
# round 5, experiment 7: kernel trace of the C4 solve -> where no wide product runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/kt
RECPACK_BENCH_ITERS=2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/ease_bench.py 1000000 100000 0.0005 > gpurun_out/r05_e7.log 2>&1
grep fit gpurun_out/r05_e7.log
python tools/solve_timeline.py gpurun_out/kt > gpurun_out/r05_solve_timeline.txt 2>&1; cat gpurun_out/r05_solve_timeline.txt
python tools/chol_tail_trace.py gpurun_out/kt 700 2 > gpurun_out/r05_chol_chain_trace.txt 2>&1; head -30 gpurun_out/r05_chol_chain_trace.txt
rm -rf gpurun_out/kt
