#!/bin/bash
# round 5, experiment 15b: XCD block shape by product kind (default 4 = 16 x 2 blocks for the trailing updates, 4 x 8 elsewhere) against uniform shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e15b.log
(B="python tools/ease_bench.py 1000000 100000 0.0005"
 export RECPACK_BENCH_ITERS=3
 for v in 4 8 2 4 8; do echo "== xcd $v"; RECPACK_XCD_BLOCK=$v $B | tail -1; done
 echo "== C3"; python tools/ease_bench.py 200000 50000 0.001 | tail -1
) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-220
