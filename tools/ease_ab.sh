#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(echo "== default (n22 on fp32 kernel)"; python tools/ease_bench.py 1000000 100000 0.0005; echo "== n22 on planes (round 3)"; RP_N22_PLANES=1 python tools/ease_bench.py 1000000 100000 0.0005; echo "== default again"; python tools/ease_bench.py 1000000 100000 0.0005; echo "== C3"; python tools/ease_bench.py 200000 50000 0.001;  echo "== C3 planes"; RP_N22_PLANES=1 python tools/ease_bench.py 200000 50000 0.001) > gpurun_out/r04e_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r04e_bench.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "ease_c3_full_size and not grid" > gpurun_out/r04e_full.log 2>&1; echo "full rc=$?" >> gpurun_out/r04e_full.log; tail -2 gpurun_out/r04e_full.log
