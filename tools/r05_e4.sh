#!/bin/bash
# round 5, experiment 4: EASE parity suite on the new product kernel order + certified error by flush period
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e4.log
(python -m pytest tests/test_gpu_ease.py -x -q -m gpu 2>&1 | tail -5
 for kc in 1024 4096; do python tools/ease_flush_check.py $kc C3 200 C3 10 C4 200 C4 10 2>&1 | grep flush_k; done
) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-250 | tail -40
