#!/bin/bash
# round 5, experiment 13: transport self-test (gloo world 2 / 3 on one GPU, RCCL world 1), bench --selftest
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e13.log
(python -m pytest tests/test_gpu_ease.py tests/test_gpu_itemknn.py -x -q -m gpu -k "sharded" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
 python bench.py --config C2 --steps 2 --warmup 1 --cpu-baseline-seconds 0 --ease-config none --tars-config none --selftest 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['status'], d['value'], d['transport_selftest'])"
) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-600
