#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --config C4 --steps 5 --warmup 2 --cpu-baseline-seconds 0 --no-ease-in-step --ease-config none --tars-config none > gpurun_out/r04f_b.json 2> gpurun_out/r04f_b.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04f_b.json').read().strip().splitlines()[-1])
print('value', d['value'], d['phases_ms_per_step'])
PY
python -m pytest tests/test_gpu_itemknn.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04f_tests.log; tail -3 gpurun_out/r04f_tests.log | cut -c1-200
