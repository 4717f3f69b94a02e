#!/bin/bash
# round 5, experiment 16: gathered model in two phases (int8 rows first, f32 strips on their own stream): gloo parity + 8-rank emulation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e16.log
(echo
 python tools/dist_rank_emulate.py --world 8 --ranks 0,7 --repeats 1 --out gpurun_out/r05_rank_emulation_b.json 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['single_gpu']); print(d['speedup_emulated']); print({k: v for k, v in d['ranks']['7'].items() if k.endswith('_s')})"
) > $L 2>&1
grep -v amdgpu.ids $L | cut -c1-900
