#!/bin/bash
# round 5, experiment 6: wave priorities in K8 / K9 / the SpGEMM kernels (bench phases per variant, one box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e6.log
(for v in "" _x11 _x20 ""; do
   export RECPACK_HIP_LIB=$GRAFT_REPO_ROOT/recpack_amd/csrc/librecpack_hip$v.so
   echo "== variant '$v'"
   python bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --ease-config none --tars-config none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['phases_ms_per_step']
print(d['value'], {k: p[k] for k in ('csr_transpose','gram_topk','gram_topk_wave','topk_merge','predict','predict_dense','cholesky','inverse')})"
 done) > $L 2>&1
cat $L
