#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/kt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/ease_bench.py 1000000 100000 0.0005 > gpurun_out/r04f_kt.log 2>&1
tail -2 gpurun_out/r04f_kt.log
python3 tools/chol_tail_trace.py gpurun_out/kt 600 2 > gpurun_out/r04f_chain.txt 2>&1
python3 tools/chol_tail_trace.py gpurun_out/kt 700 2 >> gpurun_out/r04f_chain.txt 2>&1
python3 tools/chol_tail_trace.py gpurun_out/kt 200 1 >> gpurun_out/r04f_chain.txt 2>&1
rm -rf gpurun_out/kt
cat gpurun_out/r04f_chain.txt
