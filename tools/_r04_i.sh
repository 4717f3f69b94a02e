#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/dist_rank_emulate.py --world 8 --ranks 0,7 --out gpurun_out/r04_rank_emulation.json > gpurun_out/r04_emul.log 2>&1
tail -3 gpurun_out/r04_emul.log | cut -c1-1500
