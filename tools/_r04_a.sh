#!/bin/bash
# round-4 iteration script: ItemKNN parity + fit timing at C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_itemknn.py tests/test_gpu_fuzz.py tests/test_gpu_tars.py -x -q -m gpu > gpurun_out/r04a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04a_tests.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "itemknn" > gpurun_out/r04a_full.log 2>&1; echo "full rc=$?" >> gpurun_out/r04a_full.log
python tools/fit_ablate.py recpack_amd/csrc/librecpack_hip.so C4 gram > gpurun_out/r04a_fit.log 2>&1
python tools/fit_ablate.py recpack_amd/csrc/librecpack_hip.so C2 >> gpurun_out/r04a_fit.log 2>&1
tail -3 gpurun_out/r04a_tests.log; tail -3 gpurun_out/r04a_full.log; cat gpurun_out/r04a_fit.log
