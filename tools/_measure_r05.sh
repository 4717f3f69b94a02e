#!/bin/bash
# round-5 measurement pass on the GPU box: bench lines, kernel stats of the same command, PMC passes, solve timeline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err; tail -c 300 gpurun_out/bench_r05.json
python3 bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --ease-config none --tars-config none > gpurun_out/bench_r05_driver.json 2> gpurun_out/bench_r05_driver.err; tail -c 300 gpurun_out/bench_r05_driver.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -- python3 bench.py --steps 3 --warmup 2 --cpu-baseline-seconds 0 --ease-config none --tars-config none > gpurun_out/prof_r05.log 2>&1
cp gpurun_out/prof_r05/*/*kernel_stats.csv gpurun_out/kernel_stats_r05.csv; rm -rf gpurun_out/prof_r05
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.txt 2>&1
bash tools/pmc_sq.sh k_ C4 > gpurun_out/pmc_sq_all.txt 2>&1
bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma.txt 2>&1
tail -5 gpurun_out/pmc_mfma.txt
bash tools/pmc_gemm.sh 8192 1 > gpurun_out/pmc_gemm.txt 2>&1
bash tools/r05_e7.sh > gpurun_out/r05_e7.txt 2>&1
ls -la gpurun_out | tail -15
