#!/bin/bash
# round-4 measurement pass on the GPU box: bench lines, kernel stats of the same command, PMC passes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err; tail -c 300 gpurun_out/bench_r04.json
python3 bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --ease-config none --tars-config none > gpurun_out/bench_r04_driver.json 2> gpurun_out/bench_r04_driver.err; tail -c 300 gpurun_out/bench_r04_driver.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04 -- python3 bench.py --steps 3 --warmup 2 --cpu-baseline-seconds 0 --ease-config none --tars-config none > gpurun_out/prof_r04.log 2>&1
cp gpurun_out/prof_r04/*/*kernel_stats.csv gpurun_out/kernel_stats_r04.csv; rm -rf gpurun_out/prof_r04
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.txt 2>&1
bash tools/pmc_sq.sh k_ C4 > gpurun_out/pmc_sq_all.txt 2>&1
bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma.txt 2>&1
tail -5 gpurun_out/pmc_mfma.txt
bash tools/pmc_gemm.sh 8192 1 > gpurun_out/pmc_gemm.txt 2>&1
bash tools/pmc_ta.sh C4 > gpurun_out/pmc_ta.txt 2>&1
ls -la gpurun_out | tail -15
