cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP_CHOL_TRACE=1 python3 tools/plane_cache_check.py 1000000 100000 0.0005 1 > gpurun_out/trace_pc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pc -- python3 tools/plane_cache_check.py 400000 60000 0.001 1 > gpurun_out/prof_pc.log 2>&1
cp gpurun_out/prof_pc/*/*kernel_stats.csv gpurun_out/kernel_stats_pc.csv
cp gpurun_out/prof_pc/*/*kernel_trace.csv gpurun_out/kernel_trace_pc.csv; rm -rf gpurun_out/prof_pc
head -12 gpurun_out/kernel_stats_pc.csv | cut -c1-150
