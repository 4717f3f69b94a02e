cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" _abl1024; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k9$v -- python3 tools/k9_ablate.py recpack_amd/csrc/librecpack_hip$v.so C4 > gpurun_out/prof_k9$v.log 2>&1
cp gpurun_out/prof_k9$v/*/*kernel_stats.csv gpurun_out/kernel_stats_k9$v.csv; rm -rf gpurun_out/prof_k9$v
grep -E "k_predict|k_dense" gpurun_out/kernel_stats_k9$v.csv | cut -c1-140
done
