python tools/k8_variants.py C4 > gpurun_out/k8v.log 2>&1; tail -8 gpurun_out/k8v.log
python tools/ease_cg_check.py C3 C4 > gpurun_out/cg2.log 2>&1; grep -v "cg iter" gpurun_out/cg2.log | tail -6
python tools/ease_bench.py C4 > gpurun_out/easeb.log 2>&1; tail -5 gpurun_out/easeb.log
