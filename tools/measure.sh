#!/bin/bash
# The round's measurement pass on the GPU box (outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/):
#   tools/measure.sh r06
# 1. the default bench line; 2. the driver-shaped run (25 steps after 5); 3. rocprofv3 kernel stats of the same command;
# 4. PMC traffic (FETCH / WRITE, separate passes) of the ItemKNN step at C4 and of the whole step at C3; 5. SQ counters of the SpGEMM
# kernels; 6. MFMA counters of the solve at C3; 7. the solve's timeline at C4.  Every rocprofv3 command has the program right after `--`.
t=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 bench.py > gpurun_out/${t}_default_bench.json 2> gpurun_out/${t}_default_bench.err
timeout 600 python3 bench.py --steps 25 --warmup 5 --cpu-baseline-seconds 0 --ease-config none --tars-config none --itemknn-config none > gpurun_out/${t}_driver_shaped_bench.json 2>> gpurun_out/${t}_default_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${t}_prof -- python3 bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --ease-config none --tars-config none --itemknn-config none --no-parity-check > gpurun_out/${t}_prof.log 2>&1
cp $(find gpurun_out/${t}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${t}_default_kernel_stats.csv
rm -rf gpurun_out/${t}_prof
timeout 900 bash tools/pmc_traffic.sh > gpurun_out/${t}_pmc_traffic.log 2>&1
mv gpurun_out/pmc_traffic_c4.json gpurun_out/${t}_pmc_traffic_c4.json
mv gpurun_out/pmc_traffic_ease_c3.json gpurun_out/${t}_pmc_traffic_step_c3.json
timeout 600 bash tools/pmc_sq.sh k_itemknn C4 > gpurun_out/${t}_pmc_sq_c4.txt 2>&1
timeout 600 bash tools/pmc_sq.sh k_predict_wave C4 >> gpurun_out/${t}_pmc_sq_c4.txt 2>&1
timeout 600 bash tools/pmc_mfma.sh > gpurun_out/${t}_pmc_mfma.log 2>&1
mv gpurun_out/pmc_mfma.json gpurun_out/${t}_pmc_mfma_solve_c3.json
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${t}_kt -- python3 tools/ease_bench.py 1000000 100000 0.0005 > gpurun_out/${t}_kt.log 2>&1
python3 tools/solve_timeline.py gpurun_out/${t}_kt > gpurun_out/${t}_solve_timeline.txt 2>&1
rm -rf gpurun_out/${t}_kt
python3 - "$t" <<'PY'
import json, sys
t = sys.argv[1]
for f in (f"gpurun_out/{t}_default_bench.json", f"gpurun_out/{t}_driver_shaped_bench.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["status"], d["value"], d["roofline"]["frac"], d["roofline_hbm"]["spgemm_all_kernels"]["ms"], d.get("parity", {}).get("ease_probe_rel"))
PY
head -30 gpurun_out/${t}_solve_timeline.txt
