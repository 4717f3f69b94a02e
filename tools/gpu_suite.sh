#!/bin/bash
# the whole `-m gpu` suite as the driver runs it (timed), then the default bench line:  tools/gpu_suite.sh [out-prefix]
p=${1:-gpurun_out/suite}
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 900 --timeout-method thread --durations=12 ) > ${p}_tests.log 2>&1
grep -n "passed\|failed\|^real\|s call\|s setup" ${p}_tests.log | tail -20
timeout 900 python bench.py > ${p}_bench.json 2> ${p}_bench.err
python - <<PY
import json
d = json.loads(open("${p}_bench.json").read().strip().splitlines()[-1])
print(d["status"], "value", d["value"], "itemknn", d["value_itemknn"], "ease fit", d["value_ease_fit"], "predict", d["value_ease_predict"])
print("roofline", d["roofline"]["frac"], d["roofline"]["frac_n3_model"], d["roofline"]["solve_ms_per_step"])
print("spgemm", d["roofline_hbm"]["spgemm_all_kernels"])
print("parity", d.get("parity"))
print("c2", {k: d["itemknn_c2"][k] for k in ("fit_s", "predict_s", "spgemm_all_kernels")})
print("c3", {k: d["ease_c3"][k] for k in ("fit_s", "predict_s")})
print("device", d["config"]["device"].get("under_load"))
print("phases", d["phases_ms_per_step"])
PY
