#!/bin/bash
# round 5, experiment 5: the default bench (5 steps after 2) on the new product kernel, with the clock / power sampler
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ls /sys/class/drm/ 2>&1 | head; ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -40
python bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --ease-config none --tars-config none > gpurun_out/r05_e5_bench.json 2> gpurun_out/r05_e5_bench.err
tail -3 gpurun_out/r05_e5_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_e5_bench.json'))
print(d['value'], d['value_itemknn'], d['value_ease_fit'], d['value_ease_predict'])
print(d['roofline']['frac'], d['roofline']['solve_ms_per_step'])
print(d['config']['device'])
print(d['phases_ms_per_step'])
PY
