#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RECPACK_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config C2 --steps 2 --warmup 1 --cpu-baseline-seconds 0 --ease-config none --tars-config none > gpurun_out/r04_bench_g2.json 2> gpurun_out/r04_bench_g2.err; echo "g2 rc=$?"
tail -c 1500 gpurun_out/r04_bench_g2.json; tail -5 gpurun_out/r04_bench_g2.err | cut -c1-300
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 4 > gpurun_out/r04_bench_quick.json 2> gpurun_out/r04_bench_quick.err; echo "quick rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_quick.json').read().strip().splitlines()[-1])
for k in ('value','status','cold_first_step_s','value_itemknn','value_ease_fit','value_ease_predict','transport'): print(k, d.get(k))
print('roofline', {k:d['roofline'].get(k) for k in ('bound','achieved','peak','frac','frac_n3_model','share_of_step','solve_ms_per_step')})
print('roofline_hbm', {k:d['roofline_hbm'].get(k) for k in ('kernel','frac','avg_launch_ms')}, d['roofline_hbm']['spgemm_all_kernels'])
print('k9', d.get('k9_predict'))
print('phases', d['phases_ms_per_step'])
print('ease_c3', {k:d['ease_c3'].get(k) for k in ('fit_s','predict_s','wall_s_all_runs')})
PY
