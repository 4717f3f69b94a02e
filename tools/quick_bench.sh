#!/bin/bash
# usage: tools/quick_bench.sh C2 C4   (prints ms/step, phases, roofline frac per config)
for c in "$@"; do
python bench.py --config $c --steps 3 --warmup 1 --cpu-baseline-seconds 0 --ease-config none 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['phases_ms_per_step'], d['roofline']['frac'], d['roofline']['spgemm_all_kernels'])"
done
