#!/bin/bash
# the ItemKNN half's phases at a config (default C2 and C4), no extra legs: tools/quick_bench.sh [C2 C4 ...]
for c in ${@:-C2 C4}; do
timeout 600 python bench.py --config $c --steps 3 --warmup 1 --cpu-baseline-seconds 0 --ease-config none --tars-config none --itemknn-config none --no-parity-check --no-ease-in-step 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['status'], d['ms_per_step'], d['phases_ms_per_step'], d['roofline']['spgemm_all_kernels'])"
done
