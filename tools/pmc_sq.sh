#!/bin/bash
# SQ counters of one kernel family: tools/pmc_sq.sh <kernel-substring> [config]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-k_predict_wave}; C=${2:-C4}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --config $C --steps 1 --warmup 1 --cpu-baseline-seconds 0 --no-ease-in-step --ease-config none --tars-config none --itemknn-config none --no-parity-check > gpurun_out/pmc_sq.log 2>&1
python3 - "$K" <<'PY'
import csv, glob, collections, sys
k = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if k in row["Kernel_Name"]:
            a = acc[(row["Kernel_Name"][:60], row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print(f"{kn:60s} {c:22s} {v/n:16.0f}  (x{n})")
PY
rm -rf gpurun_out/pmc_sq
