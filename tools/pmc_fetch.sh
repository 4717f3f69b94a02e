#!/bin/bash
# HBM fetch bytes of one command's kernels: tools/pmc_fetch.sh <kernel-substring> <command...>   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=$1; shift
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- "$@" > gpurun_out/pmc_fetch.log 2>&1
python3 - "$K" <<'PY'
import csv, glob, collections, sys
k = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_fetch/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if k in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
            a = acc[row["Kernel_Name"][:70]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for f in glob.glob("gpurun_out/pmc_fetch/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if k in row["Kernel_Name"]:
            dur[row["Kernel_Name"][:70]] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
for kn, (v, n) in sorted(acc.items()):
    print(f"{kn:70s} FETCH_SIZE avg {v / n * 1024 / 1e9:10.2f} GB per dispatch (x{n}), {dur[kn] / n * 1e3:9.2f} ms per dispatch -> {v * 1024 / dur[kn] / 1e12:6.2f} TB/s")
PY
tail -3 gpurun_out/pmc_fetch.log
rm -rf gpurun_out/pmc_fetch
