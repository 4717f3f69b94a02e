#!/bin/bash
# HBM traffic of the SpGEMM kernels from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes):
#   tools/pmc_traffic.sh            (on the GPU box; writes gpurun_out/pmc_*/ and prints per-kernel averages)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --config C4 --steps 2 --warmup 1 --cpu-baseline-seconds 0 --ease-config none"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- $CMD > gpurun_out/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"]
            if "k_itemknn_fit" in k or "k_predict" in k or "k_rank_hist" in k or "k_scatter" in k or "k_itemknn_merge" in k:
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
    for k, (v, n) in acc.items():
        out.setdefault(k, {})[c] = {"avg_per_dispatch": v / max(1, n), "dispatches": n}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_traffic_c4.json", "w"), indent=1)
PY
