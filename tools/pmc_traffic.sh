#!/bin/bash
# HBM traffic from PMC counters (separate passes per counter, as MI355X_MICROARCH.md prescribes; --kernel-trace only):
#   tools/pmc_traffic.sh            (on the GPU box; prints per-kernel averages)
# 1. the ItemKNN step at C4 -> gpurun_out/pmc_traffic_c4.json          (what bench.py's roofline.traffic reads from profiles/)
# 2. the whole step (ItemKNN + EASE: K4-K7, K9's two passes) at C3  -> gpurun_out/pmc_traffic_ease_c3.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # tag, command...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -- "$@" > gpurun_out/pmc_${tag}_$c.log 2>&1
  done
  python3 - "$tag" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"]
            if any(s in k for s in ("k_itemknn", "k_predict", "k_rows", "k_partition", "k_bucket", "k_gemm", "k_split", "k_ease", "k_potrf", "k_dense")):
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
    for k, (v, n) in acc.items():
        out.setdefault(k, {})[c] = {"avg_per_dispatch": v / max(1, n), "dispatches": n}
# what the pass was taken on: bench.py compares these with the sources it runs beside (a checkout's mtimes say nothing)
import hashlib
out["_sources_sha256"] = {f: hashlib.sha256(open(f"recpack_amd/csrc/{f}", "rb").read()).hexdigest()
                          for f in ("rp_predict.hip", "rp_predict_i8.hip", "rp_itemknn.hip", "rp_csr.hip", "rp_wave.h", "rp_select.h")}
print(json.dumps(out, indent=1))
json.dump(out, open(f"gpurun_out/pmc_traffic_{tag}.json", "w"), indent=1)
PY
  rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE  # (raw CSVs: tens of MB; gpurun copies back <= 64 MiB)
}
run c4 python3 bench.py --config C4 --steps 2 --warmup 1 --cpu-baseline-seconds 0 --no-ease-in-step --ease-config none --tars-config none --itemknn-config none --no-parity-check
run ease_c3 python3 bench.py --config C3 --steps 1 --warmup 1 --cpu-baseline-seconds 0 --ease-config none --tars-config none --itemknn-config none --no-parity-check
