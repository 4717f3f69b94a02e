#!/bin/bash
# round 5, experiment 12: kernel stats of the K9 routes (with / without history removal)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 tools/k9_route_bench.py C4 > gpurun_out/r05_e12.log 2>&1
grep -v amdgpu.ids gpurun_out/r05_e12.log | tail -12
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/ks/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))]
rows.sort()
# group by predict call: a call starts with k_predict_user_i8 or the first k_predict_tile<true, true after a gap
calls, cur = [], []
for r in rows:
    n = r[2]
    if not (n.startswith("k_predict") or n.startswith("k_dense_to")):
        continue
    if n.startswith("k_predict_user_i8") or (n.startswith("k_predict_tile<true, true") and (not cur or not cur[-1][2].startswith("k_predict_tile<true, true") and not cur[-1][2].startswith("k_predict_merge") and not cur[-1][2].startswith("k_predict_rescore"))):
        if cur: calls.append(cur)
        cur = []
    cur.append(r)
if cur: calls.append(cur)
for c in calls:
    d = {}
    for s, e, n in c:
        d[n] = d.get(n, 0) + (e - s) / 1e6
    print(" | ".join(f"{k[:38]} {v:.1f}" for k, v in d.items()))
PY
rm -rf gpurun_out/ks
