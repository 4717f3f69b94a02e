#!/bin/bash
# round 5, experiment 17: kernel timeline of an emulated rank's gather + predict (does the first pass run beside the f32 strips?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/kt2
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/kt2 -- python3 tools/dist_rank_emulate.py --world 8 --ranks 7 --repeats 0 > gpurun_out/r05_e17.log 2>&1
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/kt2/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
# the last k_predict_user_i8 launch with more than 10 ms = the rank's predict
pu = [r for r in rows if r[2].startswith("k_predict_user_i8") and r[1] - r[0] > 10e6][-1]
t0 = pu[0] - 60e6
print("first pass: start 60.0 ms, end %.1f ms, queue %s" % ((pu[1] - t0) / 1e6, pu[3]))
agg = {}
for s, e, n, q in rows:
    if e < t0 or s > pu[1] + 80e6: continue
    k = (n[:34], q)
    a = agg.setdefault(k, [0, 0.0, 1e18, 0])
    a[0] += 1; a[1] += (e - s) / 1e6; a[2] = min(a[2], s); a[3] = max(a[3], e)
for k, a in sorted(agg.items(), key=lambda kv: kv[1][2]):
    print(f"{k[0]:36s} q{k[1]:>3} x{a[0]:4d} busy {a[1]:8.1f} ms  first {(a[2]-t0)/1e6:7.1f} last end {(a[3]-t0)/1e6:7.1f}")
PY
rm -rf gpurun_out/kt2
