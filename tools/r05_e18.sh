#!/bin/bash
# round 5, experiment 18: kernel time by stream of an emulated rank's distributed EASE fit (W = 8, rank 7)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/kt3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt3 -- python3 tools/dist_rank_emulate.py --world 8 --ranks 7 --repeats 0 --skip-single > gpurun_out/r05_e18.log 2>&1
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/kt3/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?"), r.get("Grid_Size", "?")) for r in csv.DictReader(open(f))]
rows.sort()
# the fit: from the first k_blk_transpose / k_unpack_pieces region: take between the dense Gram kernel (k_itemknn_fit<false, 1024, 32>) and k_dist_scale
g = [r for r in rows if r[2].startswith("k_itemknn_fit<false, 1024, 32>")][-1]
e = [r for r in rows if r[2].startswith("k_dist_scale")][-1]
t0, t1 = g[0], e[1]
print("fit window %.1f ms" % ((t1 - t0) / 1e6))
agg = {}
for s, en, n, q, gr in rows:
    if en < t0 or s > t1: continue
    k = (q, n[:30])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (en - s) / 1e6
for k, a in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    if a[1] > 0.5: print(f"q{k[0]:>3} {k[1]:32s} x{a[0]:5d} {a[1]:8.1f} ms")
PY
rm -rf gpurun_out/kt3
