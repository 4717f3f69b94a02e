#!/bin/bash
# round 5, experiment 23: where an emulated rank's distributed fit goes (kernel trace of the second fit, by kernel and queue)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/kt3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt3 -- python3 tools/r05_e23_fit_w8.py > gpurun_out/r05_e23.log 2>&1
grep "^fit" gpurun_out/r05_e23.log
python - <<'PY'
import csv, glob, re
f = sorted(glob.glob("gpurun_out/kt3/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
wall = float(re.findall(r"fit 1: wall ([0-9.]+) ms", open("gpurun_out/r05_e23.log").read())[0])
end = max(e for s, e, n, q in rows if n.startswith("k_dist_scale"))
t0 = end - wall * 1e6
agg = {}
for s, e, n, q in rows:
    if s < t0 or e > end + 1e6: continue
    k = (n[:40], q)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e6
print(f"window {wall:.1f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:42s} q{k[1]:>3} x{a[0]:5d} busy {a[1]:8.1f} ms")
# union of busy time over all kernels (any kernel running) and of the wide product
def union(sel):
    iv = sorted((s, e) for s, e, n, q in rows if s >= t0 and e <= end + 1e6 and sel(n))
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot / 1e6
# gaps of the update stream's queue (the one with most k_gemm_hp16w time) and what ran in them
byq = {}
for s_, e_, n, q in rows:
    if s_ >= t0 and e_ <= end + 1e6 and n.startswith("k_gemm_hp16w"): byq[q] = byq.get(q, 0) + e_ - s_
uq = max(byq, key=byq.get)
ur = sorted((s_, e_) for s_, e_, n, q in rows if q == uq and s_ >= t0 and e_ <= end + 1e6)
gaps = [(ur[i][1], ur[i + 1][0]) for i in range(len(ur) - 1) if ur[i + 1][0] - ur[i][1] > 2e5]
gt = sum(b - a for a, b in gaps) / 1e6
print(f"update queue q{uq}: {len(gaps)} gaps > 0.2 ms, {gt:.1f} ms in all; kernels of other queues inside them:")
ins = {}
for s_, e_, n, q in rows:
    if q == uq: continue
    for a, b in gaps:
        o = min(e_, b) - max(s_, a)
        if o > 0:
            k = (n[:40], q)
            ins[k] = ins.get(k, 0) + o / 1e6
for k, v in sorted(ins.items(), key=lambda kv: -kv[1])[:14]:
    print(f"   {k[0]:42s} q{k[1]:>3} {v:7.1f} ms")
big = sorted(gaps, key=lambda g: g[0] - g[1])[:6]
print("   longest gaps (ms from window start, length):", [(round((a - t0) / 1e6, 1), round((b - a) / 1e6, 2)) for a, b in big])
print(f"any kernel running: {union(lambda n: True):.1f} ms; a wide product (k_gemm_hp16w) running: {union(lambda n: n.startswith('k_gemm_hp16w')):.1f} ms; delay kernels (links busy): {union(lambda n: n.startswith('k_comm_delay')):.1f} ms")
PY
rm -rf gpurun_out/kt3
