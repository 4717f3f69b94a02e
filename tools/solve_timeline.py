"""Where the EASE solve's wall time goes, from a rocprofv3 kernel trace (diagnostic).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/ease_bench.py 1000000 100000 0.0005
    python tools/solve_timeline.py gpurun_out/kt [fit index, default: the last]

For one fit: the factorisation (first .. last k_potrf_trtri) and the inverse (from there to k_ease_scale_sym) split into the time
during which a wide plane product (k_gemm_hp16w) is running and the time during which none is - the chip then runs only the
per-block chain (or nothing) - per eighth of each phase, with the kernels that fill the gaps."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
scale = [i for i, r in enumerate(rows) if r[2].startswith("k_ease_scale")]
init = [i for i, r in enumerate(rows) if r[2].startswith("k_ease_init")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(scale) - 1
a, b = init[which], scale[which]
fit = rows[a:b + 1]
potrf = [r for r in fit if r[2].startswith("k_potrf_trtri")]
t_c0, t_c1, t_i1 = potrf[0][0], potrf[-1][1], fit[-1][0]


def phase(name, t0, t1, parts=8):
    ev = [r for r in fit if r[1] > t0 and r[0] < t1]
    wide = sorted((max(r[0], t0), min(r[1], t1)) for r in ev if r[2].startswith("k_gemm_hp16w"))
    # union of the wide products' intervals
    merged = []
    for s, e in wide:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    busy = sum(e - s for s, e in merged)
    print(f"{name}: {(t1 - t0) / 1e6:.1f} ms, a wide product running {busy / 1e6:.1f} ms, none {(t1 - t0 - busy) / 1e6:.1f} ms")
    step = (t1 - t0) / parts
    for p in range(parts):
        s0, s1 = t0 + p * step, t0 + (p + 1) * step
        b = sum(max(0, min(e, s1) - max(s, s0)) for s, e in merged)
        gaps = defaultdict(float)
        # kernels (other than the wide product) that run while no wide product does
        holes = []
        cur = s0
        for s, e in merged:
            if e <= s0 or s >= s1:
                continue
            if s > cur:
                holes.append((cur, min(s, s1)))
            cur = max(cur, e)
        if cur < s1:
            holes.append((cur, s1))
        for r in ev:
            if r[2].startswith("k_gemm_hp16w"):
                continue
            for hs, he in holes:
                o = min(r[1], he) - max(r[0], hs)
                if o > 0:
                    gaps[r[2]] += o
        top = sorted(gaps.items(), key=lambda kv: -kv[1])[:4]
        print(f"  part {p}: wide {b / 1e6:7.1f} ms of {step / 1e6:6.1f}; in the holes: " + ", ".join(f"{k} {v / 1e6:.1f}" for k, v in top))


phase("factorisation", t_c0, t_c1)
phase("inverse", t_c1, t_i1)
tot = defaultdict(lambda: [0, 0.0])
for r in fit:
    tot[r[2]][0] += 1
    tot[r[2]][1] += (r[1] - r[0]) / 1e6
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k:40s} {v[0]:6d} launches {v[1]:9.1f} ms of kernel time")
