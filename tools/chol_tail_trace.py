"""The per-block chain of the blocked Cholesky from a rocprofv3 kernel trace (diagnostic).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/ease_bench.py 1000000 100000 0.0005
    python tools/chol_tail_trace.py gpurun_out/kt [first_block] [blocks]

Prints every kernel between the `first_block`-th and the (first_block + blocks)-th launch of k_potrf_trtri of the first fit: start
(us after the first of them), duration, queue, grid - what the chain of one 128-column block consists of and where it waits."""
import csv
import glob
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 600
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")),
                     r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
rows.sort()
potrf = [i for i, r in enumerate(rows) if r[2].startswith("k_potrf_trtri")]
print(f"{len(rows)} kernels, {len(potrf)} k_potrf_trtri launches")
a, b = potrf[first], potrf[first + nblk]
t0 = rows[a][0]
for r in rows[a:b + 1]:
    name = r[2].split("(")[0].replace("void ", "")
    print(f"{(r[0] - t0) / 1e3:9.1f} us  +{(r[1] - r[0]) / 1e3:8.1f} us  q{r[3]:>3}  grid {r[4]:>9} wg {r[5]:>5}  {name}")
# the period of the chain over the whole factorisation
per = [(rows[potrf[i + 1]][0] - rows[potrf[i]][0]) / 1e3 for i in range(min(781, len(potrf) - 1))]
for lo in range(0, len(per), 64):
    seg = per[lo:lo + 64]
    print(f"blocks {lo:4d}..{lo + len(seg) - 1:4d}: period mean {sum(seg) / len(seg):7.1f} us, min {min(seg):7.1f}")
