#!/bin/bash
# Texture-addresser / L1 counters per kernel (is a kernel bound by scattered-address processing?): tools/pmc_ta.sh [config]
# Four short passes; prints per-kernel averages.  TA_TA_BUSY_sum / (n_TA x GRBM_GUI_ACTIVE / 8) = share of the
# kernel's time the address units were busy; TCP_TOTAL_ACCESSES = cache-line requests the L1s processed.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=${1:-C4}
pass() {
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_ta -- python3 bench.py --config $C --steps 1 --warmup 1 --cpu-baseline-seconds 0 --no-ease-in-step --ease-config none --tars-config none --itemknn-config none --no-parity-check > gpurun_out/pmc_ta.log 2>&1
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_ta/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_predict" in k or "k_itemknn_fit" in k:
            a = acc[(k[:58], row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for f in glob.glob("gpurun_out/pmc_ta/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_predict" in k or "k_itemknn_fit" in k:
            d = dur[k[:58]]
            d[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6; d[1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print(f"{kn:58s} {c:36s} {v/n:18.0f}  (x{n}, {dur[kn][0]/max(1,dur[kn][1]):8.2f} ms per dispatch)")
PY
  rm -rf gpurun_out/pmc_ta
}
# (at most two counters of a block per pass: more "exceeds the capabilities of the hardware to collect" and the profiler aborts)
pass TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
