#!/bin/bash
# MFMA counters of the EASE solve's GEMM (k_gemm_nt): tools/pmc_mfma.sh [users items density]   (on the GPU box)
# One rocprofv3 --pmc pass (SQ block: 8 slots) with --kernel-trace only; prints per-kernel sums and the derived utilisation:
#   MFMA busy share  = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs)   [cycles the matrix pipes were busy]
#   flops from MOPS  = (SQ_INSTS_VALU_MFMA_MOPS_F16 + _BF16) x 512   (the default solve runs fp16 plane products; the _BF16 counter is kept
#                      in the pass and reads 0 since round 4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
U=${1:-200000}; I=${2:-50000}; D=${3:-0.001}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
  --output-format csv -d gpurun_out/pmc_mfma -- python3 tools/ease_bench.py $U $I $D > gpurun_out/pmc_mfma.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
dur = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_gemm_" in k or "k_potrf" in k:
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for f in glob.glob("gpurun_out/pmc_mfma/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_gemm_" in k or "k_potrf" in k:
            dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
            n[k] += 1
out = {}
for k, c in acc.items():
    d = dict(c)
    d["dispatches"] = n[k]
    d["total_seconds_profiled"] = dur[k]
    if c.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_share"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"])
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    if mops and dur[k] > 0:
        d["tflops_from_mops"] = mops * 512.0 / dur[k] / 1e12
    out[k] = d
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_mfma.json", "w"), indent=1)
PY
tail -3 gpurun_out/pmc_mfma.log
rm -rf gpurun_out/pmc_mfma
