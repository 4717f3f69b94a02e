#!/bin/bash
# SQ counters of the plane product kernels on a square product: tools/pmc_gemm.sh [n [wide]]   (on the GPU box; two --pmc passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-8192}; W=${2:-1}
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmc_gemm$i -- python3 tools/gemm_rate.py $N $W 3 > gpurun_out/pmc_gemm$i.log 2>&1
  tail -2 gpurun_out/pmc_gemm$i.log
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_gemm*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_gemm_hp" in k or "k_split" in k or "k_gemm_amax" in k:
            a = acc[(k, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print(f"{kn:24s} {c:26s} {v/n:18.0f}  (x{n})")
PY
rm -rf gpurun_out/pmc_gemm1 gpurun_out/pmc_gemm2
