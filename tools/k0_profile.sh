#!/bin/bash
# per-kernel time of K0 (rphip_csr_from_device) at C4: rocprofv3 kernel stats of five constructions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/k0p
cat > /tmp/k0_run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from recpack_amd import _lib
from recpack_amd.synth import CONFIGS, config_csr
dev = torch.device("cuda", 0)
ip, ix = config_csr("C4", device=dev)
U, I, _ = CONFIGS["C4"]
ctx = _lib.Context(0)
for it in range(6):
    ctx.reset_timings()
    dX = _lib.DeviceCSR.from_device(ctx, ip, ix, (U, I))
    ctx.sync()
    print("K0 ms", round(ctx.timings()["csr_transpose"]["ms"], 3), flush=True)
    dX.close()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k0p -- python3 /tmp/k0_run.py 2>&1 | grep "K0 ms"
f=$(find gpurun_out/k0p -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_") or "fill" in n.lower():
        print(f"{n[:44]:46s} x{int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf gpurun_out/k0p
