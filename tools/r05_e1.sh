#!/bin/bash
# round 5, experiment 1: K0 empty-row test; C4 solve baseline with traces; flush period and panel sweeps on ONE box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_e1.log
(python -m pytest tests/test_gpu_itemknn.py -x -q -m gpu -k "empty_users or edge_cases or row_shard" 2>&1 | tail -3
 B="python tools/ease_bench.py 1000000 100000 0.0005"
 echo "== baseline + traces"; RP_INV_TRACE=1 RECPACK_BENCH_ITERS=2 $B
 echo "== baseline"; $B
 echo "== flush 2048"; RECPACK_GEMM_FLUSH_K=2048 $B
 echo "== flush 4096"; RECPACK_GEMM_FLUSH_K=4096 $B
 echo "== inv_panel 32"; RECPACK_INV_PANEL=32 $B
 echo "== inv_panel 8"; RECPACK_INV_PANEL=8 $B
 echo "== chol_panel 16"; RECPACK_CHOL_PANEL=16 $B
 echo "== chol_panel 64"; RECPACK_CHOL_PANEL=64 $B
 echo "== chol_sub 4"; RECPACK_CHOL_SUB=4 $B
 echo "== chol_sub 16"; RECPACK_CHOL_SUB=16 $B
 echo "== baseline again"; $B
 rocm-smi --showclocks --showpower 2>&1 | head -30
 echo "== flush check"; python tools/ease_flush_check.py 2048 C3 200 C3 10; python tools/ease_flush_check.py 4096 C3 200 C3 10
) > $L 2>&1
grep -v amdgpu.ids $L | tail -80
