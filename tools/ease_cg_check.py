"""EASE parity at a full-size config against host-certified float64 columns (tests/ease_cg.py; the same check
tests/test_gpu_fullsize.py asserts).  usage: python tools/ease_cg_check.py C3 [C4 C5]"""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ease_cg  # noqa: E402

from recpack_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
if os.environ.get("RECPACK_GEMM_MODE"):
    ctx.set_option("gemm_mode", int(os.environ["RECPACK_GEMM_MODE"]))
if os.environ.get("RECPACK_GEMM_FLUSH_K"):  # tuning: error and time against the flush period of the compensated accumulation
    ctx.set_option("gemm_flush_k", int(os.environ["RECPACK_GEMM_FLUSH_K"]))
for cfg in sys.argv[1:] or ["C3"]:
    big = cfg == "C5"
    r = ease_cg.run_parity(_lib, ctx, cfg, n_cols=(8, 4, 4) if big else (16, 8, 8), n_users=32 if big else 48)
    print(json.dumps(r), flush=True)
    import torch

    torch.cuda.empty_cache()
