#!/bin/bash
# one rank of eight at C5 (5M x 200K, 160 GB of B on one GPU).  Two processes: the single-GPU step alone,
# then ranks 0 and 7 with the owner route on the emulated fit's own data (worst case: every user takes the exact pass).
timeout 900 python tools/dist_rank_emulate.py --config C5 --world 8 --ranks "" --repeats 1 --out gpurun_out/r06_rank_emulation_c5_single.json > gpurun_out/r06_emu_c5_single.log 2>&1
tail -2 gpurun_out/r06_emu_c5_single.log | cut -c1-600
timeout 1500 python tools/dist_rank_emulate.py --config C5 --world 8 --ranks 0,7 --repeats 1 --skip-single --owner-on-emulated --out gpurun_out/r06_rank_emulation_c5_ranks.json > gpurun_out/r06_emu_c5_ranks.log 2>&1
tail -3 gpurun_out/r06_emu_c5_ranks.log | cut -c1-1500
