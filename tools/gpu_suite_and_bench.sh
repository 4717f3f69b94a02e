#!/bin/bash
bash tools/gpu_suite.sh
bash tools/bench_checks.sh
