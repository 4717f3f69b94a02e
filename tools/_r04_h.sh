#!/bin/bash
bash tools/_r04_full.sh
bash tools/_r04_bench.sh
