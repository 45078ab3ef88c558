#!/bin/bash
# round-4 evidence for the training path: kernel stats + traffic (collect_profiles) + the VALU counter passes
set -u
bash scripts/collect_profiles.sh r04 train train7m
bash scripts/pmc_valu.sh r04
timeout -k 10 300 python bench.py --no-also > gpurun_out/r04_bench_train_only.json 2> gpurun_out/r04_bench_train_only.err
echo "bench rc=$?"
