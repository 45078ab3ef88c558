#!/bin/bash
# round-4 GPU call 1: microbench, A/B of the flush (new vs round-3 library), PMC passes, lazy parity tests
set -u
mkdir -p gpurun_out
R=$(pwd)
OLD=$R/anime_recommendations_amd/libanirec_r03.so
timeout -k 10 120 scripts/valu_rate > gpurun_out/r4_valu_rate.txt 2>&1 && \
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4_b_new1.json 2> gpurun_out/r4_b_new1.err && \
ANIREC_LIB_PATH=$OLD timeout -k 10 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4_b_old1.json 2> gpurun_out/r4_b_old1.err && \
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4_b_new2.json 2> gpurun_out/r4_b_new2.err && \
ANIREC_LIB_PATH=$OLD timeout -k 10 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4_b_old2.json 2> gpurun_out/r4_b_old2.err && \
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "lazy or long_horizon or s109m" > gpurun_out/r4_t1.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t1.log
tail -5 gpurun_out/r4_t1.log
scripts/pmc_valu.sh new && scripts/pmc_valu.sh old $OLD
