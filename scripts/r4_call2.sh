#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r4_t2.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t2.log
tail -15 gpurun_out/r4_t2.log
timeout -k 10 600 python scripts/time_dist_shape.py 1 2 8 > gpurun_out/r4_dist_shape.txt 2>&1
echo "shape rc=$?"
tail -20 gpurun_out/r4_dist_shape.txt
