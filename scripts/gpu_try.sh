#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r04_gputests.log
tail -5 gpurun_out/r04_gputests.log
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
echo "bench rc=$?"
timeout -k 10 600 python scripts/time_dist_shape.py 1 2 4 8 > gpurun_out/r04_dist_shape.txt 2>&1
echo "shape rc=$?"
grep "^G" gpurun_out/r04_dist_shape.txt
