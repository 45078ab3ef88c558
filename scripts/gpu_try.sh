#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_infer_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "predict_grid or predict" > gpurun_out/try_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/try_tests.log
tail -4 gpurun_out/try_tests.log
timeout -k 10 200 python scripts/time_predict.py 100000 5 v2 v3 v2dbg1 v3dbg1 v2dbg2 v3dbg2 v3p3 v3p6 v3p17 2>&1 | grep -v amdgpu.ids > gpurun_out/try_pred.txt
cat gpurun_out/try_pred.txt
timeout -k 10 200 python scripts/time_predict.py 10000 2 v2 v3 v1 2>&1 | grep -v amdgpu.ids
