set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_infer_gpu.py tests/test_recs_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
for L in scripts/_v/libanirec_head.so anime_recommendations_amd/libanirec.so; do
ANIREC_LIB_PATH=$PWD/$L timeout -k 10 120 python scripts/time_topk_reps.py 18000 18000 100 5
ANIREC_LIB_PATH=$PWD/$L timeout -k 10 120 python scripts/time_topk_reps.py 350000 350000 100 4
ANIREC_LIB_PATH=$PWD/$L timeout -k 10 120 python scripts/time_topk_reps.py 350000 65536 10 3
done; done
timeout -k 10 300 python scripts/time_predict_topk.py 2>&1 | tail -1
