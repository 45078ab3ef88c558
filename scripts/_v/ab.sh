set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ph in 0 1 2 3 9; do
export ANIREC_RR_PHASE=$ph
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rr$ph -- python3 scripts/time_topk.py 18000 18000 100 > gpurun_out/rr$ph.log 2>&1 || true
f=$(find gpurun_out/rr$ph -name '*kernel_stats.csv' | head -1)
echo "phase $ph 18k: $(grep rerank $f | cut -d, -f1-4 | tail -1)"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rs$ph -- python3 scripts/time_topk.py 350000 65536 100 > gpurun_out/rs$ph.log 2>&1 || true
f=$(find gpurun_out/rs$ph -name '*kernel_stats.csv' | head -1)
echo "phase $ph 350k/65536: $(grep rerank $f | cut -d, -f1-4 | tail -1)"
done
