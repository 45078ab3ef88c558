#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh r03 train train7m topkall topkall10 topk100 topk18k pgrid ptk ingest recs
# For each workload: one `--kernel-trace --stats` run, then two SEPARATE `--pmc` runs (FETCH_SIZE, WRITE_SIZE:
# never combined with each other or with other trace domains).  Output: gpurun_out/<round>/<workload>_{stats,fetch,write}/.
# scripts/summarise_profiles.py turns them into profiles/<round>_*.
set -u
ROUND=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  case $W in
    train)   CMD="$ROOT/bench.py --no-also --no-cpu-baseline --steps 96 --warmup 32";;
    train7m) CMD="$ROOT/bench.py --workload s7m --no-also --no-cpu-baseline --steps 96 --warmup 32";;
    topk100) CMD="$ROOT/scripts/time_topk.py 350000 65536 100";;
    topk10)  CMD="$ROOT/scripts/time_topk.py 350000 65536 10";;
    topkall) CMD="$ROOT/scripts/time_topk.py 350000 350000 100";;
    topkall10) CMD="$ROOT/scripts/time_topk.py 350000 350000 10";;
    topk18k) CMD="$ROOT/scripts/time_topk.py 18000 18000 100";;
    pgrid)   CMD="$ROOT/scripts/time_predict.py 100000 1 v2";;
    ptk)     CMD="$ROOT/scripts/time_predict_topk.py";;
    ingest)  CMD="$ROOT/scripts/time_ingest.py";;
    recs)    CMD="$ROOT/scripts/time_recs.py";;
    *) echo "unknown workload $W"; continue;;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -o p -- python3 $CMD > $OUT/${W}_stats.log 2>&1
  echo "$W stats rc=$?"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${W}_fetch -o p -- python3 $CMD > $OUT/${W}_fetch.log 2>&1
  echo "$W fetch rc=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${W}_write -o p -- python3 $CMD > $OUT/${W}_write.log 2>&1
  echo "$W write rc=$?"
done
