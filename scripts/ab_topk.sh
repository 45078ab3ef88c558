#!/bin/bash
# same-box A/B of two builds of libanirec.so on the MFMA top-k (interleaved rounds, one process per run):
#   scripts/ab_topk.sh <other .so> [n] [nq] [k] [rounds]
OTHER=$1; N=${2:-350000}; NQ=${3:-65536}; K=${4:-100}; R=${5:-3}
for r in $(seq 1 $R); do
  echo "A (tree)  : $(timeout -k 10 120 python scripts/time_topk.py $N $NQ $K 2>/dev/null | grep 'HIP events')"
  echo "B ($OTHER): $(ANIREC_LIB_PATH=$OTHER timeout -k 10 120 python scripts/time_topk.py $N $NQ $K 2>/dev/null | grep 'HIP events')"
done
