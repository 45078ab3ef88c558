#!/bin/bash
# PMC passes over the MFMA top-k (350k keys x 65536 queries, k=10); one counter group per run.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_topk
mkdir -p $OUT
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -o pmc -- python3 /root/repo/scripts/time_topk.py 350000 65536 ${1:-10} > $OUT/g$i.log 2>&1
  echo "group $i done"
done
